"""FocalLoss (reference loss.py:5-34): -(1-pt)^gamma * alpha_t * log(pt), pt detached."""
import torch
import torch.nn as nn


class FocalLoss(nn.Module):
    def __init__(self, gamma=0, alpha=None, size_average=True):
        super().__init__()
        self.gamma = gamma
        if isinstance(alpha, (float, int)) and not isinstance(alpha, bool):
            alpha = torch.tensor([alpha, 1 - alpha], dtype=torch.float32)
        elif isinstance(alpha, list):
            alpha = torch.tensor(alpha, dtype=torch.float32)
        self.alpha = alpha
        self.size_average = size_average

    def forward(self, input, target):
        if input.dim() > 2:
            input = input.view(input.size(0), input.size(1), -1).transpose(1, 2).contiguous().view(-1, input.size(1))
        target = target.view(-1, 1)
        logpt = input.gather(1, target).view(-1)
        pt = logpt.detach().exp()
        if self.alpha is not None:
            if self.alpha.device != input.device or self.alpha.dtype != input.dtype:
                self.alpha = self.alpha.to(device=input.device, dtype=input.dtype)
            logpt = logpt * self.alpha.gather(0, target.view(-1))
        loss = -1 * (1 - pt) ** self.gamma * logpt
        return loss.mean() if self.size_average else loss.sum()
