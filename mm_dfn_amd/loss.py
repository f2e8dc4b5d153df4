"""FocalLoss (reference loss.py:5-34): -(1-pt)^gamma * alpha_t * log(pt), pt detached."""
import torch
import torch.nn as nn


# Set by train.backward() (and only there) around ``loss.backward(one)`` when ``loss`` IS the output of the fused node below and
# ``one`` is its cached scalar 1: the node's upstream gradient is then exactly 1 and the gradient the forward launch already wrote
# is the answer.  An explicit flag, not a comparison of data pointers (an address can be reused, a seed overwritten).
UNIT_SEED_ACTIVE = [False]


def is_fused_loss_output(loss):
    """True when ``loss`` comes straight out of _FocalLossHip (so a backward seeded at ``loss`` hands the node its seed unchanged)."""
    fn = getattr(loss, "grad_fn", None)
    return fn is not None and type(fn).__name__ == "_FocalLossHipBackward"


class _FocalLossHip(torch.autograd.Function):
    """One fused launch on device tensors (csrc/focal_loss.hip): when a gradient will be wanted the forward launch also
    writes d loss / d log_prob for an upstream gradient of 1, and the backward pass launches nothing for the loss when it
    is seeded with train.backward's cached 1 (any other upstream gradient takes the backward kernel)."""

    @staticmethod
    def forward(ctx, logp, target, alpha, gamma, size_average):
        from . import _hip
        logp = logp.contiguous()
        target = target.reshape(-1).contiguous()
        N, C = logp.shape
        loss = torch.empty((), dtype=torch.float32, device=logp.device)
        coef = torch.empty(N, dtype=torch.float32, device=logp.device)
        want = ctx.needs_input_grad[0]
        dunit = torch.empty(N, C, dtype=torch.float32, device=logp.device) if want else None
        rc = _hip.lib().mmdfn_focal_loss_fwd_grad(_hip.ptr(logp), _hip.ptr(target), _hip.ptr(alpha), _hip.ptr(loss),
                                                  _hip.ptr(coef), _hip.ptr(dunit), N, C, float(gamma),
                                                  1 if size_average else 0, _hip.stream())
        _hip.check(rc, "mmdfn_focal_loss_fwd_grad")
        ctx.save_for_backward(coef, target)
        ctx.dunit = dunit
        ctx.C = C
        return loss

    @staticmethod
    def backward(ctx, dloss):
        from . import _hip
        if ctx.dunit is not None and UNIT_SEED_ACTIVE[0]:
            if getattr(ctx, "dunit_handed_out", False):
                return ctx.dunit.clone(), None, None, None, None     # a second pass (retain_graph): never the same tensor twice
            ctx.dunit_handed_out = True
            return ctx.dunit, None, None, None, None
        coef, target = ctx.saved_tensors
        N = coef.shape[0]
        dlogp = torch.empty(N, ctx.C, dtype=torch.float32, device=coef.device)
        dl = dloss.contiguous().to(torch.float32)
        rc = _hip.lib().mmdfn_focal_loss_bwd(_hip.ptr(coef), _hip.ptr(target), _hip.ptr(dl), _hip.ptr(dlogp), N, ctx.C,
                                             _hip.stream())
        _hip.check(rc, "mmdfn_focal_loss_bwd")
        return dlogp, None, None, None, None


class _FocalLossIgnoreHip(torch.autograd.Function):
    """The fused launches with rows to leave out (csrc/focal_loss.hip, *_ignore): the row count of the mean is taken on
    the device, so a captured step serves batches with different numbers of rows that count."""

    @staticmethod
    def forward(ctx, logp, target, alpha, gamma, size_average, ignore_index):
        from . import _hip
        logp = logp.contiguous()
        target = target.reshape(-1).contiguous()
        N, C = logp.shape
        loss = torch.empty((), dtype=torch.float32, device=logp.device)
        scale = torch.empty(1, dtype=torch.float32, device=logp.device)
        coef = torch.empty(N, dtype=torch.float32, device=logp.device)
        rc = _hip.lib().mmdfn_focal_loss_fwd_ignore(_hip.ptr(logp), _hip.ptr(target), _hip.ptr(alpha), _hip.ptr(loss),
                                                    _hip.ptr(coef), _hip.ptr(scale), N, C, float(gamma),
                                                    1 if size_average else 0, int(ignore_index), _hip.stream())
        _hip.check(rc, "mmdfn_focal_loss_fwd_ignore")
        ctx.save_for_backward(coef, target, scale)
        ctx.C, ctx.ignore = C, int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        from . import _hip
        coef, target, scale = ctx.saved_tensors
        N = coef.shape[0]
        dlogp = torch.empty(N, ctx.C, dtype=torch.float32, device=coef.device)
        dl = dloss.contiguous().to(torch.float32)
        rc = _hip.lib().mmdfn_focal_loss_bwd_ignore(_hip.ptr(coef), _hip.ptr(target), _hip.ptr(dl), _hip.ptr(scale),
                                                    _hip.ptr(dlogp), N, ctx.C, ctx.ignore, _hip.stream())
        _hip.check(rc, "mmdfn_focal_loss_bwd_ignore")
        return dlogp, None, None, None, None, None


class FocalLoss(nn.Module):
    """``ignore_index`` (not in the reference, default None = the reference's behaviour): rows with that label add nothing to
    the loss or the gradient and are not counted by the mean."""

    def __init__(self, gamma=0, alpha=None, size_average=True, ignore_index=None):
        super().__init__()
        self.gamma = gamma
        self.ignore_index = ignore_index
        if isinstance(alpha, (float, int)) and not isinstance(alpha, bool):
            alpha = torch.tensor([alpha, 1 - alpha], dtype=torch.float32)
        elif isinstance(alpha, list):
            alpha = torch.tensor(alpha, dtype=torch.float32)
        self.alpha = alpha
        self.size_average = size_average

    def forward(self, input, target):
        if input.dim() > 2:
            input = input.view(input.size(0), input.size(1), -1).transpose(1, 2).contiguous().view(-1, input.size(1))
        if input.is_cuda and input.dtype == torch.float32 and input.dim() == 2 and target.dtype == torch.int64:
            alpha = self.alpha
            if alpha is not None and alpha.numel() < input.shape[1]:
                # the reference's alpha.gather(0, target) raises for such labels (loss.py:29); never read past the table
                raise IndexError("FocalLoss: alpha has %d entries for %d classes" % (alpha.numel(), input.shape[1]))
            if alpha is not None and (alpha.device != input.device or alpha.dtype != input.dtype):
                alpha = self.alpha = alpha.to(device=input.device, dtype=input.dtype)
            if self.ignore_index is not None:
                return _FocalLossIgnoreHip.apply(input, target, alpha, self.gamma, self.size_average, self.ignore_index)
            return _FocalLossHip.apply(input, target, alpha, self.gamma, self.size_average)
        # host tensors (the loss is plain glue in the reference; the gloo data-parallel test runs it on the CPU)
        if self.ignore_index is not None:
            keep = target.view(-1) != self.ignore_index
            input, target = input[keep], target.view(-1)[keep]
        target = target.view(-1, 1)
        logpt = input.gather(1, target).view(-1)
        pt = logpt.detach().exp()
        if self.alpha is not None:
            if self.alpha.device != input.device or self.alpha.dtype != input.dtype:
                self.alpha = self.alpha.to(device=input.device, dtype=input.dtype)
            logpt = logpt * self.alpha.gather(0, target.view(-1))
        loss = -1 * (1 - pt) ** self.gamma * logpt
        return loss.mean() if self.size_average else loss.sum()
