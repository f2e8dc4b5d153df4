"""Training / evaluation pass: counterpart of ``train_or_eval_graph_model``
(reference run_train_erc.py:149-238), same argument list and 8-tuple result.

Kept from the reference: per-pass reseed (:164), batch tuple order ``textf,
visuf, acouf, qmask, umask, label`` (:169), lengths derived from umask (:194),
model call (:197), dialogue-major label flatten (:201), loss/backward/step
(:202-212), sklearn metrics (:229-236).  Changed for the device: lengths are
read back with ONE host sync per batch instead of B, and loss / prediction
syncs are deferred to the end of the pass.
"""
import random

import numpy as np
import torch

from . import ops


def seed_everything(seed=2021):
    """run_train_erc.py:19-26."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def lengths_from_umask(umask):
    """Index of the last 1 in each row + 1 (run_train_erc.py:194), one device->host sync."""
    L = umask.shape[1]
    pos = torch.arange(1, L + 1, device=umask.device).unsqueeze(0)
    return ((umask == 1).to(torch.int64) * pos).max(1).values.tolist()


def flatten_labels(label, lengths):
    """run_train_erc.py:201 without B slice ops: mask-select in dialogue-major order."""
    L = label.shape[1]
    keep = torch.arange(L, device=label.device).unsqueeze(0) < torch.as_tensor(lengths, device=label.device).unsqueeze(1)
    return label[keep]


def train_or_eval_graph_model(model, loss_f, dataloader, epoch=0, train_flag=False, optimizer=None, cuda_flag=False,
                              modals=None, target_names=None, test_label=False, tensorboard=False, seed=2021,
                              step_hook=None):
    losses, preds, labels = [], [], []
    assert not train_flag or optimizer is not None
    model.train() if train_flag else model.eval()
    seed_everything(seed)
    vids = []
    for data in dataloader:
        if train_flag:
            optimizer.zero_grad()
        textf, visuf, acouf, qmask, umask, label = [d.cuda() for d in data[:6]] if cuda_flag else data[:6]
        lengths = lengths_from_umask(umask)
        log_prob, e_i, e_n, e_t, e_l = model(textf, qmask, umask, lengths, acouf, visuf, test_label)
        flat = flatten_labels(label, lengths)
        loss = loss_f(log_prob, flat)
        preds.append(torch.argmax(log_prob, 1))
        labels.append(flat)
        losses.append(loss.detach())
        if train_flag:
            loss.backward()
            ops.join_weight_grads()       # weight gradients may have been computed on the side stream
            if step_hook is not None:
                step_hook(model)          # e.g. data-parallel gradient all-reduce
            optimizer.step()
        if len(data) > 6:
            vids = data[6]
    if not preds:
        return [], [], float('nan'), float('nan'), [], [], float('nan'), []
    preds = torch.cat(preds).cpu().numpy()
    labels = torch.cat(labels).cpu().numpy()
    losses = torch.stack(losses).cpu().numpy()
    avg_loss = round(float(np.sum(losses)) / len(losses), 4)
    from sklearn import metrics
    from sklearn.metrics import accuracy_score, f1_score
    avg_accuracy = round(accuracy_score(labels, preds) * 100, 2)
    avg_fscore = round(f1_score(labels, preds, average='weighted') * 100, 2)
    all_each, all_acc = "", ["ACC"]
    if target_names is not None:
        all_each = metrics.classification_report(labels, preds, target_names=target_names, digits=4,
                                                 labels=list(range(len(target_names))), zero_division=0)
        for i, name in enumerate(target_names):
            sel = labels == i
            acc = accuracy_score(labels[sel], preds[sel]) if sel.any() else float('nan')
            all_acc.append("{}: {:.4f}".format(name, acc))
    return all_each, all_acc, avg_loss, avg_accuracy, labels, preds, avg_fscore, [np.array(vids), losses]
