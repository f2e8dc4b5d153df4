"""Data parallelism over dialogues: one process per GPU, ONE flat-bucket gradient all-reduce per step.

The reference is single-process; dialogues never interact (GRUs are per sequence,
the adjacency is block-diagonal per dialogue, the loss is a mean over utterances), so the batch
shards with no data-path collective.  The only exchange is the gradient sum: all live-parameter
gradients are views into ONE contiguous fp32 buffer (~4-5 MB) which is all-reduced in a single
RCCL call over xGMI (backend "nccl" on ROCm) -- one large message instead of ~50 small ones,
which is what a point-to-point xGMI ring wants.  Parameters the MM-DFN configuration never
reaches (38 of 86 tensors get no gradient) are excluded, as Adam skips them in the reference.

Exactness vs one big batch: FocalLoss is a mean over the GLOBAL utterance count, so each rank
scales its local mean loss by n_local * world / n_global before backward and the bucket is
averaged (sum / world).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend=backend)


def all_reduce_scalar(value, device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())


def shard_dialogues(lengths, world, rank):
    """Balance dialogues over ranks by sum(L_i^2) (adjacency cost), greedy longest-first; returns indices."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    load = [0] * world
    bins = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        bins[r].append(i)
        load[r] += lengths[i] * lengths[i]
    return sorted(bins[rank])


def bucket_order(model, live):
    """Flat-buffer order of the live parameters: model order, except that every ``weight_ih_l*_reverse`` / ``bias_ih_l*_reverse`` directly
    follows its forward twin -- the fused GRU path reads the two as one stacked (600, K) operand (gru._stacked_view), and FlatAdam
    lays the parameters out in this order."""
    names = {id(p): n for n, p in model.named_parameters()}
    by_name = {names[id(p)]: p for p in live}
    out, placed = [], set()
    for p in live:
        if id(p) in placed:
            continue
        out.append(p)
        placed.add(id(p))
        n = names[id(p)]
        if (".weight_ih_l" in "." + n or ".bias_ih_l" in "." + n) and not n.endswith("_reverse"):
            twin = by_name.get(n + "_reverse")
            if twin is not None and id(twin) not in placed:
                out.append(twin)
                placed.add(id(twin))
    return out


class GradientBucket:
    """Flat fp32 gradient bucket over the parameters that receive gradients.

    ``flatten()`` packs all live gradients into ONE contiguous buffer with a single multi-tensor copy
    (torch.cat -> one or two launches instead of one add/copy per parameter) and re-points every ``.grad``
    at its slice, so the all-reduce result is what the optimizer reads.  Backward passes therefore always
    run with ``.grad = None`` (autograd hands over its buffers, no accumulation kernels)."""

    def __init__(self, model, average=True):
        self.model = model
        self.average = average
        self.flat = None
        self.params = None

    def flatten(self):
        live = [p for p in self.model.parameters() if p.requires_grad and p.grad is not None]
        if self.params is None:
            self.params = bucket_order(self.model, live)
            self._ids = [id(p) for p in live]
        elif [id(p) for p in live] != self._ids:
            # the bucket layout is frozen at the first step (flat optimizer state and all-reduce offsets depend on
            # it); a parameter that starts / stops receiving a gradient later would silently never be reduced or
            # updated -- or crash in the pack.  The reference's Adam skips grad-less parameters step by step
            # (run_train_erc.py:512), so a changing live set needs a new bucket (and optimizer state), not a patch.
            names = {id(p): n for n, p in self.model.named_parameters()}
            now, was = set(id(p) for p in live), set(self._ids)
            raise RuntimeError("GradientBucket: the set of parameters receiving gradients changed since the first step "
                               "(new: %s; missing: %s)" % (sorted(names[i] for i in now - was),
                                                           sorted(names[i] for i in was - now)))
        grads = [p.grad.reshape(-1) for p in self.params]
        if self.flat is None:
            self.flat = torch.cat(grads)
        else:
            torch.cat(grads, out=self.flat)
        self.attach_views()
        return self.flat

    def attach_views(self):
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def reduce_flat(self):
        dist.all_reduce(self.flat)
        if self.average:
            self.flat.div_(dist.get_world_size())
        return self.flat

    def all_reduce(self):
        """Eager step: pack, all-reduce, leave .grad pointing into the reduced bucket."""
        self.flatten()
        return self.reduce_flat()
