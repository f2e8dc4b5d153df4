"""FlatAdam: the reference's optimizer (torch.optim.Adam(lr, weight_decay=l2), run_train_erc.py:512) as ONE
fused HIP launch per step over flat buffers.

All parameters that receive gradients are re-pointed at slices of one contiguous fp32 buffer and their
gradients are packed into a matching flat buffer (the same bucket the data-parallel all-reduce uses), so the
update is a single elementwise kernel (csrc/optimizer.hip).  Parameters the MM-DFN configuration never reaches
get no gradient and are left untouched, exactly like torch.optim.Adam skips ``grad is None``.
"""
import torch

from . import _hip
from .distributed import GradientBucket, register_slots, slot_pieces, slot_size, slot_view


class FlatAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, bucket=None):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.bucket = bucket if bucket is not None else GradientBucket(model, average=True)
        self.t = 0
        self.flat_p = self.m = self.v = None
        # torch.optim-style handle for LR schedulers: ``step`` reads lr / weight_decay from here
        self.param_groups = [dict(params=[p for p in model.parameters() if p.requires_grad], lr=self.lr,
                                  betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)]

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad(set_to_none=True)

    def _materialise(self):
        params = self.bucket.params
        # one slot per parameter (16-byte aligned starts; a row-padded block for weights of odd contraction width,
        # distributed.slot_size): the same layout as the gradient bucket, so the update is one elementwise pass
        flat = torch.cat([piece for p in params for piece in slot_pieces(p.detach(), p)])
        register_slots(flat, params)
        off = 0
        for p in params:
            p.data = slot_view(flat, off, p)
            p._mmdfn_flat = True       # this storage layout is owned here: nobody may re-point the parameter
            off += slot_size(p)
        self.flat_p = flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)

    @torch.no_grad()
    def step(self, grads_already_flat=False):
        """Pack gradients (unless the caller already did, e.g. after the data-parallel all-reduce) and update."""
        if not grads_already_flat:
            # (single process: nothing reads the gradients as views of the bucket, so they are only packed)
            self.bucket.flatten(attach=self.flat_p is None)
        g = self.bucket.flat
        _hip.require_cuda(g)
        if self.flat_p is None:
            self._materialise()
        self.t += 1
        grp = self.param_groups[0]
        rc = _hip.lib().mmdfn_adam_step(_hip.ptr(self.flat_p), _hip.ptr(g), _hip.ptr(self.m), _hip.ptr(self.v),
                                        g.numel(), float(grp["lr"]), self.betas[0], self.betas[1], self.eps,
                                        float(grp["weight_decay"]), self.t, _hip.stream())
        _hip.check(rc, "mmdfn_adam_step")
        # the kernel wrote the parameters behind autograd's version counters: piece planes cut from them are stale now
        from . import ops
        ops.invalidate_planes()

    # ---- checkpointing: per-parameter moments under the parameter NAMES (layout-independent, loads into a bucket
    # whose flat order differs), plus the step count and the hyper-parameters
    def state_dict(self):
        names = {id(p): n for n, p in self.model.named_parameters()}
        state = {}
        if self.flat_p is not None:
            off = 0
            for p in self.bucket.params:
                state[names[id(p)]] = dict(exp_avg=slot_view(self.m, off, p).contiguous().clone(),
                                           exp_avg_sq=slot_view(self.v, off, p).contiguous().clone())
                off += slot_size(p)
        grp = self.param_groups[0]
        return dict(step=self.t, lr=float(grp["lr"]), betas=self.betas, eps=self.eps,
                    weight_decay=float(grp["weight_decay"]), state=state)

    def load_state_dict(self, sd):
        """Needs the bucket layout, i.e. call after one backward pass (or pass a bucket that has been flattened)."""
        self.t = int(sd["step"])
        self.betas, self.eps = (float(sd["betas"][0]), float(sd["betas"][1])), float(sd["eps"])
        self.param_groups[0]["lr"] = self.lr = float(sd["lr"])
        self.param_groups[0]["weight_decay"] = self.weight_decay = float(sd["weight_decay"])
        if not sd["state"]:
            return
        if self.bucket.params is None:
            raise RuntimeError("FlatAdam.load_state_dict: run one backward pass first (the flat layout is the set of "
                               "parameters that receive gradients)")
        if self.flat_p is None:
            self._materialise()
        names = {id(p): n for n, p in self.model.named_parameters()}
        off = 0
        for p in self.bucket.params:
            st = sd["state"][names[id(p)]]
            slot_view(self.m, off, p).copy_(st["exp_avg"].view_as(p))
            slot_view(self.v, off, p).copy_(st["exp_avg_sq"].view_as(p))
            off += slot_size(p)
