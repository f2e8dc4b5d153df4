"""Classifier head (K9).

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import torch

from . import _hip
from .ops_flags import keep_flags, keep_scale
from .ops_linear import linear
from .ops_wgrad import queue_slab_reduce, slab_reduce_queueable


class _Head(torch.autograd.Function):
    """log_softmax(relu(F (.) mask * mscale) W^T + b): the classifier head of model.py:1328-1337 as one launch each way
    (csrc/head.hip); mask = 0 / 1 keep flags of the head dropout or None.  ``Fm``: (N, W), or the (M, N, Wm) output of
    the graph stack standing for cat([Fm[0], .., Fm[M-1]], -1) (model_mm.py:113-117): the kernels read the blocks in
    place and write dF in the same layout, so neither the concatenation nor its backward exists."""

    @staticmethod
    def forward(ctx, Fm, mask, mscale, weight, bias):
        _hip.require_cuda(Fm, weight)
        _hip.require_f32(Fm, mask, weight, bias)
        if Fm.dim() == 3:
            Fm = Fm.contiguous()
            N, split = Fm.shape[1], Fm.shape[2]
            Wd, ldf = Fm.shape[0] * split, split
        else:
            if Fm.stride(1) != 1 or Fm.stride(0) % 4 or Fm.data_ptr() % 16:
                Fm = Fm.contiguous()
            (N, Wd), split, ldf = Fm.shape, 0, Fm.stride(0)
        C = weight.shape[0]
        ctx.refs = (weight, bias)          # the parameters themselves (slab_reduce_queueable looks at .is_leaf / hooks)
        weight, bias = weight.contiguous(), bias.contiguous()
        mask = mask.contiguous() if mask is not None else None
        logp = torch.empty(N, C, dtype=torch.float32, device=Fm.device)
        rc = _hip.lib().mmdfn_head_fwd(_hip.ptr(Fm), _hip.ptr(mask), _hip.ptr(weight), _hip.ptr(bias), _hip.ptr(logp), N, Wd, C,
                                       ldf, split, float(mscale), _hip.stream())
        _hip.check(rc, "mmdfn_head_fwd")
        ctx.mscale = float(mscale)
        ctx.dims = (N, Wd, split, ldf)
        ctx.save_for_backward(Fm, mask, weight, logp)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        Fm, mask, weight, logp = ctx.saved_tensors
        N, Wd, split, ldf = ctx.dims
        C = weight.shape[0]
        dlogp = dlogp.contiguous()
        lib = _hip.lib()
        dF = torch.empty(Fm.shape, dtype=torch.float32, device=Fm.device)
        ws = torch.empty(int(lib.mmdfn_head_bwd_workspace(Wd, C)), dtype=torch.float32, device=Fm.device)
        pw, pb = ctx.refs
        if (ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and tuple(pw.shape) == (C, Wd) and pw.is_contiguous()
                and slab_reduce_queueable(pw, [pb])):
            # dW / db stay slab stacks: the reduction launch of the step's weight-gradient batch sums them (no launch of their own)
            rc = lib.mmdfn_head_bwd_partial(_hip.ptr(dlogp), _hip.ptr(logp), _hip.ptr(Fm), _hip.ptr(mask), _hip.ptr(weight),
                                            _hip.ptr(dF), _hip.ptr(ws), N, Wd, C, ldf, split if split else Wd, split, ctx.mscale,
                                            _hip.stream())
            _hip.check(rc, "mmdfn_head_bwd_partial")
            G = int(lib.mmdfn_head_bwd_groups())
            queue_slab_reduce(ws[:G * C * Wd], ws[G * C * Wd:], G, C, Wd, weight=pw, biases=[pb])
            return dF, None, None, None, None
        dW = torch.empty(C, Wd, dtype=torch.float32, device=Fm.device)
        db = torch.empty(C, dtype=torch.float32, device=Fm.device)
        rc = lib.mmdfn_head_bwd(_hip.ptr(dlogp), _hip.ptr(logp), _hip.ptr(Fm), _hip.ptr(mask), _hip.ptr(weight), _hip.ptr(dF),
                                _hip.ptr(dW), _hip.ptr(db), _hip.ptr(ws), N, Wd, C, ldf, split if split else Wd, split,
                                ctx.mscale, _hip.stream())
        _hip.check(rc, "mmdfn_head_bwd")
        return dF, None, None, dW, db


def _head_width(Fm):
    return Fm.shape[0] * Fm.shape[2] if Fm.dim() == 3 else Fm.shape[1]


def head_supported(Fm, weight):
    return (Fm.is_cuda and Fm.dtype == torch.float32 and Fm.dim() in (2, 3) and weight.shape[0] <= 8 and Fm.shape[-1] % 4 == 0
            and weight.shape[0] * _head_width(Fm) * 4 <= 150 * 1024)


def head(Fm, weight, bias, p=0.0, training=False):
    """log_softmax(Linear(relu(dropout(Fm)))) (reference model.py:1328-1337).  ``Fm``: the fused features (N, W), or
    the stacked graph output (M, N, Wm) standing for its column-wise concatenation (N, M Wm).  Wide heads (> 8 classes)
    take the library composition."""
    if not head_supported(Fm, weight) or bias is None:
        if Fm.dim() == 3:
            Fm = Fm.permute(1, 0, 2).reshape(Fm.shape[1], -1)
        z = torch.relu(torch.nn.functional.dropout(Fm, p, training))
        return torch.log_softmax(linear(z, weight, bias), 1)
    mask, mscale = None, 1.0
    if training and p > 0:
        N = Fm.shape[1] if Fm.dim() == 3 else Fm.shape[0]
        mask = keep_flags(N * _head_width(Fm), p, Fm.device).view(N, _head_width(Fm))
        mscale = keep_scale(p)
    return _Head.apply(Fm, mask, mscale, weight, bias)
