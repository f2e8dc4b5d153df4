"""Kernels of the fusion modules (MFN, MMGatedAttention).

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import torch

from . import _hip


class _SoftmaxScale(torch.autograd.Function):
    """out = softmax(z, dim=1) * c (MFN attention, model_fusion.py:96-97)."""

    @staticmethod
    def forward(ctx, z, c):
        _hip.require_cuda(z, c)
        z, c = z.contiguous(), c.contiguous()
        att, out = torch.empty_like(z), torch.empty_like(z)
        _hip.check(_hip.lib().mmdfn_softmax_scale_fwd(_hip.ptr(z), _hip.ptr(c), _hip.ptr(att), _hip.ptr(out), z.shape[0],
                                                      z.shape[1], _hip.stream()), "mmdfn_softmax_scale_fwd")
        ctx.save_for_backward(att, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        att, c = ctx.saved_tensors
        dout = dout.contiguous()
        dz, dc = torch.empty_like(att), torch.empty_like(att)
        _hip.check(_hip.lib().mmdfn_softmax_scale_bwd(_hip.ptr(att), _hip.ptr(c), _hip.ptr(dout), _hip.ptr(dz), _hip.ptr(dc),
                                                      att.shape[0], att.shape[1], _hip.stream()), "mmdfn_softmax_scale_bwd")
        return dz, dc


def softmax_scale(z, c):
    return _SoftmaxScale.apply(z, c)


class _MfnMem(torch.autograd.Function):
    """mem' = sigmoid(v1) mem + sigmoid(v2) tanh(u)  (model_fusion.py:98-102)."""

    @staticmethod
    def forward(ctx, u, v1, v2, mem):
        _hip.require_cuda(u, v1, v2, mem)
        u, v1, v2, mem = u.contiguous(), v1.contiguous(), v2.contiguous(), mem.contiguous()
        out = torch.empty_like(mem)
        saved = torch.empty(3, mem.numel(), dtype=mem.dtype, device=mem.device)
        _hip.check(_hip.lib().mmdfn_mfn_mem_fwd(_hip.ptr(u), _hip.ptr(v1), _hip.ptr(v2), _hip.ptr(mem), _hip.ptr(out),
                                                _hip.ptr(saved), mem.numel(), _hip.stream()), "mmdfn_mfn_mem_fwd")
        ctx.save_for_backward(saved, mem)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved, mem = ctx.saved_tensors
        dout = dout.contiguous()
        du, dv1, dv2, dmem = (torch.empty_like(mem) for _ in range(4))
        _hip.check(_hip.lib().mmdfn_mfn_mem_bwd(_hip.ptr(saved), _hip.ptr(mem), _hip.ptr(dout), _hip.ptr(du), _hip.ptr(dv1),
                                                _hip.ptr(dv2), _hip.ptr(dmem), mem.numel(), _hip.stream()), "mmdfn_mfn_mem_bwd")
        return du, dv1, dv2, dmem


def mfn_mem(u, v1, v2, mem):
    return _MfnMem.apply(u, v1, v2, mem)


class _GatedPair(torch.autograd.Function):
    """h = z tanh(p_m) + (1 - z) tanh(p_n), z = sigmoid(w . [x_m | x_n | x_m * x_n] + b)  (model.py:766-781); w: (1, 3D)."""

    @staticmethod
    def forward(ctx, xm, xn, pm, pn, w, b):
        _hip.require_cuda(xm, xn, pm, pn, w, b)
        xm, xn, pm, pn, w = xm.contiguous(), xn.contiguous(), pm.contiguous(), pn.contiguous(), w.contiguous()
        R, D = xm.shape
        C = pm.shape[1]
        out = torch.empty_like(pm)
        zs = torch.empty(R, dtype=xm.dtype, device=xm.device)
        _hip.check(_hip.lib().mmdfn_gated_pair_fwd(_hip.ptr(xm), _hip.ptr(xn), _hip.ptr(w), _hip.ptr(b), _hip.ptr(pm), _hip.ptr(pn),
                                                   _hip.ptr(out), _hip.ptr(zs), R, D, C, _hip.stream()), "mmdfn_gated_pair_fwd")
        ctx.save_for_backward(xm, xn, pm, pn, w, zs)
        return out

    @staticmethod
    def backward(ctx, dout):
        xm, xn, pm, pn, w, zs = ctx.saved_tensors
        dout = dout.contiguous()
        R, D = xm.shape
        C = pm.shape[1]
        dxm, dxn, dpm, dpn = torch.empty_like(xm), torch.empty_like(xn), torch.empty_like(pm), torch.empty_like(pn)
        dpre = torch.empty(R, dtype=xm.dtype, device=xm.device)
        lib = _hip.lib()
        _hip.check(lib.mmdfn_gated_pair_bwd(_hip.ptr(xm), _hip.ptr(xn), _hip.ptr(w), _hip.ptr(pm), _hip.ptr(pn), _hip.ptr(zs),
                                            _hip.ptr(dout), _hip.ptr(dxm), _hip.ptr(dxn), _hip.ptr(dpm), _hip.ptr(dpn),
                                            _hip.ptr(dpre), R, D, C, _hip.stream()), "mmdfn_gated_pair_bwd")
        dwb = torch.empty(3 * D + 1, dtype=xm.dtype, device=xm.device)
        _hip.check(lib.mmdfn_rowscale_colsum(_hip.ptr(dpre), _hip.ptr(xm), _hip.ptr(xn), _hip.ptr(dwb), R, D, _hip.stream()),
                   "mmdfn_rowscale_colsum")
        return dxm, dxn, dpm, dpn, dwb[:3 * D].view(1, 3 * D), dwb[3 * D:].view(1)


def gated_pair(xm, xn, pm, pn, w, b):
    return _GatedPair.apply(xm, xn, pm, pn, w, b)
