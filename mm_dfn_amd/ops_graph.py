"""Adjacency build (K5), propagate (K6 / K6'), and the large-launch pointwise stages of the GCN stack.

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import torch

from . import _hip
from .layout import BlockTileAdjacency, DialogueLayout
from .ops_pad import _lay_args, _rows_view
from .ops_wgrad import flush_queued_wgrads_early


def propagate_raw(tiles, cross, H, lay, transpose=False, out=None):
    """out = A . H  (or A^T . H) for block-tile A; H: (M*N, d) fp32 (row-strided views accepted)."""
    _hip.require_cuda(tiles, H)
    H = _rows_view(H, lay.M * lay.N)
    d = H.shape[1]
    if out is None:
        out = torch.empty(H.shape[0], d, dtype=torch.float32, device=H.device)
    rc = _hip.lib().mmdfn_propagate(_hip.ptr(tiles), _hip.ptr(cross), _hip.ptr(H), _hip.ptr(out), *_lay_args(lay),
                                    lay.B, lay.M, lay.N, d, H.stride(0), out.stride(0), lay.max_len,
                                    1 if transpose else 0, _hip.stream())
    _hip.check(rc, "mmdfn_propagate")
    return out


def tile_outer_raw(X, Y, lay, dtiles=None, dcross=None):
    """Gradient of propagate w.r.t. the stored adjacency entries: (dtiles, dcross)."""
    _hip.require_cuda(X, Y)
    X = _rows_view(X, lay.M * lay.N)
    Y = _rows_view(Y, lay.M * lay.N)
    accumulate = dtiles is not None
    if dtiles is None:
        dtiles = torch.empty(lay.tile_elems, dtype=torch.float32, device=X.device)
        dcross = torch.empty(lay.npairs, lay.N, dtype=torch.float32, device=X.device)
    rc = _hip.lib().mmdfn_tile_outer(_hip.ptr(X), _hip.ptr(Y), _hip.ptr(dtiles), _hip.ptr(dcross), *_lay_args(lay),
                                     lay.B, lay.M, lay.N, X.shape[1], X.stride(0), Y.stride(0), lay.max_len,
                                     1 if accumulate else 0, _hip.stream())
    _hip.check(rc, "mmdfn_tile_outer")
    return dtiles, dcross


class _Propagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tiles, cross, H, lay, symmetric):
        ctx.lay = lay
        ctx.symmetric = symmetric
        H = H.contiguous()
        ctx.save_for_backward(tiles, cross, H)
        return propagate_raw(tiles, cross, H, lay)

    @staticmethod
    def backward(ctx, dO):
        tiles, cross, H = ctx.saved_tensors
        lay = ctx.lay
        dO = dO.contiguous()
        dH = dtiles = dcross = None
        if ctx.needs_input_grad[2]:
            dH = propagate_raw(tiles, cross, dO, lay, transpose=not ctx.symmetric)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dtiles, dcross = tile_outer_raw(dO, H, lay)
        return dtiles, dcross, dH, None, None


def propagate(adj, H):
    """hi = A_hat . H  (reference: torch.spmm(adj, input), model_GCN.py:178)."""
    return _Propagate.apply(adj.tiles, adj.cross, H, adj.layout, adj.symmetric)


class _BuildAdjacency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats_in, lay, modal_weight):
        _hip.require_cuda(feats_in)
        feats = feats_in.contiguous()
        M, N, D = feats.shape
        dev = feats.device
        f32 = dict(dtype=torch.float32, device=dev)
        unit = torch.empty_like(feats)
        norm = torch.empty(M, N, **f32)
        cosg = torch.empty(lay.tile_elems, **f32)
        cdot = torch.empty(lay.npairs, N, **f32)
        rdeg = torch.empty(M, N, **f32)
        tiles = torch.empty(lay.tile_elems, **f32)
        cross = torch.empty(lay.npairs, N, **f32)
        rc = _hip.lib().mmdfn_adj_build(_hip.ptr(feats), _hip.ptr(unit), _hip.ptr(norm), _hip.ptr(cosg),
                                        _hip.ptr(cdot), _hip.ptr(rdeg), _hip.ptr(tiles), _hip.ptr(cross),
                                        *_lay_args(lay), lay.B, M, N, D, lay.max_len, float(modal_weight),
                                        _hip.stream())
        _hip.check(rc, "mmdfn_adj_build")
        ctx.lay = lay
        ctx.modal_weight = float(modal_weight)
        ctx.save_for_backward(unit, norm, cosg, cdot, rdeg, tiles, cross)
        ctx.set_materialize_grads(False)
        # the features come back as a third output (an identity): the GCN stack reads THAT, so the features have one
        # consumer and their two gradient paths meet inside this node's backward kernel instead of in an autograd add
        return tiles, cross, feats_in

    @staticmethod
    def backward(ctx, dtiles, dcross, dalias):
        unit, norm, cosg, cdot, rdeg, tiles, cross = ctx.saved_tensors
        if dtiles is None and dcross is None:
            return dalias, None, None
        lay = ctx.lay
        M, N, D = unit.shape
        f32 = dict(dtype=torch.float32, device=unit.device)
        dtiles = torch.zeros(lay.tile_elems, **f32) if dtiles is None else dtiles.contiguous()
        dcross = torch.zeros(lay.npairs, N, **f32) if dcross is None else dcross.contiguous()
        wsym = torch.empty(lay.tile_elems, **f32)
        etile = torch.empty(lay.tile_elems, **f32)
        ecross = torch.empty(lay.npairs, N, **f32)
        ddeg = torch.empty(M, N, **f32)
        dunit = torch.empty_like(unit)
        dfeats = torch.empty_like(unit)
        addend = dalias.contiguous() if dalias is not None else None
        rc = _hip.lib().mmdfn_adj_build_bwd(_hip.ptr(dtiles), _hip.ptr(dcross), _hip.ptr(unit), _hip.ptr(norm),
                                            _hip.ptr(cosg), _hip.ptr(cdot), _hip.ptr(rdeg), _hip.ptr(tiles),
                                            _hip.ptr(cross), _hip.ptr(wsym), _hip.ptr(etile), _hip.ptr(ecross),
                                            _hip.ptr(ddeg), _hip.ptr(dunit), _hip.ptr(dfeats), _hip.ptr(addend),
                                            *_lay_args(lay),
                                            lay.B, M, N, D, lay.max_len, ctx.modal_weight, _hip.stream())
        _hip.check(rc, "mmdfn_adj_build_bwd")
        # the graph part of the backward pass ends here (the encoders' nodes follow): its queued weight gradients leave now
        flush_queued_wgrads_early()
        return dfeats, None, None


def build_adjacency(feats, lengths, modal_weight=1.0):
    """feats: (M, N, D) stacked modality features -> BlockTileAdjacency
    (reference: MM_GCN.create_big_adj, model_mm.py:122-180)."""
    lay = DialogueLayout.get(lengths, feats.shape[0], feats.device)
    if lay.N != feats.shape[1]:
        raise ValueError("sum(dia_len)=%d does not match %d feature rows" % (lay.N, feats.shape[1]))
    if feats.shape[2] % 4:
        raise ValueError("feature width must be a multiple of 4 for the HIP path")
    tiles, cross, feats = _BuildAdjacency.apply(feats, lay, modal_weight)
    return BlockTileAdjacency(lay, tiles, cross, symmetric=True, stacked_feats=feats)


class _PropagateConcat(torch.autograd.Function):
    """S2 = [A_hat . H | h0]  (the GCNII "support" matrix, model_GCN.py:178-180) without a concat pass:
    the propagate kernel writes straight into the left half of S2."""

    @staticmethod
    def forward(ctx, tiles, cross, H, h0, lay, symmetric):
        ctx.lay = lay
        ctx.symmetric = symmetric
        H = H.contiguous()
        d = H.shape[1]
        S2 = torch.empty(H.shape[0], 2 * d, dtype=torch.float32, device=H.device)
        S2[:, d:].copy_(h0)
        propagate_raw(tiles, cross, H, lay, out=S2[:, :d])
        ctx.save_for_backward(tiles, cross, H)
        return S2

    @staticmethod
    def backward(ctx, dS2):
        tiles, cross, H = ctx.saved_tensors
        lay = ctx.lay
        d = H.shape[1]
        dS2 = dS2.contiguous()
        dhi = dS2[:, :d]
        dH = dtiles = dcross = None
        if ctx.needs_input_grad[2]:
            dH = propagate_raw(tiles, cross, dhi, lay, transpose=not ctx.symmetric)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dtiles, dcross = tile_outer_raw(dhi, H, lay)
        return dtiles, dcross, dH, dS2[:, d:], None, None


def propagate_concat(adj, H, h0):
    return _PropagateConcat.apply(adj.tiles, adj.cross, H, h0, adj.layout, adj.symmetric)


class _LstmPointwise(torch.autograd.Function):
    """(h, c) = LSTM-cell gate math on pre-activations G (R, 4H); c_prev None = zero state."""

    @staticmethod
    def forward(ctx, G, c_prev):
        _hip.require_cuda(G)
        G = G.contiguous()
        R, H4 = G.shape
        H = H4 // 4
        if c_prev is not None:
            c_prev = c_prev.contiguous()
        h = torch.empty(R, H, dtype=torch.float32, device=G.device)
        c = torch.empty_like(h)
        rc = _hip.lib().mmdfn_lstm_pointwise_fwd(_hip.ptr(G), _hip.ptr(c_prev), _hip.ptr(h), _hip.ptr(c), R, H,
                                                 _hip.stream())
        _hip.check(rc, "mmdfn_lstm_pointwise_fwd")
        ctx.has_prev = c_prev is not None
        ctx.save_for_backward(G, c, *([c_prev] if c_prev is not None else []))
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        saved = ctx.saved_tensors
        G, c = saved[0], saved[1]
        c_prev = saved[2] if ctx.has_prev else None
        R, H = c.shape
        dh = dh.contiguous() if dh is not None else None
        dc = dc.contiguous() if dc is not None else None
        dG = torch.empty_like(G)
        dcp = torch.empty_like(c)
        rc = _hip.lib().mmdfn_lstm_pointwise_bwd(_hip.ptr(G), _hip.ptr(c_prev), _hip.ptr(c), _hip.ptr(dh),
                                                 _hip.ptr(dc), _hip.ptr(dG), _hip.ptr(dcp), R, H, _hip.stream())
        _hip.check(rc, "mmdfn_lstm_pointwise_bwd")
        return dG, (dcp if ctx.has_prev else None)


def lstm_pointwise(G, c_prev=None):
    return _LstmPointwise.apply(G, c_prev)


class _GcniiCombine(torch.autograd.Function):
    """out = relu(theta P + (1-theta)((1-alpha) hi + alpha h0)) * mask + q, with S2 = [hi | h0]."""

    @staticmethod
    def forward(ctx, P, S2, q, mask, theta, alpha):
        _hip.require_cuda(P, S2)
        P = P.contiguous()
        S2 = S2.contiguous()
        R, d = P.shape
        q_ = q.contiguous() if q is not None else None
        mask_ = mask.contiguous() if mask is not None else None
        out = torch.empty_like(P)
        rc = _hip.lib().mmdfn_gcnii_combine_fwd(_hip.ptr(P), _hip.ptr(S2), _hip.ptr(q_), _hip.ptr(mask_), _hip.ptr(out),
                                                float(theta), float(alpha), R, d, _hip.stream())
        _hip.check(rc, "mmdfn_gcnii_combine_fwd")
        ctx.theta, ctx.alpha, ctx.has_mask, ctx.has_q = float(theta), float(alpha), mask is not None, q is not None
        ctx.save_for_backward(P, S2, *([mask_] if mask is not None else []))
        return out

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        P, S2 = saved[0], saved[1]
        mask = saved[2] if ctx.has_mask else None
        dout = dout.contiguous()
        R, d = P.shape
        dP = torch.empty_like(P)
        dS2 = torch.empty_like(S2)
        rc = _hip.lib().mmdfn_gcnii_combine_bwd(_hip.ptr(P), _hip.ptr(S2), _hip.ptr(mask), _hip.ptr(dout), _hip.ptr(dP),
                                                _hip.ptr(dS2), ctx.theta, ctx.alpha, R, d, _hip.stream())
        _hip.check(rc, "mmdfn_gcnii_combine_bwd")
        return dP, dS2, (dout if ctx.has_q else None), None, None, None


def gcnii_combine(P, S2, q, mask, theta, alpha):
    return _GcniiCombine.apply(P, S2, q, mask, theta, alpha)
