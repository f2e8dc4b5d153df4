"""autograd.Functions over the C-ABI kernels (libmmdfn_hip.so).

Every function here launches hand-written gfx950 kernels on the current HIP
stream.  There is no CPU / eager fallback: CPU tensors raise.
"""
import torch

from . import _hip
from .layout import BlockTileAdjacency, DialogueLayout


def _lay_args(lay):
    return (_hip.ptr(lay.dia_len), _hip.ptr(lay.row_start), _hip.ptr(lay.tile_base))


def propagate_raw(tiles, cross, H, lay, transpose=False):
    """out = A . H  (or A^T . H) for block-tile A; H: (M*N, d) fp32 contiguous."""
    _hip.require_cuda(tiles, H)
    H = H.contiguous()
    if H.dtype != torch.float32 or H.dim() != 2 or H.shape[0] != lay.M * lay.N:
        raise ValueError("propagate expects an fp32 (M*N, d) matrix, got %s %s" % (tuple(H.shape), H.dtype))
    d = H.shape[1]
    out = torch.empty_like(H)
    rc = _hip.lib().mmdfn_propagate(_hip.ptr(tiles), _hip.ptr(cross), _hip.ptr(H), _hip.ptr(out), *_lay_args(lay),
                                    lay.B, lay.M, lay.N, d, lay.max_len, 1 if transpose else 0, _hip.stream())
    _hip.check(rc, "mmdfn_propagate")
    return out


def tile_outer_raw(X, Y, lay, dtiles=None, dcross=None):
    """Gradient of propagate w.r.t. the stored adjacency entries: (dtiles, dcross)."""
    _hip.require_cuda(X, Y)
    X = X.contiguous()
    Y = Y.contiguous()
    accumulate = dtiles is not None
    if dtiles is None:
        dtiles = torch.empty(lay.tile_elems, dtype=torch.float32, device=X.device)
        dcross = torch.empty(lay.npairs, lay.N, dtype=torch.float32, device=X.device)
    rc = _hip.lib().mmdfn_tile_outer(_hip.ptr(X), _hip.ptr(Y), _hip.ptr(dtiles), _hip.ptr(dcross), *_lay_args(lay),
                                     lay.B, lay.M, lay.N, X.shape[1], lay.max_len, 1 if accumulate else 0,
                                     _hip.stream())
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
    def forward(ctx, feats, lay, modal_weight):
        _hip.require_cuda(feats)
        feats = feats.contiguous()
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
        return tiles, cross

    @staticmethod
    def backward(ctx, dtiles, dcross):
        unit, norm, cosg, cdot, rdeg, tiles, cross = ctx.saved_tensors
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
        rc = _hip.lib().mmdfn_adj_build_bwd(_hip.ptr(dtiles), _hip.ptr(dcross), _hip.ptr(unit), _hip.ptr(norm),
                                            _hip.ptr(cosg), _hip.ptr(cdot), _hip.ptr(rdeg), _hip.ptr(tiles),
                                            _hip.ptr(cross), _hip.ptr(wsym), _hip.ptr(etile), _hip.ptr(ecross),
                                            _hip.ptr(ddeg), _hip.ptr(dunit), _hip.ptr(dfeats), *_lay_args(lay),
                                            lay.B, M, N, D, lay.max_len, ctx.modal_weight, _hip.stream())
        _hip.check(rc, "mmdfn_adj_build_bwd")
        return dfeats, None, None


def build_adjacency(feats, lengths, modal_weight=1.0):
    """feats: (M, N, D) stacked modality features -> BlockTileAdjacency
    (reference: MM_GCN.create_big_adj, model_mm.py:122-180)."""
    lay = DialogueLayout.get(lengths, feats.shape[0], feats.device)
    if lay.N != feats.shape[1]:
        raise ValueError("sum(dia_len)=%d does not match %d feature rows" % (lay.N, feats.shape[1]))
    if feats.shape[2] % 4:
        raise ValueError("feature width must be a multiple of 4 for the HIP path")
    tiles, cross = _BuildAdjacency.apply(feats, lay, modal_weight)
    return BlockTileAdjacency(lay, tiles, cross, symmetric=True, stacked_feats=feats)
