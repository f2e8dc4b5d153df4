"""autograd.Functions over the C-ABI kernels (libmmdfn_hip.so).

Every function here launches hand-written gfx950 kernels on the current HIP
stream.  There is no CPU / eager fallback: CPU tensors raise.
"""
import ctypes

import torch

from . import _hip
from .layout import BlockTileAdjacency, DialogueLayout


def _lay_args(lay):
    return (_hip.ptr(lay.dia_len), _hip.ptr(lay.row_start), _hip.ptr(lay.tile_base))


def _rows_view(t, rows):
    """(rows, d) fp32 tensor whose rows are contiguous (a column slice of a wider matrix is fine)."""
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[0] != rows:
        raise ValueError("expected an fp32 (%d, d) matrix, got %s %s" % (rows, tuple(t.shape), t.dtype))
    if t.stride(1) != 1 or t.stride(0) % 4 or t.stride(0) < t.shape[1] or t.data_ptr() % 16:
        t = t.contiguous()
    return t



# ---------------------------------------------------------------------------------------------------
# Contraction widths that are not multiples of 4 (the reference's own feature widths: 1582-d IS10 audio, 342-d denseface,
# run_train_erc.py:359-362).  The MFMA kernels fetch operand rows in 16-byte units, so such a layer runs as the layer of
# the next multiple of 4 whose extra input features are exactly zero: the rows of both operands live in storage padded to
# that width (pad columns zero), and every kernel -- projection, weight gradient, optimizer -- sees the padded width.  The
# result is bit-identical to the unpadded contraction (the extra terms are 0 * 0) and no step launches anything extra:
#   * the PARAMETER keeps its (N, K) shape and state_dict entry but is a row-strided view of an (N, Kp) buffer
#     (ensure_row_padded re-points .data once, like gru._stacked_view does for the GRU weight pairs); its gradient is the
#     same kind of view; FlatAdam / GradientBucket give such a parameter an N * Kp slot;
#   * FEATURES arrive padded from whoever stages them (data.DevicePrefetcher's pinned buffer, the static input buffers of
#     train.StepGraphCache, synthetic.make_batch); a plain contiguous (..., K) tensor handed to the modules is copied into
#     a padded buffer first (pad_rows: the one place where an odd width costs a launch).
# Only buffers registered here are trusted to have zero pad columns.
# ---------------------------------------------------------------------------------------------------
_PAD_REG = {}      # storage data_ptr -> [(weakref to the owning (rows, Kp) tensor, storage offset, rows, Kp)]


def pad4(n):
    return (int(n) + 3) & ~3


# positions of the FEATURE tensors in the reference's batch tuple (textf, visuf, acouf, qmask, umask, label), run_train_erc.py:169
FEATURE_SLOTS = (0, 1, 2)


def is_odd_feature_tensor(t, is_feature):
    """An (L, B, D) fp32 FEATURE tensor whose width is not a multiple of 4 (1582-d audio, 342-d visual): what the data
    pipeline stages row-padded.  ``is_feature`` is the tensor's ROLE, stated by the caller (its slot in the batch tuple is in
    FEATURE_SLOTS): the speaker mask (L, B, P) has the same rank and dtype and is never padded whatever P is (a width
    heuristic took a 12-d feature stream for a mask and a 17-speaker mask for features)."""
    return bool(is_feature) and t.dim() == 3 and t.dtype == torch.float32 and t.shape[-1] % 4 != 0


def register_row_padded(base, region=None, K=None):
    """``base``: a contiguous (..., Kp) fp32 tensor whose columns past the logical width ``K`` are zero and stay zero (nobody
    writes them).  Views of its leading K columns are recognised as row-padded operands WHILE THE ``base`` OBJECT IS ALIVE
    (the registry holds a weak reference: whoever stages the buffer keeps it).  ``region = (storage offset, rows, Kp)``
    registers a padded block inside a larger flat buffer ``base`` (FlatAdam's parameter slots).  ``K``: the logical width; a
    view of FEWER columns with the same padded width (x[:, :K-1]) is then NOT taken for a zero-padded operand (its column
    K-1 holds data) -- None accepts any width that pads to Kp (callers that do not know K)."""
    import weakref
    if region is None:
        b2 = base.view(-1, base.shape[-1])
        region = (b2.storage_offset(), b2.shape[0], b2.shape[1])
    region = tuple(int(v) for v in region)[:3] + (None if K is None else int(K),)
    key = base.untyped_storage().data_ptr()
    live = [e for e in _PAD_REG.get(key, []) if e[0]() is not None and (e[0]() is not base or e[1:] != region)]
    live.append((weakref.ref(base),) + region)
    _PAD_REG[key] = live
    if len(_PAD_REG) > 4096:                      # stale keys of freed buffers
        for k in [k for k, v in _PAD_REG.items() if all(e[0]() is None for e in v)]:
            del _PAD_REG[k]
    return base


def padded_zeros(shape, device, keep=None):
    """Logical (..., K) view of a fresh zero buffer whose last dimension is padded to a multiple of 4.  The caller must keep
    the returned BASE alive for as long as views of it are used (``keep``: a list it is appended to); returns the view."""
    *lead, K = shape
    Kp = pad4(K)
    base = torch.zeros(*lead, Kp, dtype=torch.float32, device=device)
    if Kp == K:
        return base
    register_row_padded(base, K=K)
    if keep is not None:
        keep.append(base)
    view = base[..., :K]
    view._mmdfn_padbase = base          # (keeps the base alive while this particular view object lives)
    return view


def pad_rows(x, keep=None):
    """A row-padded copy of ``x`` (..., K) (one zero fill + one copy): the generic entry for features that do not come
    from a padded staging buffer."""
    v = padded_zeros(tuple(x.shape), x.device, keep)
    v.copy_(x)
    return v


def row_padded_view(x2):
    """(R, Kp) view of a registered row-padded operand ``x2`` (R, K), or None."""
    if x2.dim() != 2 or x2.dtype != torch.float32 or (x2.shape[1] > 1 and x2.stride(1) != 1):
        return None
    R, K = x2.shape
    Kp = pad4(K)
    ents = _PAD_REG.get(x2.untyped_storage().data_ptr())
    if not ents or x2.data_ptr() % 16:
        return None
    for ref, off, rows, width, logical in ents:
        if ref() is None or width != Kp or (logical is not None and logical != K):
            continue
        rel = x2.storage_offset() - off
        if rel < 0 or rel % Kp or rel // Kp + R > rows or (R > 1 and x2.stride(0) != Kp):
            continue
        return x2.as_strided((R, Kp), (Kp, 1))
    return None


def row_operand(x2, keep=None):
    """The (R, K') operand the kernels contract over: ``x2`` itself when K % 4 == 0, else its zero-padded form (the
    registered view, or a padded copy)."""
    if x2.shape[1] % 4 == 0:
        return x2
    v = row_padded_view(x2)
    if v is None:
        v = row_padded_view(pad_rows(x2, keep))
    return v


def ensure_row_padded(p):
    """Parameter (N, K) with K % 4 != 0: re-point ``p.data`` ONCE at the leading columns of a zero-padded (N, Kp) buffer
    (same values, same shape, same state_dict entry).  Parameters laid out by FlatAdam already are."""
    if p.dim() != 2 or p.shape[1] % 4 == 0 or row_padded_view(p.data) is not None:
        return p
    if getattr(p, "_mmdfn_flat", False):
        raise _hip.HipLibraryError("a flat-laid-out parameter lost its row padding")
    with torch.no_grad():
        base = torch.zeros(p.shape[0], pad4(p.shape[1]), dtype=p.dtype, device=p.device)
        base[:, :p.shape[1]].copy_(p.data)
        register_row_padded(base, K=p.shape[1])
        p.data = base[:, :p.shape[1]]
        p._mmdfn_padbase = base
        if p.grad is not None:
            p.grad = None if not p.grad.any() else padded_grad_like(p, p.grad)
    return p


def padded_grad_like(p, g=None, zero=True):
    """Gradient tensor for a row-padded parameter: (N, K) view of an (N, Kp) buffer -- zero-filled (optionally holding
    ``g``), or uninitialised when the caller's kernel writes every one of the Kp columns (the weight-gradient batch does:
    the pad columns come out as dY^T . 0)."""
    base = (torch.zeros if zero else torch.empty)(p.shape[0], pad4(p.shape[1]), dtype=torch.float32, device=p.device)
    register_row_padded(base, K=p.shape[1])
    view = base[:, :p.shape[1]]
    view._mmdfn_padbase = base
    if g is not None:
        view.copy_(g)
    return view


def weight_operand(w):
    """(N, K') form of a dense layer's weight (N, K): the parameter itself when K % 4 == 0, else the padded view of its
    storage (leaf parameters are re-laid once, ensure_row_padded; anything else is copied)."""
    if w.shape[1] % 4 == 0:
        return w
    v = row_padded_view(w.detach() if w.requires_grad else w)
    if v is not None:
        return v
    if w.is_leaf and w.requires_grad and not getattr(w, "_mmdfn_flat", False):
        ensure_row_padded(w)
        return row_padded_view(w.detach())
    return row_padded_view(pad_rows(w.detach()))


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


class _PartyGather(torch.autograd.Function):
    """(X_0..X_{Mn-1} each (L,B,H) -- or ONE stacked (Mn,L,B,H) tensor --, qmask[, bias]) -> S (L, Mn*B*P, H) (+ bias on
    every row); also returns rank (L,B,P) int32 (no grad).  With ``passthrough`` the inputs come back as extra outputs
    (identities): a caller that also needs X_m elsewhere (the combine stage adds the party encoding onto it) uses the
    returned alias, so X_m has ONE consumer and its two gradient contributions meet inside the backward kernel instead of
    in an autograd accumulation launch per modality."""

    @staticmethod
    def forward(ctx, qmask, bias, passthrough, *Xs):
        _hip.require_cuda(qmask, *Xs)
        stacked = len(Xs) == 1 and Xs[0].dim() == 4
        if stacked:
            X4 = Xs[0].contiguous()
            mods = [X4[m] for m in range(X4.shape[0])]
        else:
            mods = [x.contiguous() for x in Xs]
        qmask = qmask.contiguous()
        L, B, P = qmask.shape
        H = mods[0].shape[-1]
        Mn = len(mods)
        bias_ = bias.contiguous() if bias is not None else None
        S = torch.empty(L, Mn * B * P, H, dtype=torch.float32, device=qmask.device)
        rank = torch.empty(L, B, P, dtype=torch.int32, device=qmask.device)
        rc = _hip.lib().mmdfn_party_gather(Mn, _hip.ptr_array(mods), _hip.ptr(qmask), _hip.ptr(bias_), _hip.ptr(S),
                                           _hip.ptr(rank), L, B, P, H, _hip.stream())
        _hip.check(rc, "mmdfn_party_gather")
        ctx.dims = (L, B, P, H, Mn)
        ctx.stacked = stacked
        ctx.has_bias = bias is not None
        ctx.passthrough = bool(passthrough) and not stacked
        ctx.save_for_backward(rank)
        ctx.mark_non_differentiable(rank)
        ctx.set_materialize_grads(False)       # no zero-filled stand-ins for the integer output / unused identities
        if ctx.passthrough:
            return (S, rank) + tuple(Xs)
        return S, rank

    @staticmethod
    def backward(ctx, dS, _drank, *dpass):
        (rank,) = ctx.saved_tensors
        L, B, P, H, Mn = ctx.dims
        dev = rank.device
        dS = dS.contiguous() if dS is not None else torch.zeros(L, Mn * B * P, H, dtype=torch.float32, device=dev)
        dX = torch.empty(Mn, L, B, H, dtype=torch.float32, device=dev)
        held = [None if d is None else d.contiguous() for d in dpass]
        addend = _hip.ptr_array(held) if any(h is not None for h in held) else None
        rc = _hip.lib().mmdfn_party_gather_bwd(Mn, _hip.ptr(dS), _hip.ptr(rank), _hip.ptr_array([dX[m] for m in range(Mn)]),
                                               addend, L, B, P, H, _hip.stream())
        _hip.check(rc, "mmdfn_party_gather_bwd")
        dbias = dS.sum((0, 1)) if ctx.has_bias and ctx.needs_input_grad[1] else None
        if ctx.stacked:
            return None, dbias, None, dX
        return (None, dbias, None) + tuple(dX[m] for m in range(Mn))


def party_gather(Xs, qmask, bias=None, passthrough=False):
    """Xs: list of (L, B, H) tensors or one stacked (Mn, L, B, H) tensor.  passthrough (list form only): returns
    (S, rank, X_0', .., X_{Mn-1}') with X_m' identities of the inputs (see _PartyGather)."""
    if torch.is_tensor(Xs):
        return _PartyGather.apply(qmask, bias, False, Xs)
    return _PartyGather.apply(qmask, bias, passthrough, *Xs)



def colsum(A):
    """Column sums of an (R, H) fp32 matrix (unit inner stride) in one bit-reproducible launch (csrc/encoder_glue.hip)."""
    _hip.require_cuda(A)
    _hip.require_f32(A)
    if A.stride(1) != 1:
        A = A.contiguous()
    R, H = A.shape
    lib = _hip.lib()
    ws = torch.empty(int(lib.mmdfn_colsum_workspace(H)), dtype=torch.float32, device=A.device)
    out = torch.empty(H, dtype=torch.float32, device=A.device)
    _hip.check(lib.mmdfn_colsum(_hip.ptr(A), R, H, A.stride(0), _hip.ptr(out), _hip.ptr(ws), _hip.stream()), "mmdfn_colsum")
    return out


class _HalvesGrad:
    """Stand-in 'parameter' of queue_slab_reduce for two biases that share one column-sum stack: its .grad is the (N,) buffer
    whose halves were handed to the two biases."""

    def __init__(self, buf):
        self.grad = buf


def _queue_bias_halves(A, b1, b2, n1):
    """b1.grad, b2.grad = halves of the column sums of A (R, N): the slab kernel runs now, the sum over slabs rides on the
    backward pass's last reduction launch."""
    _hip.require_cuda(A)
    _hip.require_f32(A)
    if A.stride(1) != 1:
        A = A.contiguous()
    R, N = A.shape
    lib = _hip.lib()
    ws = torch.empty(int(lib.mmdfn_colsum_workspace(N)), dtype=torch.float32, device=A.device)
    nsl = lib.mmdfn_colsum_partial(_hip.ptr(A), R, N, A.stride(0), _hip.ptr(ws), _hip.stream())
    if nsl <= 0:
        raise _hip.HipLibraryError("mmdfn_colsum_partial rejected the operand (%d)" % nsl)
    buf = torch.empty(N, dtype=torch.float32, device=A.device)
    b1.grad, b2.grad = buf[:n1], buf[n1:]
    _WGQ["ext"].append(dict(part=None, colpart=ws, splits=int(nsl), M=int(N), N=0, weight=None, bias=_HalvesGrad(buf), acc=0))
    if not _WGQ["armed"]:
        _WGQ["armed"] = True
        torch.autograd.Variable._execution_engine.queue_callback(flush_queued_wgrads)


class _ProjectGather(torch.autograd.Function):
    """First party-GRU layer without projecting padded party rows: gi_p = party_gather(X_m [W1; W2]^T) + [b1; b2] for every
    speaker-encoded modality m, as ONE node: a grouped launch of the few-row kernel for the projections (each modality its
    own problem, no stacked copy), the gather kernel, and on the way back the scatter, ONE grouped K-major launch for the
    input gradients (accumulated onto the gradient that reaches X_m through its passthrough alias), the column-sum kernel
    for the bias gradient and queued weight-gradient segments.  Outputs: (gi_p (L, Mn*B*P, N), rank, X_0', .., X_{Mn-1}')
    with X_m' identities of the inputs (see _PartyGather)."""

    @staticmethod
    def forward(ctx, qmask, w1, w2, b1, b2, wcat, bcat, *Xs):
        _hip.require_cuda(qmask, w1, w2, *Xs)
        mods = [x.contiguous() for x in Xs]
        qmask = qmask.contiguous()
        L, B, P = qmask.shape
        H = mods[0].shape[-1]
        Mn = len(mods)
        w1c, w2c = w1.contiguous(), w2.contiguous()
        n1, N = w1.shape[0], w1.shape[0] + w2.shape[0]
        G = torch.empty(Mn, L * B, N, dtype=torch.float32, device=qmask.device)
        linear_group_raw([dict(x=m.view(L * B, H), w=w1c, w2=w2c, out=G[i]) for i, m in enumerate(mods)])
        bias = None
        if b1 is not None:
            bias = bcat if bcat is not None else torch.cat([b1, b2])
        S = torch.empty(L, Mn * B * P, N, dtype=torch.float32, device=qmask.device)
        rank = torch.empty(L, B, P, dtype=torch.int32, device=qmask.device)
        rc = _hip.lib().mmdfn_party_gather(Mn, _hip.ptr_array([G[i] for i in range(Mn)]), _hip.ptr(qmask), _hip.ptr(bias),
                                           _hip.ptr(S), _hip.ptr(rank), L, B, P, N, _hip.stream())
        _hip.check(rc, "mmdfn_party_gather")
        ctx.dims = (L, B, P, H, Mn, n1, N)
        ctx.refs = (w1, w2, b1, b2)
        ctx.save_for_backward(rank, w1c, w2c, wcat, *mods)
        ctx.mark_non_differentiable(rank)
        ctx.set_materialize_grads(False)
        return (S, rank) + tuple(Xs)

    @staticmethod
    def backward(ctx, dS, _drank, *dpass):
        rank, w1, w2, wcat, *mods = ctx.saved_tensors
        p1, p2, b1, b2 = ctx.refs
        L, B, P, H, Mn, n1, N = ctx.dims
        dev = rank.device
        dS = dS.contiguous() if dS is not None else torch.zeros(L, Mn * B * P, N, dtype=torch.float32, device=dev)
        dG = torch.empty(Mn, L * B, N, dtype=torch.float32, device=dev)
        rc = _hip.lib().mmdfn_party_gather_bwd(Mn, _hip.ptr(dS), _hip.ptr(rank), _hip.ptr_array([dG[m] for m in range(Mn)]),
                                               None, L, B, P, N, _hip.stream())
        _hip.check(rc, "mmdfn_party_gather_bwd")
        db1 = db2 = None
        if b1 is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            if (ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and slab_reduce_queueable(None, [b1, b2])
                    and b1.grad is None and b2.grad is None):
                # the column sums stay slab stacks; the step's last reduction launch sums them into ONE (N,) buffer of which the two
                # biases' .grad are the halves (views handed over here, filled at the end of the backward pass)
                _queue_bias_halves(dS.view(-1, N), b1, b2, n1)
            else:
                db = colsum(dS.view(-1, N))
                db1, db2 = db[:n1], db[n1:]
        # input gradients: dX_m = dG_m [W1; W2] (+ the gradient that reached X_m's alias), one grouped launch
        dXs = [None] * Mn
        need = [m for m in range(Mn) if ctx.needs_input_grad[7 + m]]
        if need:
            # dX_m = dG_m [W1; W2] + (the gradient that reached X_m's passthrough alias), out of place: the incoming gradient is
            # only READ (the kernel's addend), never written -- autograd may hand the same tensor to several nodes (ADVICE r03;
            # an in-place accumulation needed a private copy of every view / duplicate: two 5 us copies per cfg2 step)
            adds = []
            for m in need:
                d = dpass[m] if m < len(dpass) else None
                adds.append(None if d is None else d.reshape(L * B, H))
            if wcat is not None:
                probs = []
                for m, d in zip(need, adds):
                    q = dict(x=dG[m], wk=wcat)
                    if d is not None:
                        q.update(addend=d)
                    probs.append(q)
                res = linear_group_raw(probs)
            else:
                res = []
                for m, d in zip(need, adds):
                    q = dict(x=dG[m][:, :n1], wk=w1)
                    if d is not None:
                        q.update(addend=d)
                    o = linear_group_raw([q])[0]
                    linear_group_raw([dict(x=dG[m][:, n1:], wk=w2, out=o, accumulate=True)])
                    res.append(o)
            for m, o in zip(need, res):
                dXs[m] = o.view(L, B, H)
        else:
            dXs = [d for d in dpass] + [None] * (Mn - len(dpass))
        # weight gradients: one segment per modality and direction
        dw1 = dw2 = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            x2 = [m.view(L * B, H) for m in mods]
            if _queueable(p1, [], n1, H) and _queueable(p2, [], N - n1, H):
                for m in range(Mn):
                    queue_wgrad(dG[m][:, :n1], x2[m], p1)
                    queue_wgrad(dG[m][:, n1:], x2[m], p2)
            else:
                for m in range(Mn):
                    a, _ = _wgrad_inline(dG[m][:, :n1], x2[m], False)
                    b, _ = _wgrad_inline(dG[m][:, n1:], x2[m], False)
                    dw1 = a if dw1 is None else dw1 + a
                    dw2 = b if dw2 is None else dw2 + b
        return (None, dw1, dw2, db1, db2, None, None) + tuple(dXs)


def project_gather(Xs, qmask, w1, w2, b1, b2, wcat=None, bcat=None):
    """(gi_p, rank, X_0', ..): see _ProjectGather.  ``wcat`` / ``bcat``: optional stacked views of [w1; w2] / [b1; b2] (no
    gradient flows through them; without wcat the input gradient takes two launches per modality, without bcat the bias is
    concatenated per call)."""
    if w1.shape[1] % 4 or (w1.shape[0] + w2.shape[0]) % 4:
        raise ValueError("project_gather: widths must be multiples of 4")
    return _ProjectGather.apply(qmask, w1, w2, b1, b2, wcat, bcat, *Xs)


class _PartyCombine(torch.autograd.Function):
    """out (Mn, N, H) = strip_pad(base_m + w_m * scatter(E)); E may be None (no speaker encoder), else it holds one
    (B*P)-column block per modality with a NON-ZERO weight, in modality order."""

    @staticmethod
    def forward(ctx, E, rank, flat_idx, weights, *bases):
        _hip.require_cuda(rank, *bases)
        bases = [x.contiguous() for x in bases]
        L, B, P = rank.shape
        H = bases[0].shape[-1]
        Mn = len(bases)
        N = flat_idx.numel()
        E_ = E.contiguous() if E is not None else None
        out = torch.empty(Mn, N, H, dtype=torch.float32, device=rank.device)
        rc = _hip.lib().mmdfn_party_combine(Mn, _hip.ptr_array(bases), _hip.ptr(E_), _hip.ptr(rank), _hip.ptr(flat_idx),
                                            _hip.ptr(out), _hip.float_array(weights), L, B, P, N, H, _hip.stream())
        _hip.check(rc, "mmdfn_party_combine")
        ctx.dims = (L, B, P, H, Mn, N)
        ctx.weights = list(weights)
        ctx.has_E = E is not None
        ctx.save_for_backward(rank, flat_idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        rank, flat_idx = ctx.saved_tensors
        L, B, P, H, Mn, N = ctx.dims
        dout = dout.contiguous()
        nact = sum(1 for w in ctx.weights[:Mn] if w != 0.0)
        nb, ne = Mn * L * B * H, (L * nact * B * P * H if ctx.has_E else 0)
        zero = torch.zeros(nb + ne, dtype=torch.float32, device=dout.device)       # one fill for both (pad rows stay 0)
        dbase = zero[:nb].view(Mn, L, B, H)
        dE = zero[nb:].view(L, nact * B * P, H) if ctx.has_E else None
        rc = _hip.lib().mmdfn_party_combine_bwd(Mn, _hip.ptr(dout), _hip.ptr(rank), _hip.ptr(flat_idx),
                                                _hip.ptr_array([dbase[m] for m in range(Mn)]), _hip.ptr(dE),
                                                _hip.float_array(ctx.weights), L, B, P, N, H, _hip.stream())
        _hip.check(rc, "mmdfn_party_combine_bwd")
        return (dE, None, None, None) + tuple(dbase[m] for m in range(Mn))


def party_combine(bases, E, rank, flat_idx, weights):
    return _PartyCombine.apply(E, rank, flat_idx, list(weights), *bases)


def linear_raw(x2d, weight, bias=None, act=0, out=None, accumulate=False):
    """Y = act(x2d @ weight.T + bias) (+ out) on the MFMA kernel; x2d (R, K) row-strided ok, weight (N, K)."""
    _hip.require_cuda(x2d, weight)
    _hip.require_f32(x2d, weight, bias, out)
    R, K = x2d.shape
    N = weight.shape[0]
    if x2d.stride(1) != 1 or x2d.stride(0) % 4 or x2d.data_ptr() % 16:
        x2d = x2d.contiguous()
    weight = weight.contiguous()
    if out is None:
        out = torch.empty(R, N, dtype=torch.float32, device=x2d.device)
    rc = _hip.lib().mmdfn_linear(_hip.ptr(x2d), _hip.ptr(weight), _hip.ptr(bias), _hip.ptr(out), R, K, N,
                                 x2d.stride(0), out.stride(0), int(act), 1 if accumulate else 0, _hip.stream())
    _hip.check(rc, "mmdfn_linear")
    return out


def linear_group_raw(problems, act=0):
    """One launch for up to 8 few-row projections (csrc/linear_small.hip).  Each problem is a dict: x (R, K) fp32 rows,
    either ``w`` (N, K) [+ ``w2`` (N2, K): second row block] with optional ``b`` / ``b2``, or ``wk`` (K, N) (n-contiguous:
    y = x @ wk); optional ``out`` (+ ``accumulate``: y += ...), optional ``addend`` (R, N): y = ... + addend, out of place
    (the addend is only read).  Returns the outputs."""
    n = len(problems)
    X, W, W2, B1, B2, Y, Z = [], [], [], [], [], [], []
    R, K, N, N1, ldx, ldw, ldy, ldz, km, acc = [], [], [], [], [], [], [], [], [], []
    for q in problems:
        x = q["x"]
        if x.dtype != torch.float32 or x.dim() != 2:
            raise ValueError("linear_group_raw: x must be an fp32 matrix")
        if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
            x = x.contiguous()
        if "wk" in q:
            w = q["wk"]
            w = w if (w.stride(1) == 1 and w.data_ptr() % 16 == 0 and w.stride(0) % 4 == 0) else w.contiguous().clone()
            k_, n_ = w.shape
            W.append(w); W2.append(None); B1.append(None); B2.append(None); N1.append(n_); km.append(1); ldw.append(w.stride(0))
        else:
            w, w2 = q["w"], q.get("w2")
            # (16-byte aligned rows: the LDS-DMA form fetches them in 16-byte units, the register form refuses K > 768;
            # FlatAdam's slots and freshly allocated parameters are aligned, an odd view of somebody else's buffer is copied)
            w = w if (w.is_contiguous() and w.data_ptr() % 16 == 0) else w.contiguous().clone()
            if w2 is not None and not (w2.is_contiguous() and w2.data_ptr() % 16 == 0):
                w2 = w2.contiguous().clone()
            n_, k_ = w.shape[0] + (w2.shape[0] if w2 is not None else 0), w.shape[1]
            W.append(w); W2.append(w2); B1.append(q.get("b")); B2.append(q.get("b2")); N1.append(w.shape[0]); km.append(0)
            ldw.append(k_)
        if x.shape[1] != k_:
            raise ValueError("linear_group_raw: contraction widths differ")
        out = q.get("out")
        if out is None:
            out = torch.empty(x.shape[0], n_, dtype=torch.float32, device=x.device)
        X.append(x); Y.append(out); R.append(x.shape[0]); K.append(k_); N.append(n_); ldx.append(x.stride(0)); ldy.append(out.stride(0))
        acc.append(1 if q.get("accumulate") else 0)
        z = q.get("addend")
        if z is not None:
            if z.dtype != torch.float32 or tuple(z.shape) != (x.shape[0], n_):
                raise ValueError("linear_group_raw: addend must be an fp32 (R, N) matrix")
            if z.stride(1) != 1:
                z = z.contiguous()
            _hip.require_cuda(z)
        Z.append(z); ldz.append(z.stride(0) if z is not None else 0)
    _hip.require_cuda(*X, *W)
    _hip.require_f32(*X, *W)
    ia, pa = _hip.int_array, _hip.ptr_array
    if any(z is not None for z in Z):
        rc = _hip.lib().mmdfn_linear_group_addend(n, pa(X), pa(W), pa(W2), ia(N1), pa(B1), pa(B2), pa(Y), pa(Z), ia(ldz), ia(R),
                                                  ia(K), ia(N), ia(ldx), ia(ldw), ia(ldy), ia(km), ia(acc), int(act),
                                                  _hip.stream())
        _hip.check(rc, "mmdfn_linear_group_addend")
        return Y
    rc = _hip.lib().mmdfn_linear_group(n, pa(X), pa(W), pa(W2), ia(N1), pa(B1), pa(B2), pa(Y), ia(R), ia(K), ia(N), ia(ldx),
                                       ia(ldw), ia(ldy), ia(km), ia(acc), int(act), _hip.stream())
    _hip.check(rc, "mmdfn_linear_group")
    return Y


def linear_group_supported(R, K, N):
    return bool(_hip.lib().mmdfn_linear_group_supported(int(R), int(K), int(N)))


def dense_nk(x2, weight, bias=None, act=0, out=None, accumulate=False):
    """act(x2 W^T + b) (+ out) for W stored (N, K).  Engine per shape, both hand-written: the many-row kernels where they
    win (linear_preferred: csrc/linear.hip, linear_split.hip), the LDS-staged few-row kernel (csrc/linear_small.hip)
    otherwise.  A contraction width that is not a multiple of 4 runs on the zero-padded operands (row_operand /
    weight_operand above): there is no library GEMM on any path."""
    if weight.shape[1] % 4:
        x2, weight = row_operand(x2), weight_operand(weight)
    N, K = weight.shape
    if K < 4:
        raise _hip.HipLibraryError("dense_nk: empty contraction")
    if linear_preferred(x2.shape[0], K, N):
        return linear_raw(x2, weight, bias, act, out=out, accumulate=accumulate)
    q = dict(x=x2, w=weight, b=bias)
    if out is not None:
        q.update(out=out, accumulate=accumulate)
    return linear_group_raw([q], act)[0]


def dense_kn(x2, wk):
    """x2 @ wk for wk stored (K, N) (an input gradient dX = dY . W read as stored; GraphConvolution.weight): the few-row
    kernel's K-major form.  K (the rows of wk) not a multiple of 4: the contraction runs over zero-padded copies; N (the
    row length of wk) not a multiple of 4: over the row-padded form of wk, the result is cut back to N columns."""
    K, N = wk.shape
    if K % 4:
        x2 = row_operand(x2)
        wkp = torch.zeros(pad4(K), wk.shape[1], dtype=wk.dtype, device=wk.device)
        wkp[:K].copy_(wk)
        wk = wkp
    if N % 4:
        return linear_group_raw([dict(x=x2, wk=weight_operand(wk))])[0][:, :N]
    return linear_group_raw([dict(x=x2, wk=wk)])[0]


def linear_supported(x, weight):
    return x.is_cuda and x.dtype == torch.float32 and weight.shape[1] >= 4


def linear_preferred(rows, K, N):
    """Shapes that run on the many-row kernels (csrc/linear.hip, linear_split.hip; tools/bench_linear.py, round 1): many rows
    and a short contraction (the batched party-GRU input projection, the GCN input layer, the LSTM gate pre-activations), or
    many 128 x 128 tiles.  Everything else goes to the LDS-staged few-row kernel (csrc/linear_small.hip, dense_nk / dense_kn)."""
    if rows >= 4096 and K <= 256:
        return True
    # many 128 x 128 output tiles: the bf16-piece variant (csrc/linear_split.hip) also wins at long K
    return ((rows + 127) // 128) * ((N + 127) // 128) >= 256 and K <= 1024


def _strided_rows(t):
    """2-D fp32 view usable by the strided kernels (unit inner stride, 16-byte aligned rows) or a copy."""
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def gemm_tn_supported(M, N):
    return M % 4 == 0 and N % 4 == 0


def gemm_tn(A, B, want_colsum=False):
    """C = A^T @ B for A (R, M), B (R, N) (row-strided views accepted) and optionally colsum = A.sum(0);
    the reduction over the R rows is split across workgroups (csrc/gemm_tn.hip)."""
    _hip.require_cuda(A, B)
    _hip.require_f32(A, B)
    A = _strided_rows(A)
    B = _strided_rows(B)
    R, M = A.shape
    N = B.shape[1]
    lib = _hip.lib()
    splits = lib.mmdfn_gemm_tn_splits(R, M, N)
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    colsum = torch.empty(M, dtype=torch.float32, device=A.device) if want_colsum else None
    ws = torch.empty(splits * (M * N + M), dtype=torch.float32, device=A.device)
    rc = lib.mmdfn_gemm_tn(_hip.ptr(A), _hip.ptr(B), _hip.ptr(C), _hip.ptr(colsum), _hip.ptr(ws), R, M, N,
                           A.stride(0), B.stride(0), N, splits, _hip.stream())
    _hip.check(rc, "mmdfn_gemm_tn")
    return C, colsum


def gemm_tn_grouped(problems):
    """ONE launch pair for up to 8 contractions  C_p = sum_r A_p[r]^T B_p[r + shift_p]  (+ column sums of A_p).

    problems: list of dicts with A (R, M), B (R, N) row-strided views, C (M, N) output view (row stride ldc), optional
    colsum (M,) output view and shift (int, rows; B rows outside [0, R) count as zero).  Outputs are written in place."""
    n = len(problems)
    if not 1 <= n <= 8:
        raise ValueError("gemm_tn_grouped takes 1..8 problems")
    A = [_strided_rows(p["A"]) for p in problems]
    B = [_strided_rows(p["B"]) for p in problems]
    _hip.require_cuda(*A, *B)
    _hip.require_f32(*A, *B)
    C = [p["C"] for p in problems]
    cs = [p.get("colsum") for p in problems]
    R = [a.shape[0] for a in A]
    M = [a.shape[1] for a in A]
    N = [b.shape[1] for b in B]
    for p, a, b, c in zip(problems, A, B, C):
        if b.shape[0] != a.shape[0] or tuple(c.shape) != (a.shape[1], b.shape[1]) or c.stride(1) != 1:
            raise ValueError("gemm_tn_grouped: inconsistent problem shapes")
    lib = _hip.lib()
    ia = _hip.int_array
    nws = lib.mmdfn_gemm_tn_grouped_workspace(n, ia(R), ia(M), ia(N))
    ws = torch.empty(int(nws), dtype=torch.float32, device=A[0].device)
    cs_arr = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in cs])
    rc = lib.mmdfn_gemm_tn_grouped(n, _hip.ptr_array(A), _hip.ptr_array(B), _hip.ptr_array(C), cs_arr, ia(R), ia(M), ia(N),
                                   ia([a.stride(0) for a in A]), ia([b.stride(0) for b in B]),
                                   ia([c.stride(0) for c in C]), ia([int(p.get("shift", 0)) for p in problems]),
                                   _hip.ptr(ws), _hip.stream())
    _hip.check(rc, "mmdfn_gemm_tn_grouped")
    return ws   # kept alive by the caller's frame until the launches are enqueued (stream-ordered allocator)


# ---------------------------------------------------------------------------------------------------
# Weight-gradient queue.  dW / db of the dense layers, the GRU weights, the LSTM gate and the GCN layers feed
# nothing else in the backward pass and each is far too small to fill the chip (cfg2: ~25 contractions of
# 1.7k-7k rows into 100x200 .. 600x200 outputs, ten launch pairs and 300 us per step when issued one by one).
# Inside a ``wgrad_batch()`` scope -- ``train.backward(loss)`` opens one around ``loss.backward()``, so do the captured
# steps and the pass loop -- they are only QUEUED during backward; an autograd end-of-backward callback issues all of
# them as ONE launch pair (csrc/gemm_tn.hip, batch form) that writes straight into the parameters' .grad: no autograd
# accumulation kernels, and the contributions of a parameter used several times (the layer-shared LSTM gate) are summed
# inside the slab reduction.  This bypasses autograd for those parameters (backward returns None for them), so it is
# opt-in by scope and per parameter:
#   * outside a scope (a plain ``loss.backward()``, ``torch.autograd.grad(...)``) every node computes its weight
#     gradients in line and RETURNS them: autograd.grad sees them, nothing is written to .grad behind its back;
#   * a parameter with Python tensor hooks or post-accumulate-grad hooks always takes the in-line path.  torch's
#     DistributedDataParallel registers its reducer on the AccumulateGrad node in C++, which is NOT visible here: do not run
#     train.backward / ops.wgrad_batch() on a DDP-wrapped model (the queued parameters would bypass the reducer) -- use the
#     package's own GradientBucket (distributed.py), or a plain loss.backward();
#   * non-leaf weights always take the in-line path.
# The queue belongs to one backward pass: entering the outermost scope drops anything a failed backward left behind, and
# leaving it flushes what the engine callback did not (or clears the queue when the backward raised).
# ---------------------------------------------------------------------------------------------------
_WGQ = {"segs": [], "outs": {}, "ext": [], "armed": False, "scope": 0, "side": None, "held": [], "pending_join": False}
# MMDFN_EARLY_WGRAD=1: issue the graph-side weight gradients (GCN stack, LSTM gate) on a second stream as soon as the graph
# part of the backward pass is done, concurrently with the GRU backward recurrence.  OFF by default: measured slower at
# cfg2 (1.133 vs 1.107 ms per step; the split batches cost 47.6 + 111.5 us against 140.3 us for one, and the concurrent
# recurrence slows from 82.8 to 89.0 us -- timeline in profiles/r03_wgrad_overlap.md).  Kept because it is where a
# two-part gradient bucket would start its first all-reduce on a multi-GPU node.
EARLY_WGRAD = __import__("os").environ.get("MMDFN_EARLY_WGRAD", "0") == "1"
_WG_MAX = 40        # TN_MAXSEG / TN_MAXOUT of csrc/gemm_tn.hip


class wgrad_batch:
    """``with ops.wgrad_batch(): loss.backward()`` -- weight gradients of leaf parameters are batched into one launch pair
    and written to ``.grad`` directly (see the comment above).  Re-entrant; exception-safe."""

    def __enter__(self):
        if _WGQ["scope"] == 0 and (_WGQ["outs"] or _WGQ["ext"] or _WGQ["armed"]):
            _WGQ["outs"], _WGQ["ext"], _WGQ["armed"] = {}, [], False        # stale entries of a backward that raised
        if _WGQ["scope"] == 0:
            drop_grad_addends()                                             # (same: its callback never ran)
        _WGQ["scope"] += 1
        return self

    def __exit__(self, exc_type, exc, tb):
        _WGQ["scope"] -= 1
        if _WGQ["scope"] == 0:
            if exc_type is None:
                if _WGQ["outs"] or _WGQ["ext"]:
                    flush_queued_wgrads()                  # a backward driven without the engine callback
            else:
                _WGQ["outs"], _WGQ["ext"], _WGQ["armed"] = {}, [], False    # the callback never ran: drop the half-built batch
                drop_grad_addends()
            _join_side()
        return False


def wgrad_batching():
    return _WGQ["scope"] > 0


def _leaf(p):
    return p is not None and p.is_leaf and p.requires_grad


def _hooked(p):
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


def _queueable(weight, biases, M, N):
    return (_WGQ["scope"] > 0 and _leaf(weight) and all(_leaf(b) for b in biases) and gemm_tn_supported(M, N)
            and len(biases) <= 2 and not _hooked(weight) and not any(_hooked(b) for b in biases))


def queue_wgrad(A, B, weight, biases=(), shift=0, rows=None):
    """weight.grad (M, N) += sum_r A[r]^T B[r + shift];  b.grad (M) += column sums of A for every b in ``biases``.
    ``rows = (r0, r1)``: the contraction fills rows r0..r1-1 of weight.grad only (GraphConvolution.weight takes its two
    halves from hi^T dP and h0^T dP, the concatenated operand [hi | h0] never exists).
    Only valid inside a backward pass under ``wgrad_batch()`` (the flush is an end-of-backward callback)."""
    if _WGQ["scope"] <= 0:
        raise RuntimeError("queue_wgrad outside a wgrad_batch() scope")
    A = _strided_rows(A)
    B = _strided_rows(B)
    r0, r1 = rows if rows is not None else (0, weight.shape[0])
    key = (id(weight), r0)
    out = _WGQ["outs"].get(key)
    if out is None:
        out = dict(weight=weight, biases=[], M=A.shape[1], N=B.shape[1], segs=[], rows=(r0, r1))
        _WGQ["outs"][key] = out
    for b in biases:
        if all(b is not x for x in out["biases"]):
            out["biases"].append(b)
    if (len(out["biases"]) > 2 or (A.shape[1], B.shape[1]) != (out["M"], out["N"]) or out["rows"] != (r0, r1)
            or r1 - r0 != out["M"] or pad4(weight.shape[1]) != pad4(out["N"]) or out["N"] < weight.shape[1]
            or (rows is not None and biases)):
        raise RuntimeError("queue_wgrad: inconsistent contributions to one parameter")
    out["segs"].append((A, B, int(shift)))
    if not _WGQ["armed"]:
        _WGQ["armed"] = True
        torch.autograd.Variable._execution_engine.queue_callback(flush_queued_wgrads)


SLAB_RIDE = __import__("os").environ.get("MMDFN_SLAB_RIDE", "1") == "1"      # (0: A/B aid, the stacks get their own launches)


def slab_reduce_queueable(weight, biases):
    """May the slab stacks of ``weight`` / ``biases`` (partial sums another kernel wrote) be summed by the end-of-backward
    reduction launch instead of a launch of their own?  Same rules as queue_wgrad."""
    ps = ([weight] if weight is not None else []) + list(biases)
    return SLAB_RIDE and _WGQ["scope"] > 0 and bool(ps) and all(_leaf(p) and not _hooked(p) for p in ps)


def queue_slab_reduce(part, colpart, splits, M, N, weight=None, biases=()):
    """weight.grad (M, N) += sum of the ``splits`` slabs of ``part`` ([splits][M][N]); b.grad (M) += sum of the slabs of
    ``colpart`` ([splits][M]) for the ONE bias in ``biases`` -- summed by the reduction launch of the backward pass's
    weight-gradient batch (mmdfn_gemm_tn_batch_ext).  Only inside a backward pass under ``wgrad_batch()``."""
    if _WGQ["scope"] <= 0:
        raise RuntimeError("queue_slab_reduce outside a wgrad_batch() scope")
    if len(biases) > 1 or (weight is None) != (part is None) or (colpart is None) != (len(biases) == 0):
        raise RuntimeError("queue_slab_reduce: one weight and / or one bias per slab stack")
    _WGQ["ext"].append(dict(part=part, colpart=colpart, splits=int(splits), M=int(M), N=int(N) if weight is not None else 0,
                            weight=weight, bias=biases[0] if biases else None))
    if not _WGQ["armed"]:
        _WGQ["armed"] = True
        torch.autograd.Variable._execution_engine.queue_callback(flush_queued_wgrads)


def _join_side():
    """The main stream waits for the side-stream batch (if one is in flight); its operands may be released afterwards."""
    if _WGQ["pending_join"]:
        torch.cuda.current_stream().wait_stream(_WGQ["side"])
        _WGQ["pending_join"] = False
    _WGQ["held"] = []


_GRAPH_DONE_HOOK = [None]


def set_graph_backward_done_hook(fn):
    """``fn()`` is called ONCE where the graph part of the next backward pass ends (the adjacency builder's backward: the
    graph stack's, the fusion modules' and the head's gradients are complete, the encoders' nodes follow).  A two-part
    gradient bucket (distributed.GradientBucket(parts=2)) starts its first all-reduce there."""
    _GRAPH_DONE_HOOK[0] = fn


def flush_queued_wgrads_now():
    """Issue the weight gradients queued so far on the current stream (the end-of-backward callback flushes the rest)."""
    if _WGQ["outs"] or _WGQ["ext"]:
        outs, ext = list(_WGQ["outs"].values()), _WGQ["ext"]
        _WGQ["outs"], _WGQ["ext"] = {}, []   # 'armed' stays set: the end-of-backward callback still runs for the rest
        _flush_outs(outs, None, ext)


def flush_queued_wgrads_early():
    """Called where the graph part of the backward pass ends (the adjacency builder's backward): what is queued so far
    leaves NOW on a side stream, concurrently with the encoder backward that follows on the main stream.  Gradient
    buffers and the workspace are allocated on the main stream (the caching allocator's stream of record), the operands
    stay referenced until the main stream has waited for the side stream (end-of-backward callback)."""
    hook = _GRAPH_DONE_HOOK[0]
    if hook is not None:
        hook()                               # (a two-part bucket: flushes what is queued and starts its first collective)
        return
    if not (EARLY_WGRAD and _WGQ["scope"] > 0 and _WGQ["outs"]):
        return
    outs = list(_WGQ["outs"].values())
    _WGQ["outs"] = {}                        # 'armed' stays: the end-of-backward callback flushes the rest and joins
    if _WGQ["side"] is None:
        _WGQ["side"] = torch.cuda.Stream()
    _flush_outs(outs, _WGQ["side"])


_GRAD_ADDENDS = []      # (parameter, tensor, event recorded on the producing stream)
_GRAD_ADDENDS_ARMED = [False]    # the end-of-backward callback of the RUNNING backward pass has been queued


def drop_grad_addends():
    """Forget addends (and the callback flag) left behind by a backward pass that raised before its end-of-backward callback
    ran; without this every later pass would see a non-empty list, queue no callback and silently lose its addends."""
    del _GRAD_ADDENDS[:]
    _GRAD_ADDENDS_ARMED[0] = False


def add_grad_addends(pairs):
    """``p.grad += t`` for every (p, t) at the END of the running backward pass, on the stream that runs it, behind an
    event recorded now on the CURRENT stream (the producer: gru._GruTable's backward on the side stream).  For small
    gradient pieces computed off the main stream that neither autograd nor the weight-gradient batch should wait for."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    for prm, t in pairs:
        if prm.requires_grad:
            _GRAD_ADDENDS.append((prm, t, ev))
    if _GRAD_ADDENDS and not _GRAD_ADDENDS_ARMED[0]:
        _GRAD_ADDENDS_ARMED[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(apply_grad_addends)


def apply_grad_addends():
    _GRAD_ADDENDS_ARMED[0] = False
    if not _GRAD_ADDENDS:
        return
    items = list(_GRAD_ADDENDS)
    del _GRAD_ADDENDS[:]
    cur = torch.cuda.current_stream()
    seen = set()
    for _, t, ev in items:
        if id(ev) not in seen:
            seen.add(id(ev))
            cur.wait_event(ev)
        t.record_stream(cur)
    dst, src = [], []
    for prm, t, _ in items:
        if prm.grad is None:
            prm.grad = t if tuple(t.shape) == tuple(prm.shape) else t.view(prm.shape).clone()
        else:
            dst.append(prm.grad)
            src.append(t.view(prm.grad.shape))
    if dst:
        torch._foreach_add_(dst, src)


def flush_queued_wgrads():
    """Issue every queued weight-gradient contraction (one launch pair per <= 40 segments) into the .grad fields."""
    outs, ext = list(_WGQ["outs"].values()), _WGQ["ext"]
    _WGQ["outs"], _WGQ["ext"], _WGQ["armed"] = {}, [], False
    _join_side()             # first: a parameter may collect contributions from both batches (the later one accumulates)
    if outs or ext:
        _flush_outs(outs, None, ext)


def _ext_destinations(ext):
    """.grad destinations of foreign slab stacks: fresh buffers (written) or the existing .grad (accumulated)."""
    items = []
    for e in ext:
        w, b = e["weight"], e["bias"]
        have = [p.grad is not None for p in (w, b) if p is not None]
        acc = e["acc"] if e.get("acc") is not None else (1 if any(have) else 0)
        C = cs = None
        if w is not None:
            if w.grad is None:
                w.grad = (torch.zeros if acc else torch.empty)(e["M"], e["N"], dtype=torch.float32, device=e["part"].device)
            elif not w.grad.is_contiguous():
                w.grad = w.grad.contiguous()
            C = w.grad.view(e["M"], e["N"])
        if b is not None:
            if b.grad is None:
                b.grad = (torch.zeros if acc else torch.empty)(e["M"], dtype=torch.float32, device=e["colpart"].device)
            cs = b.grad
        items.append((e, C, cs, acc))
    return items


def _flush_outs(outs, side, ext=()):
    ext_items = _ext_destinations(ext) if ext else []
    if not outs:
        _prepare_wgrad_batch([], ext_items)(_hip.stream())
        return
    dev = outs[0]["weight"].device
    # gradient destinations: fresh buffers handed to .grad (the usual case: backward runs with .grad = None), or the
    # existing .grad accumulated in place
    work = []                                   # (out, C, colsum targets, accumulate, segment slice)
    fresh = set()                               # gradients allocated by this flush
    covered = {}                                # id(weight) -> rows filled by this flush (row-range contributions)
    for o in outs:
        covered[id(o["weight"])] = covered.get(id(o["weight"]), 0) + (o["rows"][1] - o["rows"][0])
    for o in outs:
        w = o["weight"]
        have = [w.grad is not None and id(w) not in fresh] + [b.grad is not None for b in o["biases"]]
        acc = any(have)
        full = o["rows"] == (0, w.shape[0])
        padded = o["N"] != w.shape[1]            # odd-width layer: the batch writes the row-padded (rows, Kp) gradient
        if w.grad is None:
            # row ranges that together cover the parameter (GraphConvolution.weight: [hi^T dP ; h0^T dP]) need no
            # zero fill; a range that leaves rows nobody writes does
            whole = covered[id(w)] >= w.shape[0]
            if padded:
                w.grad = padded_grad_like(w, zero=not (whole and not acc))
            else:
                w.grad = (torch.empty if (whole and not acc) else torch.zeros)(tuple(w.shape), dtype=torch.float32, device=dev)
            fresh.add(id(w))
            if whole and not acc:
                fresh.add(("written", id(w)))
        elif padded:
            if row_padded_view(w.grad) is None:
                w.grad = padded_grad_like(w, w.grad)
        elif not w.grad.is_contiguous():
            w.grad = w.grad.contiguous()
        if padded:
            gfull = row_padded_view(w.grad)
            C = gfull if full else gfull[o["rows"][0]:o["rows"][1]]
        else:
            C = w.grad if full else w.grad[o["rows"][0]:o["rows"][1]]
        if id(w) in fresh and not full:
            acc_here = 0 if ("written", id(w)) in fresh else 1       # zero-initialised: adding is the same as writing
        else:
            acc_here = 1 if acc else 0
        cs = []
        for b in o["biases"]:
            if b.grad is None:
                b.grad = (torch.zeros if acc_here else torch.empty)(o["M"], dtype=torch.float32, device=dev)
            cs.append(b.grad)
        segs = o["segs"]
        for i in range(0, len(segs), _WG_MAX):          # a parameter with > 40 contributions: later pieces accumulate
            work.append((o, C, cs, 1 if (acc_here or i > 0) else 0, segs[i:i + _WG_MAX]))
    batches, batch, nseg = [], [], 0
    for item in work:
        if batch and (nseg + len(item[4]) > _WG_MAX or len(batch) >= _WG_MAX):
            batches.append(batch)
            batch, nseg = [], 0
        batch.append(item)
        nseg += len(item[4])
    if batch:
        batches.append(batch)
    # foreign slab stacks ride on the last batch's reduction launch (a launch of their own when it has no room left, or when
    # the batch leaves on the side stream)
    ride = bool(ext_items) and side is None and len(batches[-1]) + len(ext_items) <= _WG_MAX
    prepared = [_prepare_wgrad_batch(b, ext_items if (ride and b is batches[-1]) else None) for b in batches]          # allocations (workspace) on the current stream
    if ext_items and not ride:
        _prepare_wgrad_batch([], ext_items)(_hip.stream())
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())             # operands, zero fills and allocations are ordered before
        _WGQ["held"].append((outs, prepared))
        _WGQ["pending_join"] = True
    stream = _hip.stream() if side is None else ctypes.c_void_p(side.cuda_stream)
    for call in prepared:
        call(stream)


def _launch_wgrad_batch(batch):
    _prepare_wgrad_batch(batch)(_hip.stream())


def _prepare_wgrad_batch(batch, ext_items=None):
    lib = _hip.lib()
    ia = _hip.int_array
    pa = lambda ts: (ctypes.c_void_p * max(1, len(ts)))(*[None if t is None else t.data_ptr() for t in ts])
    ext_items = ext_items or []
    ext_args = None
    if ext_items:
        ep = [e["part"] for e, _, _, _ in ext_items]
        ec = [e["colpart"] for e, _, _, _ in ext_items]
        eC = [C for _, C, _, _ in ext_items]
        es = [cs for _, _, cs, _ in ext_items]
        _hip.require_f32(*[t for t in ep + ec + eC + es if t is not None])
        ext_args = (len(ext_items), pa(ep), pa(ec), pa(eC), pa(es), ia([e["M"] for e, _, _, _ in ext_items]),
                    ia([e["N"] for e, _, _, _ in ext_items]), ia([0 if C is None else C.stride(0) for _, C, _, _ in ext_items]),
                    ia([e["splits"] for e, _, _, _ in ext_items]), ia([a for _, _, _, a in ext_items]))
        ext_keep = (ep, ec, eC, es)
    if not batch:
        if not ext_items:
            return lambda stream: None

        def call_ext(stream, _keep=ext_keep):
            rc = lib.mmdfn_gemm_tn_batch_ext(0, None, None, None, None, None, None, None, 0, None, None, None, None, None, None,
                                             None, None, *ext_args, stream)
            _hip.check(rc, "mmdfn_gemm_tn_batch_ext")
        return call_ext
    A, B, R, lda, ldb, sh, oi = [], [], [], [], [], [], []
    C, cs1, cs2, M, N, ldc, acc = [], [], [], [], [], [], []
    for k, (o, Ct, cs, a, segs) in enumerate(batch):
        C.append(Ct)
        cs1.append(cs[0] if len(cs) > 0 else None)
        cs2.append(cs[1] if len(cs) > 1 else None)
        M.append(o["M"]); N.append(o["N"]); ldc.append(Ct.stride(0)); acc.append(a)
        for (At, Bt, s_) in segs:
            A.append(At); B.append(Bt); R.append(At.shape[0]); lda.append(At.stride(0)); ldb.append(Bt.stride(0))
            sh.append(s_); oi.append(k)
    _hip.require_cuda(*A, *B)
    _hip.require_f32(*A, *B, *C)
    nws = lib.mmdfn_gemm_tn_batch_workspace(len(A), ia(R), ia(oi), len(C), ia(M), ia(N))
    if nws < 0:
        raise _hip.HipLibraryError("mmdfn_gemm_tn_batch_workspace rejected the batch")
    ws = torch.empty(int(nws), dtype=torch.float32, device=A[0].device)

    def call(stream, _keep=(A, B, C, cs1, cs2, ws, ext_items)):
        if ext_args is not None:
            rc = lib.mmdfn_gemm_tn_batch_ext(len(A), pa(A), pa(B), ia(R), ia(lda), ia(ldb), ia(sh), ia(oi), len(C), pa(C), pa(cs1),
                                             pa(cs2), ia(M), ia(N), ia(ldc), ia(acc), _hip.ptr(ws), *ext_args, stream)
            _hip.check(rc, "mmdfn_gemm_tn_batch_ext")
            return
        rc = lib.mmdfn_gemm_tn_batch(len(A), pa(A), pa(B), ia(R), ia(lda), ia(ldb), ia(sh), ia(oi), len(C), pa(C), pa(cs1),
                                     pa(cs2), ia(M), ia(N), ia(ldc), ia(acc), _hip.ptr(ws), stream)
        _hip.check(rc, "mmdfn_gemm_tn_batch")
    return call


def join_weight_grads():
    """Kept for callers of earlier versions: the queue is flushed by the end-of-backward callback (or by the exit of the
    ``wgrad_batch()`` scope); anything still queued here is issued now."""
    if _WGQ["outs"] and _WGQ["scope"] == 0:
        flush_queued_wgrads()


def set_async_weight_grads(flag):
    """Removed option (side-stream weight gradients measured slower on MI355X, profiles/r01_propagate_tuning.md)."""
    if flag:
        raise NotImplementedError("side-stream weight gradients were removed: they are batched into one launch now")


def _wgrad_inline(dy2, x2, want_b):
    """(dW = dy2^T x2, db = column sums of dy2 or None) computed now and returned to autograd.  x2 may be the row-padded
    operand of an odd-width layer: the caller cuts dW back to the parameter's columns.  An output width that is not a
    multiple of 4 (no layer of the model: the class scores go through the head kernel) contracts over a zero-padded copy
    of dY and cuts the rows back."""
    M = dy2.shape[1]
    if M % 4:
        dw, db = gemm_tn(row_operand(dy2), row_operand(x2), want_colsum=want_b)
        return dw[:M], (db[:M] if want_b else None)
    return gemm_tn(dy2, row_operand(x2), want_colsum=want_b)


def _wgrad(dy2, x2, weight, bias):
    """dW = dy2^T x2 (+ db = column sums of dy2): queued for the end-of-backward batch when the targets are leaf
    parameters (returns (None, None): the batch writes .grad itself), computed in line otherwise.  ``x2`` is the operand
    the forward contracted over (row-padded for an odd-width layer: its gradient then has the padded layout too)."""
    want_b = bias is not None
    if _queueable(weight, [bias] if want_b else [], dy2.shape[1], x2.shape[1]):
        queue_wgrad(dy2, x2, weight, [bias] if want_b else [])
        return None, None
    dw, db = _wgrad_inline(dy2, x2, want_b)
    if dw.shape[1] != weight.shape[1]:
        dw = dw[:, :weight.shape[1]]
    return dw, db


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (+ base).  Engine per shape (dense_nk): the many-row MFMA kernels where linear_preferred says so,
    the few-row kernel otherwise; dW / db through the step's weight-gradient batch (or in line outside ops.wgrad_batch())."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, base):
        shp = x.shape
        x2 = row_operand(x.reshape(-1, shp[-1]))          # (an odd contraction width: the zero-padded operands)
        wop = weight_operand(weight)
        N, K = wop.shape
        mfma = linear_supported(x2, wop) and linear_preferred(x2.shape[0], K, N)
        if mfma:
            if base is not None:
                y = linear_raw(x2, wop, bias, 0, out=base.reshape(-1, N).clone(), accumulate=True)
                if act:
                    y = torch.relu_(y)
            else:
                y = linear_raw(x2, wop, bias, act)
        elif base is not None:
            y = dense_nk(x2, wop, bias, act, out=base.reshape(-1, N).clone(), accumulate=True)
        else:
            y = dense_nk(x2, wop, bias, act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.has_base = base is not None
        ctx.weight_ref, ctx.bias_ref = weight, bias        # the parameter objects themselves (leaf test in backward)
        ctx.save_for_backward(x2, weight, y if act else None)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        N, K = weight.shape
        dy2 = dy.reshape(-1, N)
        if ctx.act:
            dy2 = dy2 * (y > 0).to(dy2.dtype)
        dy2 = dy2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = _wgrad(dy2, x2, ctx.weight_ref, ctx.bias_ref)         # dW and db in one pass over dY
        if ctx.needs_input_grad[0]:
            dx = _linear_dx(dy2, weight_operand(weight))[:, :K].reshape(*dy.shape[:-1], K)
        dbase = dy2.view(dy.shape) if ctx.has_base and ctx.needs_input_grad[4] else None
        return dx, dw, db, None, dbase


class _MatmulKN(torch.autograd.Function):
    """y = x @ W with W stored (K, N) (GraphConvolution.weight, model_GCN.py:169,186); dW on the side stream."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.weight_ref = w
        ctx.save_for_backward(x, w)
        return dense_kn(x if x.dim() == 2 and x.stride(1) == 1 else x.contiguous(), w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[1]:
            dw, _ = _wgrad(x, dy, ctx.weight_ref, None)                    # dW = x^T dy
        if ctx.needs_input_grad[0]:
            dx = dense_nk(dy, w)                     # dX = dY W^T: w (K, N) is the (N_out = K, contraction = N) weight as stored
        return dx, dw


def _linear_forward(x2, weight, bias, act):
    """act(x2 W^T + b) with the engine _Linear picks for the shape."""
    return dense_nk(x2, weight, bias, act)


def _linear_dx(dy2, weight):
    N, K = weight.shape
    if N % 4 == 0 and linear_preferred(dy2.shape[0], N, K) and dy2.shape[0] >= 16384:
        # very many rows: the bf16-piece kernel on a transposed copy of the weight (cfg3's 19 008 party rows: 59 us
        # against 68 us for the K-major few-row form, tools/bench_linear_group.py)
        return linear_raw(dy2, weight.t().contiguous(), None, 0)
    return dense_kn(dy2, weight)


class _LinearGroup(torch.autograd.Function):
    """n independent projections y_g = act(x_g W_g^T + b_g) that become available together (the three modality
    projections model.py:1065,1094,1129; the hoisted input contractions of the context and the party GRU).  Forward
    is n launches as before; backward computes every dW_g / db_g in one grouped launch pair."""

    @staticmethod
    def forward(ctx, act, n, *args):
        force = False
        if isinstance(n, tuple):                 # (n, "hip"): every product of the node on the few-row kernel, never the library
            n, force = n[0], True
        xs, ws, bs = args[:n], args[n:2 * n], args[2 * n:3 * n]
        ys, saved = [], []
        # (an odd contraction width -- 1582-d audio, 342-d visual features -- runs on the zero-padded operands)
        x2s = [row_operand(x.reshape(-1, x.shape[-1])) for x in xs]
        wops = [weight_operand(w) for w in ws]
        ctx.force = force
        if force or (1 < n <= 8 and all(x2.shape[0] <= GROUP_ROWS and linear_group_supported(x2.shape[0], w.shape[1], w.shape[0])
                               and not linear_preferred(x2.shape[0], w.shape[1], w.shape[0]) for x2, w in zip(x2s, wops))):
            if force and not all(linear_group_supported(x2.shape[0], w.shape[1], w.shape[0]) for x2, w in zip(x2s, wops)):
                raise ValueError("linear_group(hip=True): a contraction wider than 768")
            # few rows (BASELINE cfg2 / cfg3 / cfg4: 1 056 .. 3 520): all projections of the group in ONE launch of the
            # few-row kernel (csrc/linear_small.hip) instead of n library GEMMs (cfg2: 16.7 us against 18.3 us for three
            # hipBLASLt launches, tools/bench_linear_group.py)
            outs = linear_group_raw([dict(x=x2, w=w, b=b) for x2, w, b in zip(x2s, wops, bs)], act)
        else:
            outs = [_linear_forward(x2, w, b, act) for x2, w, b in zip(x2s, wops, bs)]
        for x, x2, w, y in zip(xs, x2s, ws, outs):
            saved += [x2, w, y if act else None]
            ys.append(y.view(*x.shape[:-1], w.shape[0]))
        ctx.n, ctx.act = n, act
        ctx.has_bias = [b is not None for b in bs]
        ctx.param_refs = list(zip(ws, bs))
        ctx.save_for_backward(*saved)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n, sv = ctx.n, ctx.saved_tensors
        dxs, wg, dy2s = [], [], []
        for g in range(n):
            x2, w, y = sv[3 * g], sv[3 * g + 1], sv[3 * g + 2]
            dy2 = dys[g].reshape(-1, w.shape[0]) if dys[g] is not None else torch.zeros(x2.shape[0], w.shape[0], dtype=x2.dtype, device=x2.device)
            if ctx.act:
                dy2 = dy2 * (y > 0).to(dy2.dtype)
            dy2 = dy2.contiguous()
            dy2s.append(dy2)
            wg.append(_wgrad(dy2, x2, ctx.param_refs[g][0], ctx.param_refs[g][1]))
        if ctx.force:
            # input gradients dX_g = dY_g . W_g of the whole group in one launch: the weight is read as stored ((N, K) = the
            # K-major form of the product over N)
            need = [g for g in range(n) if ctx.needs_input_grad[2 + g]]
            outs = linear_group_raw([dict(x=dy2s[g], wk=weight_operand(sv[3 * g + 1])) for g in need]) if need else []
            dxs = [None] * n
            for g, o in zip(need, outs):
                K = sv[3 * g + 1].shape[1]
                dxs[g] = o[:, :K].reshape(*dys[g].shape[:-1], K) if dys[g] is not None else o[:, :K]
        else:
            for g in range(n):
                w = sv[3 * g + 1]
                dxs.append(_linear_dx(dy2s[g], weight_operand(w))[:, :w.shape[1]].reshape(*dys[g].shape[:-1], w.shape[1])
                           if ctx.needs_input_grad[2 + g] else None)
        return (None, None) + tuple(dxs) + tuple(r[0] for r in wg) + tuple(r[1] for r in wg)


def linear_group(xs, weights, biases, act=0, hip=False):
    """[act(x W^T + b) for each group] with all weight gradients computed by one grouped launch.  ``hip=True``: forward and
    input gradients on the few-row kernel whatever the shape (csrc/linear_small.hip; contraction <= 768): the fusion
    modules use it so that no library GEMM appears on their path."""
    for x in xs:
        _hip.require_cuda(x)
    n = len(xs)
    return list(_LinearGroup.apply(act, (n, "hip") if hip else n, *xs, *weights, *biases))


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


class _GateLinear(torch.autograd.Function):
    """G = q W_ih^T + h W_hh^T + (b_ih + b_hh): the pre-activation of the layer-shared LSTM cell (model_GCN.py:466,
    seq_len 1) as one op; h may be None (first layer: zero state).  ``bsum`` is b_ih + b_hh computed once per forward
    pass; the bias gradient goes to both parameters (they share it)."""

    @staticmethod
    def forward(ctx, q, h, w_ih, w_hh, bsum, b_ih, b_hh):
        G = _linear_forward(q, w_ih, bsum, 0)
        if h is not None:
            G = dense_nk(h, w_hh, None, 0, out=G, accumulate=True)
        ctx.has_h = h is not None
        ctx.refs = (w_ih, w_hh, b_ih, b_hh)
        ctx.save_for_backward(q, h, w_ih, w_hh)
        return G

    @staticmethod
    def backward(ctx, dG):
        q, h, w_ih, w_hh = ctx.saved_tensors
        p_ih, p_hh, b_ih, b_hh = ctx.refs
        dG = dG.contiguous()
        dq = _linear_dx(dG, w_ih) if ctx.needs_input_grad[0] else None
        dh = _linear_dx(dG, w_hh) if (ctx.has_h and ctx.needs_input_grad[1]) else None
        dwi = dwh = dbs = dbi = dbh = None
        M, N = dG.shape[1], q.shape[1]
        if _queueable(p_ih, [b_ih, b_hh], M, N) and (not ctx.has_h or _queueable(p_hh, [], M, h.shape[1])):
            queue_wgrad(dG, q, p_ih, [b_ih, b_hh])
            if ctx.has_h:
                queue_wgrad(dG, h, p_hh)
        else:
            dwi, dbi = _wgrad_inline(dG, q, True)
            dbh = dbi
            if ctx.has_h:
                dwh, _ = _wgrad_inline(dG, h, False)
        return dq, dh, dwi, dwh, dbs, dbi, dbh


def gate_linear(q, h, w_ih, w_hh, bsum, b_ih, b_hh):
    _hip.require_cuda(q, h)
    return _GateLinear.apply(q, h, w_ih, w_hh, bsum, b_ih, b_hh)


class _Linear2(torch.autograd.Function):
    """y = x [W1; W2]^T + [b1; b2]: one projection whose weight rows live in two parameters (the two directions of a
    bidirectional GRU layer, nn.GRU weight_ih_l*/ *_reverse) -- one launch on the parameters themselves instead of a
    concatenated copy per step (csrc/linear.hip, mmdfn_linear2)."""

    @staticmethod
    def forward(ctx, x, w1, w2, b1, b2, wcat, bcat):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        _hip.require_cuda(x2, w1, w2)
        _hip.require_f32(x2, w1, w2, b1, b2)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        w1c, w2c = w1.contiguous(), w2.contiguous()
        R, K = x2.shape
        n1, N = w1.shape[0], w1.shape[0] + w2.shape[0]
        if R < LINEAR2_FEW_ROWS:
            # few rows: the LDS-staged few-row kernel on the two parameters (csrc/linear_small.hip; 1 760 x 200 -> 600 in
            # 12.4 us against 16.0 us for the 64 x 64-tile kernel below, tools/bench_linear_group.py)
            y = linear_group_raw([dict(x=x2, w=w1c, w2=w2c, b=b1, b2=b2)])[0]
        else:
            y = torch.empty(R, N, dtype=torch.float32, device=x2.device)
            rc = _hip.lib().mmdfn_linear2(_hip.ptr(x2), _hip.ptr(w1c), _hip.ptr(w2c), n1, _hip.ptr(b1), _hip.ptr(b2), _hip.ptr(y),
                                          R, K, N, x2.stride(0), N, 0, 0, _hip.stream())
            _hip.check(rc, "mmdfn_linear2")
        ctx.refs = (w1, w2, b1, b2)
        ctx.save_for_backward(x2, w1c, w2c, wcat)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, wcat = ctx.saved_tensors
        p1, p2, b1, b2 = ctx.refs
        n1 = w1.shape[0]
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        d1, d2 = dy2[:, :n1], dy2[:, n1:]
        dx = None
        if ctx.needs_input_grad[0]:
            # the input gradient is one K-major launch of the few-row kernel over the stacked weight when the caller provides a
            # stacked view (gru._stacked_view), two accumulating launches on the parameters otherwise
            if wcat is not None and dy2.shape[0] >= 16384 and linear_preferred(dy2.shape[0], dy2.shape[1], w1.shape[1]):
                # very many rows (cfg3's 19 008 party rows: 54-59 us against 67 us for the K-major few-row form,
                # tools/bench_linear_group.py): the bf16-piece kernel on the transposed stacked weight
                dx = linear_raw(dy2, wcat.t().contiguous(), None, 0).view(*dy.shape[:-1], w1.shape[1])
            elif wcat is not None:
                dx = dense_kn(dy2, wcat).view(*dy.shape[:-1], w1.shape[1])
            else:
                # no stacked copy from the caller: two K-major products on the parameters, the second accumulating
                dx = linear_group_raw([dict(x=d1, wk=w1)])[0]
                linear_group_raw([dict(x=d2, wk=w2, out=dx, accumulate=True)])
                dx = dx.view(*dy.shape[:-1], w1.shape[1])
        dw1, db1 = _wgrad(d1, x2, p1, b1)
        dw2, db2 = _wgrad(d2, x2, p2, b2)
        return dx, dw1, dw2, db1, db2, None, None


LINEAR2_FEW_ROWS = 2048
GROUP_ROWS = 4096          # _LinearGroup: row count up to which a group of projections runs as one few-row launch


def linear2(x, w1, w2, b1, b2, wcat=None, bcat=None):
    """``wcat`` / ``bcat``: optional stacked views (or copies) of [w1; w2] (n1 + n2, K) and [b1; b2] -- no gradient flows
    through them; wcat serves the input gradient, and both serve the forward pass of launches with few rows."""
    if w1.shape[1] % 4 or w1.shape[1] < 4:
        raise ValueError("linear2: the contraction width must be a multiple of 4")
    return _Linear2.apply(x, w1, w2, b1, b2, wcat, bcat)


# ---- dropout keep flags: one generator launch per step -----------------------------------------------------------------
# Every dropout site of the fused path (GRU inter-layer dropout, the GCN stack, the head) consumes 0 / 1 keep flags that
# its kernel scales by 1/(1-p).  Inside a ``flag_pool()`` scope (the models open one per forward) the sites share one
# buffer per (device, p) drawn by ONE bernoulli_ launch: the first request of a step draws as many flags as the previous
# step with the same scope key used, later requests take slices (a request that does not fit draws its own buffer).
# A fresh tensor per draw: slices saved for backward are never overwritten.  Outside a scope every request draws its own.
_FLAG_SCOPE = None
_FLAG_HINT = {}


class flag_pool:
    def __init__(self, key=None):
        self.key = key
        self.bufs = {}       # (device, p) -> [buffer, offset, used]

    def __enter__(self):
        global _FLAG_SCOPE
        self.outer = _FLAG_SCOPE
        if self.outer is None:
            _FLAG_SCOPE = self
        return self

    def __exit__(self, *exc):
        global _FLAG_SCOPE
        if self.outer is None:
            _FLAG_SCOPE = None
            for k, (_, _, used) in self.bufs.items():
                if len(_FLAG_HINT) > 256:
                    _FLAG_HINT.clear()
                _FLAG_HINT[(self.key,) + k] = used
        return False


def keep_scale(p):
    """The factor dropout multiplies the kept elements by, 1 / (1 - p); p = 1 drops everything (all flags are 0), and the
    factor is 0 rather than inf so that 0 * inf never appears."""
    return 0.0 if p >= 1.0 else 1.0 / (1.0 - p)


_FLAG_STATE = {}       # device index -> [device state (seed, offset, workgroup counter), host mirror (seed, offset)]
_FLAG_CONSUMED = {}    # device index -> Philox counters consumed so far (eager and captured launches alike; host-side tally)


def flags_consumed(idx):
    return _FLAG_CONSUMED.get(idx, 0)


def flag_state_snapshot(idx):
    """(device state clone, host mirror) of the keep-flag generator, or None before the first draw on the device."""
    ent = _FLAG_STATE.get(idx)
    return None if ent is None else (ent[0].clone(), ent[1])


def flag_state_restore(idx, snap):
    """Put the keep-flag generator back where ``flag_state_snapshot`` found it (graphs.CapturedStep: building a captured
    step consumes no random numbers).  With no earlier state the device generator is re-seeded from torch's CUDA generator."""
    ent = _FLAG_STATE.get(idx)
    if ent is None:
        return
    if snap is None:
        ent[1] = None
        flag_state_sync(idx)
    else:
        ent[0].copy_(snap[0])
        ent[1] = snap[1]


def flag_state_sync(idx):
    """Re-seed the device generator from torch's CUDA generator if the two disagree (torch.manual_seed, a restored RNG
    state, torch's own random ops since the last draw).  Eager draws do this themselves; a captured step calls it before a
    replay, whose launches read the device state as it is."""
    ent = _FLAG_STATE.get(idx)
    if ent is None:
        return
    gen = torch.cuda.default_generators[idx]
    now = (int(gen.initial_seed()), int(gen.get_offset()))
    if ent[1] != now:
        seed = now[0] - (1 << 64) if now[0] >= (1 << 63) else now[0]
        ent[0].copy_(torch.tensor([seed, now[1], 0, 0], dtype=torch.int64), non_blocking=False)
        ent[1] = now


def flags_advance_host(idx, counters):
    """After a replay that consumed ``counters`` Philox counters on the device: move torch's generator (and the host mirror)
    by the same amount, so that the next eager draw continues the stream instead of re-seeding it backwards."""
    ent = _FLAG_STATE.get(idx)
    if ent is None or counters <= 0:
        return
    gen = torch.cuda.default_generators[idx]
    gen.set_offset(int(gen.get_offset()) + int(counters))
    ent[1] = (int(gen.initial_seed()), int(gen.get_offset()))


def draw_flags(n, p, device):
    """n (a multiple of 4) fresh fp32 keep flags from the package's Philox kernel (csrc/encoder_glue.hip).  The generator state
    lives on the device and every launch advances it, so replays of a captured graph draw new flags.  In eager mode the state
    follows torch's CUDA generator: it is re-seeded from (initial_seed, offset) whenever those differ from what this function
    left behind (torch.manual_seed, a restored RNG state, other random ops in between), and the generator's offset is advanced
    by the counters consumed -- `torch.manual_seed(s)` reproduces a run exactly as it does for torch's own dropout."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ent = _FLAG_STATE.get(idx)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None:
        if capturing:
            raise RuntimeError("the first dropout draw on a device cannot happen inside a stream capture (run one eager step first)")
        ent = _FLAG_STATE[idx] = [torch.zeros(4, dtype=torch.int64, device=device), None]
    n8 = ((n + 7) // 8 + 63) // 64 * 64    # Philox counters one launch consumes (8 flags each, whole waves)
    if not capturing:
        gen = torch.cuda.default_generators[idx]
        now = (int(gen.initial_seed()), int(gen.get_offset()))
        if ent[1] != now:
            seed = now[0] - (1 << 64) if now[0] >= (1 << 63) else now[0]
            ent[0].copy_(torch.tensor([seed, now[1], 0, 0], dtype=torch.int64), non_blocking=False)
        # leave torch's generator behind the counters this launch consumes (its offset moves in multiples of 4)
        gen.set_offset(now[1] + 4 * ((n8 + 3) // 4))
        ent[1] = (now[0], int(gen.get_offset()))
    _FLAG_CONSUMED[idx] = _FLAG_CONSUMED.get(idx, 0) + 4 * ((n8 + 3) // 4)
    out = torch.empty(n, dtype=torch.float32, device=device)
    _hip.check(_hip.lib().mmdfn_keep_flags(_hip.ptr(out), n, float(1.0 - p), _hip.ptr(ent[0]), _hip.stream()), "mmdfn_keep_flags")
    return out


def keep_flags(n, p, device):
    """n fp32 keep flags (1 with probability 1 - p), 16-byte aligned."""
    n = int(n)
    scope = _FLAG_SCOPE
    if scope is None:
        return draw_flags((n + 3) & ~3, p, device)[:n]
    k = (device, float(p))
    ent = scope.bufs.get(k)
    if ent is None:
        ent = scope.bufs[k] = [None, 0, 0]
    n4 = (n + 3) & ~3
    if ent[0] is None or ent[1] + n4 > ent[0].numel():
        want = max(n4, _FLAG_HINT.get((scope.key,) + k, 0) - ent[2])
        ent[0] = draw_flags(want, p, device)
        ent[1] = 0
    out = ent[0][ent[1]:ent[1] + n]
    ent[1] += n4
    ent[2] += n4
    return out


class _MaskScale(torch.autograd.Function):
    """outs[g] = xs[g] * masks[g] * scale for up to 4 tensors in ONE launch (csrc/encoder_glue.hip); the backward pass is
    the same launch on the incoming gradients.  masks: flat 0 / 1 keep flags (ops.keep_flags)."""

    @staticmethod
    def forward(ctx, scale, masks, *xs):
        _hip.require_cuda(*xs)
        _hip.require_f32(*xs, *masks)
        xs = [x.contiguous() for x in xs]
        outs = [torch.empty_like(x) for x in xs]
        ctx.masks, ctx.scale = list(masks), float(scale)
        _MaskScale._launch(xs, ctx.masks, outs, ctx.scale)
        return tuple(outs)

    @staticmethod
    def _launch(xs, masks, outs, scale):
        for x, m in zip(xs, masks):
            if m.numel() != x.numel() or x.numel() % 4 or x.data_ptr() % 16 or m.data_ptr() % 16:
                raise _hip.HipLibraryError("mask_scale: flags must match the tensor (multiple of 4 elements, 16-byte aligned)")
        rc = _hip.lib().mmdfn_mask_scale(len(xs), _hip.ptr_array(xs), _hip.ptr_array(masks), _hip.ptr_array(outs),
                                         _hip.long_array([x.numel() for x in xs]), scale, _hip.stream())
        _hip.check(rc, "mmdfn_mask_scale")

    @staticmethod
    def backward(ctx, *douts):
        live = [i for i, d in enumerate(douts) if d is not None]
        grads = [None] * len(douts)
        if live:
            ds = [douts[i].contiguous() for i in live]
            outs = [torch.empty_like(d) for d in ds]
            _MaskScale._launch(ds, [ctx.masks[i] for i in live], outs, ctx.scale)
            for i, o in zip(live, outs):
                grads[i] = o
        return (None, None) + tuple(grads)


def mask_scale(xs, masks, scale):
    """Dropout as a multiply by precomputed keep flags for a list of (<= 4) tensors, one launch each way."""
    if len(xs) > 4:
        return tuple(o for k in range(0, len(xs), 4) for o in mask_scale(xs[k:k + 4], masks[k:k + 4], scale))
    return _MaskScale.apply(scale, list(masks), *xs)


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


def matmul_kn(x, w):
    _hip.require_cuda(x)
    return _MatmulKN.apply(x, w)


def linear(x, weight, bias=None, act=0, base=None):
    """Drop-in for F.linear with optional fused ReLU and an optional addend (y = base + x W^T + b)."""
    _hip.require_cuda(x)
    return _Linear.apply(x, weight, bias, act, base)
