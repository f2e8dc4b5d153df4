"""Small helpers shared by the operator modules + ROW PADDING of contraction operands whose width is not a multiple of 4.

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import torch

from . import _hip


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
