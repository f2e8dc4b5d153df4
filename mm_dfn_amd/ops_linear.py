"""Dense projections: raw launches, the dispatch per shape, weight piece planes, and the autograd nodes over them.

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import torch

from . import _hip
from .ops_pad import pad4, row_operand, weight_operand
from .ops_wgrad import _queueable, _wgrad, _wgrad_inline, queue_wgrad


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
        ctx.planes = R >= PLANES_MIN_ROWS and planes_supported(w1, w2)
        if ctx.planes:
            # the weight's bf16 piece planes, cut once per step (csrc/linear_planes.hip): 7 040 x 200 -> 600 in ~11 us against
            # 33 us for the 128 x 128-tile form below, which cuts both operands in every workgroup
            y = linear_planes_raw(x2, w1, w2, b1, b2)
        elif R < LINEAR2_FEW_ROWS:
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
            if ctx.planes:
                dx = linear_planes_raw(dy2, p1, p2, transposed=True).view(*dy.shape[:-1], w1.shape[1])
            elif wcat is not None and dy2.shape[0] >= 16384 and linear_preferred(dy2.shape[0], dy2.shape[1], w1.shape[1]):
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


# ---- weight piece planes (csrc/linear_planes.hip) -----------------------------------------------------------------------
# A weight changes once per optimizer step (reference run_train_erc.py:512), so the three exact bf16 pieces of the GRU input
# weights are cut ONCE per optimizer step into MFMA B-fragment order; the projection kernel then cuts only its own X rows.  An
# entry belongs to a pair of PARAMETER objects (the two directions of a bidirectional nn.GRU layer) and an orientation (forward
# product / input gradient).  An entry is FRESH while (a) the parameters' tensor versions and storages are the ones it was cut
# from -- torch optimizers, load_state_dict and friends write in place and bump ``_version`` -- and (b) nobody has called
# invalidate_planes() since (FlatAdam does after its kernel, whose writes autograd's counters do not see).  Stale entries are
# re-cut, grouped into one launch, wherever the planes are about to be read:
#   * refresh_planes()        -- a model's forward pass calls it first (eager steps);
#   * CapturedStep.replay()   -- for the entries its graph reads (the graph itself contains NO cut: a captured fwd + bwd step whose
#                                weights do not change between replays pays nothing; one that follows an optimizer step pays one
#                                grouped launch in front of the replay);
#   * weight_planes()         -- by itself, for an entry that is new or stale at the moment it is asked for.
PLANES_MIN_ROWS = int(__import__("os").environ.get("MMDFN_PLANES_MIN_ROWS", "4096"))   # (a huge value switches the form off: A/B aid)
_PLANES = {}
_PLANE_EPOCH = [0]
_PLANE_RECORDERS = []


class _PlaneEntry:
    __slots__ = ("w1", "w2", "mode", "N", "K", "n1", "buf", "epoch", "stamp", "__weakref__")


def _plane_stamp(w1, w2):
    return (w1._version, w1.data_ptr(), None if w2 is None else w2._version, None if w2 is None else w2.data_ptr())


def _plane_params(e):
    w1 = e.w1()
    w2 = None if e.w2 is None else e.w2()
    if w1 is None or (e.w2 is not None and w2 is None):
        return None
    return w1, w2


def _plane_fresh(e, prm):
    return e.epoch == _PLANE_EPOCH[0] and e.stamp == _plane_stamp(*prm)


def _cut_planes(entries):
    pa, ia = _hip.ptr_array, _hip.int_array
    live = [(e, _plane_params(e)) for e in entries]
    live = [(e, p[0], p[1]) for e, p in live if p is not None]
    for i in range(0, len(live), 16):
        chunk = live[i:i + 16]
        rc = _hip.lib().mmdfn_cut_weight_planes(
            len(chunk), pa([w1 for _, w1, _ in chunk]), pa([w2 for _, _, w2 in chunk]), ia([e.n1 for e, _, _ in chunk]),
            ia([w1.stride(0) for _, w1, _ in chunk]), ia([e.N for e, _, _ in chunk]), ia([e.K for e, _, _ in chunk]),
            ia([e.mode for e, _, _ in chunk]), pa([e.buf for e, _, _ in chunk]), _hip.stream())
        _hip.check(rc, "mmdfn_cut_weight_planes")
        for e, w1, w2 in chunk:
            e.epoch = _PLANE_EPOCH[0]
            e.stamp = _plane_stamp(w1, w2)


def planes_supported(w1, w2=None):
    ok = lambda w: (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1 and w.shape[1] % 4 == 0
                    and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0)
    return ok(w1) and (w2 is None or (ok(w2) and w2.shape[1] == w1.shape[1] and w2.stride(0) == w1.stride(0)))


def weight_planes(w1, w2=None, transposed=False, mode=None):
    """The piece-plane entry of a B operand read from the stored matrices w1 / w2 (``mode``: mmdfn_cut_weight_planes' -- 0 the
    stacked rows [w1; w2] as stored, 1 their transpose (``transposed=True``: the input gradient's operand), 2 / 3 two (K, .) matrices
    side by side, transposed (3: gate-interleaved contraction index); cut now if the entry is new or stale (see the section comment)."""
    import weakref
    if mode is None:
        mode = 1 if transposed else 0
    key = (id(w1), 0 if w2 is None else id(w2), int(mode))
    e = _PLANES.get(key)
    if e is not None and (e.w1() is not w1 or (w2 is not None and e.w2() is not w2)):
        e = None                                   # an id recycled by another tensor
    if e is None:
        rows = w1.shape[0] + (0 if w2 is None else w2.shape[0])
        cols = w1.shape[1]
        e = _PlaneEntry()
        e.w1, e.w2 = weakref.ref(w1), (None if w2 is None else weakref.ref(w2))
        e.mode = int(mode)
        if mode == 0:
            e.N, e.K, e.n1 = rows, cols, w1.shape[0]
        elif mode == 1:
            e.N, e.K, e.n1 = cols, rows, w1.shape[0]
        else:                                      # two (K, .) matrices side by side
            if w2 is not None and w2.shape[0] != w1.shape[0]:
                raise ValueError("weight_planes: modes 2 / 3 need matrices with the same number of rows")
            e.N, e.K, e.n1 = cols + (0 if w2 is None else w2.shape[1]), w1.shape[0], cols
        nbytes = _hip.lib().mmdfn_weight_planes_workspace(e.N, e.K)
        if nbytes <= 0:
            raise _hip.HipLibraryError("mmdfn_weight_planes_workspace refused N=%d K=%d" % (e.N, e.K))
        e.buf = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
        e.epoch, e.stamp = -1, None
        _PLANES[key] = e
    if not _plane_fresh(e, (w1, w2)):
        _cut_planes([e])
    for rec in _PLANE_RECORDERS:
        if e not in rec:
            rec.append(e)
    return e


def refresh_planes(entries=None):
    """Re-cut the stale ones among ``entries`` (default: every live entry) in one grouped launch per 16; entries whose
    parameters are gone are dropped from the registry."""
    if entries is None:
        if not _PLANES:
            return
        dead = [k for k, e in _PLANES.items() if _plane_params(e) is None]
        for k in dead:
            del _PLANES[k]
        entries = list(_PLANES.values())
    stale = []
    for e in entries:
        prm = _plane_params(e)
        if prm is not None and not _plane_fresh(e, prm):
            stale.append(e)
    if stale:
        _cut_planes(stale)


def invalidate_planes():
    """Weights were written by something autograd's version counters do not see (FlatAdam's kernel): every entry is stale until
    its next cut."""
    _PLANE_EPOCH[0] += 1


class planes_recording:
    """``with planes_recording() as used:`` -- every plane entry handed out inside the block is appended to ``used`` (what a
    captured step must re-check before each replay)."""

    def __enter__(self):
        self.used = []
        _PLANE_RECORDERS.append(self.used)
        return self.used

    def __exit__(self, *exc):
        _PLANE_RECORDERS.remove(self.used)
        return False


def linear_planes_raw(x2, w1, w2=None, b1=None, b2=None, transposed=False, act=0, out=None, accumulate=False):
    """act(x2 B^T + [b1; b2]) (+ out) with B = [w1; w2] (or its transpose) taken from its piece planes."""
    _hip.require_cuda(x2, w1)
    _hip.require_f32(x2, w1, w2, b1, b2, out)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    e = weight_planes(w1, w2, transposed)
    R, K = x2.shape
    if K != e.K:
        raise ValueError("linear_planes_raw: contraction width %d, planes were cut for %d" % (K, e.K))
    if out is None:
        out = torch.empty(R, e.N, dtype=torch.float32, device=x2.device)
    n1 = e.N if (transposed or w2 is None) else e.n1      # (bias split: only the forward orientation has two bias blocks)
    rc = _hip.lib().mmdfn_linear_planes(_hip.ptr(x2), _hip.ptr(e.buf), _hip.ptr(b1), _hip.ptr(b2), n1, _hip.ptr(out), R, K, e.N,
                                        x2.stride(0), out.stride(0), int(act), 1 if accumulate else 0, _hip.stream())
    _hip.check(rc, "mmdfn_linear_planes")
    return out


def linear_planes_group_raw(problems, transposed=False, act=0, mask_scale=1.0):
    """Up to four independent products x2 B^T + [b1; b2] against piece planes in ONE launch (mmdfn_linear_planes_group).
    problems: dicts with x (2-D), w1, w2 and optionally b1, b2, out, mask (keep flags with the OUTPUT's element count: the
    result is multiplied by mask * mask_scale); returns the outputs.  Every problem's result is
    bit-identical to its own linear_planes_raw launch (it keeps the tile form it takes alone)."""
    lib = _hip.lib()
    xs, ents, outs = [], [], []
    for pr in problems:
        x2 = pr["x"]
        _hip.require_cuda(x2, pr["w1"])
        _hip.require_f32(x2, pr["w1"], pr.get("w2"), pr.get("b1"), pr.get("b2"), pr.get("out"))
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        e = weight_planes(pr["w1"], pr.get("w2"), transposed)
        if x2.shape[1] != e.K:
            raise ValueError("linear_planes_group_raw: contraction width %d, planes were cut for %d" % (x2.shape[1], e.K))
        out = pr.get("out")
        if out is None:
            out = torch.empty(x2.shape[0], e.N, dtype=torch.float32, device=x2.device)
        mk = pr.get("mask")
        if mk is not None and (mk.numel() != out.numel() or mk.dtype != torch.float32 or not mk.is_contiguous() or out.stride(0) != e.N):
            raise _hip.HipLibraryError("linear_planes_group_raw: mask must be contiguous fp32 flags of the (contiguous) output's size")
        xs.append(x2)
        ents.append(e)
        outs.append(out)
    for i in range(0, len(problems), 4):
        sl = slice(i, i + 4)
        prs, x4, e4, o4 = problems[sl], xs[sl], ents[sl], outs[sl]
        n1 = [e.N if (transposed or pr.get("w2") is None) else e.n1 for pr, e in zip(prs, e4)]
        rc = lib.mmdfn_linear_planes_group(len(prs), _hip.ptr_array(x4), _hip.ptr_array([e.buf for e in e4]),
                                           _hip.ptr_array([pr.get("b1") for pr in prs]), _hip.ptr_array([pr.get("b2") for pr in prs]),
                                           _hip.int_array(n1), _hip.ptr_array(o4), _hip.int_array([x.shape[0] for x in x4]),
                                           _hip.int_array([e.K for e in e4]), _hip.int_array([e.N for e in e4]),
                                           _hip.int_array([x.stride(0) for x in x4]), _hip.int_array([o.stride(0) for o in o4]),
                                           int(act), 0, _hip.ptr_array([pr.get("mask") for pr in prs]), float(mask_scale),
                                           _hip.stream())
        _hip.check(rc, "mmdfn_linear_planes_group")
    return outs


class _Linear2Group(torch.autograd.Function):
    """G projections y_g = x_g [W1_g; W2_g]^T + [b1_g; b2_g] that do not depend on each other (the context and the party GRU's
    hoisted input contractions of one layer) as ONE node: one grouped launch against the weights' piece planes forward, one for
    the input gradients; weight / bias gradients per group as in _Linear2.  args: (x, w1, w2, b1, b2) per group.
    ``masks`` (optional, keep flags per group) / ``scale``: the inputs pass through a dropout first (nn.GRU's between its layers):
    x_g * mask_g * scale is formed by ONE launch for all groups here, and the dropout's backward is the input-gradient launch's
    epilogue instead of a launch of its own."""

    @staticmethod
    def forward(ctx, n, masks, scale, *args):
        grp = [args[5 * g:5 * g + 5] for g in range(n)]
        shapes, x2s = [], []
        if masks is not None:
            from .ops_flags import _MaskScale
            xin = [x.contiguous() for x, *_ in grp]
            dropped = [torch.empty_like(x) for x in xin]
            _MaskScale._launch(xin, list(masks), dropped, float(scale))
        for g, (x, w1, w2, b1, b2) in enumerate(grp):
            shapes.append(x.shape)
            x2 = (dropped[g] if masks is not None else x).reshape(-1, x.shape[-1])
            if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
                x2 = x2.contiguous()
            x2s.append(x2)
        ys = linear_planes_group_raw([dict(x=x2, w1=w1, w2=w2, b1=b1, b2=b2) for x2, (_, w1, w2, b1, b2) in zip(x2s, grp)])
        ctx.n = n
        ctx.masks = None if masks is None else list(masks)
        ctx.scale = float(scale)
        ctx.refs = [(w1, w2, b1, b2) for _, w1, w2, b1, b2 in grp]
        ctx.save_for_backward(*x2s)
        return tuple(y.view(*shp[:-1], y.shape[1]) for y, shp in zip(ys, shapes))

    @staticmethod
    def backward(ctx, *dys):
        x2s = ctx.saved_tensors
        n = ctx.n
        dy2s = []
        for g in range(n):
            dy = dys[g]
            if dy is None:
                N = ctx.refs[g][0].shape[0] + ctx.refs[g][1].shape[0]
                dy = torch.zeros(x2s[g].shape[0], N, dtype=torch.float32, device=x2s[g].device)
            dy2s.append(dy.reshape(-1, dy.shape[-1]).contiguous())
        need = [g for g in range(n) if ctx.needs_input_grad[3 + 5 * g]]
        dxs = [None] * n
        if need:
            got = linear_planes_group_raw([dict(x=dy2s[g], w1=ctx.refs[g][0], w2=ctx.refs[g][1],
                                                mask=None if ctx.masks is None else ctx.masks[g]) for g in need],
                                          transposed=True, mask_scale=ctx.scale)
            for g, dx in zip(need, got):
                dxs[g] = dx
        out = [None, None, None]
        for g in range(n):
            p1, p2, b1, b2 = ctx.refs[g]
            n1 = p1.shape[0]
            d1, d2 = dy2s[g][:, :n1], dy2s[g][:, n1:]
            dw1, db1 = _wgrad(d1, x2s[g], p1, b1)
            dw2, db2 = _wgrad(d2, x2s[g], p2, b2)
            dx = dxs[g]
            if dx is not None:
                lead = dys[g].shape[:-1] if dys[g] is not None else (x2s[g].shape[0],)
                dx = dx.view(*lead, p1.shape[1])
            out += [dx, dw1, dw2, db1, db2]
        return tuple(out)


def linear2_group(groups, masks=None, scale=1.0):
    """groups: [(x, w1, w2, b1, b2), ...] -> [y, ...]; one launch each way when every group's weights have piece planes and the
    launch as a whole has the rows the plane form wants; None otherwise (the caller runs one linear2 per group).  ``masks`` /
    ``scale``: keep flags of a dropout applied to the inputs first (see _Linear2Group)."""
    rows = sum(int(x.numel() // x.shape[-1]) for x, *_ in groups)
    if (len(groups) >= 2 and rows >= PLANES_MIN_ROWS and all(x.is_cuda and x.dtype == torch.float32 for x, *_ in groups)
            and all(planes_supported(w1, w2) and b1 is not None and b2 is not None for _, w1, w2, b1, b2 in groups)
            and (masks is None or all(m.numel() == x.numel() and x.numel() % 4 == 0 for m, (x, *_) in zip(masks, groups)))):
        flat = [t for grp in groups for t in grp]
        return list(_Linear2Group.apply(len(groups), None if masks is None else list(masks), float(scale), *flat))
    return None


LINEAR2_FEW_ROWS = 2048
GROUP_ROWS = 4096          # _LinearGroup: row count up to which a group of projections runs as one few-row launch


def linear2(x, w1, w2, b1, b2, wcat=None, bcat=None):
    """``wcat`` / ``bcat``: optional stacked views (or copies) of [w1; w2] (n1 + n2, K) and [b1; b2] -- no gradient flows
    through them; wcat serves the input gradient, and both serve the forward pass of launches with few rows."""
    if w1.shape[1] % 4 or w1.shape[1] < 4:
        raise ValueError("linear2: the contraction width must be a multiple of 4")
    return _Linear2.apply(x, w1, w2, b1, b2, wcat, bcat)


def matmul_kn(x, w):
    _hip.require_cuda(x)
    return _MatmulKN.apply(x, w)


def linear(x, weight, bias=None, act=0, base=None):
    """Drop-in for F.linear with optional fused ReLU and an optional addend (y = base + x W^T + b)."""
    _hip.require_cuda(x)
    return _Linear.apply(x, weight, bias, act, base)
