"""Speaker-party encoder glue (K3 / K4): gather, project-then-gather, scatter-combine.

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import torch

from . import _hip
from . import ops_linear
from .ops_linear import _wgrad, linear_group_raw
from .ops_wgrad import _WGQ, _queueable, _wgrad_inline, colsum, flush_queued_wgrads, queue_wgrad, slab_reduce_queueable


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


class _HalvesGrad:
    """Stand-in 'parameter' of queue_slab_reduce for two biases that share one column-sum stack: its .grad is the (N,) buffer
    whose halves were handed to the two biases."""

    def __init__(self, buf):
        self.grad = buf


def _queue_bias_halves(A, b1, b2, n1, slabs=None):
    """b1.grad, b2.grad = halves of the column sums of A (R, N): the slab kernel runs now (or has run: ``slabs`` = (workspace,
    number of slabs) from a launch that produced them on the way), the sum over slabs rides on the backward pass's last
    reduction launch."""
    _hip.require_cuda(A)
    _hip.require_f32(A)
    if A.stride(1) != 1:
        A = A.contiguous()
    R, N = A.shape
    lib = _hip.lib()
    if slabs is not None:
        ws, nsl = slabs
    else:
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
    def forward(ctx, qmask, w1, w2, b1, b2, wcat, bcat, riders, *rest):
        # riders: [(index into Xs, has_wcat), ...]; rest = 5 tensors per rider (w1, w2, b1, b2, wcat-or-None), then Xs.
        # A rider is ANOTHER projection of one of the gathered inputs that does not depend on this node's result (the context
        # GRU's first-layer input contraction of the same utterance rows, model.py:1132): it rides in the grouped launch of the
        # party projections instead of taking a launch of its own.  Its result comes back as an extra output.
        nr = len(riders)
        rprm = [rest[5 * i:5 * i + 5] for i in range(nr)]
        Xs = rest[5 * nr:]
        _hip.require_cuda(qmask, w1, w2, *Xs)
        mods = [x.contiguous() for x in Xs]
        qmask = qmask.contiguous()
        L, B, P = qmask.shape
        H = mods[0].shape[-1]
        Mn = len(mods)
        w1c, w2c = w1.contiguous(), w2.contiguous()
        n1, N = w1.shape[0], w1.shape[0] + w2.shape[0]
        G = torch.empty(Mn, L * B, N, dtype=torch.float32, device=qmask.device)
        routs = [torch.empty(L * B, pr[0].shape[0] + pr[1].shape[0], dtype=torch.float32, device=qmask.device) for pr in rprm]
        nprob = Mn + nr
        if (nprob <= 4 and nprob * L * B >= ops_linear.PLANES_MIN_ROWS and ops_linear.planes_supported(w1, w2)
                and all(ops_linear.planes_supported(pr[0], pr[1]) for pr in rprm)):
            # enough rows for the plane form: the projections against the weights' piece planes (cut once per optimizer step),
            # one grouped launch (4 x 1 760 rows at cfg2: 21 us against 25 for the few-row kernel)
            probs = [dict(x=m.view(L * B, H), w1=w1, w2=w2, out=G[i]) for i, m in enumerate(mods)]
            for (src, _), (rw1, rw2, rb1, rb2, _rw), o in zip(riders, rprm, routs):
                probs.append(dict(x=mods[src].view(L * B, H), w1=rw1, w2=rw2, b1=rb1, b2=rb2, out=o))
            ops_linear.linear_planes_group_raw(probs)
        else:
            probs = [dict(x=m.view(L * B, H), w=w1c, w2=w2c, out=G[i]) for i, m in enumerate(mods)]
            for (src, _), (rw1, rw2, rb1, rb2, _rw), o in zip(riders, rprm, routs):
                probs.append(dict(x=mods[src].view(L * B, H), w=rw1.contiguous(), w2=rw2.contiguous(), b=rb1, b2=rb2, out=o))
            linear_group_raw(probs)
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
        ctx.riders = [(src, tuple(pr[:4])) for (src, _), pr in zip(riders, rprm)]
        ctx.save_for_backward(rank, w1c, w2c, wcat, *[pr[4] for pr in rprm], *[pr[0] for pr in rprm], *[pr[1] for pr in rprm], *mods)
        ctx.mark_non_differentiable(rank)
        ctx.set_materialize_grads(False)
        return (S, rank) + tuple(Xs) + tuple(o.view(L, B, -1) for o in routs)

    @staticmethod
    def backward(ctx, dS, _drank, *dpass):
        nr = len(ctx.riders)
        rank, w1, w2, wcat, *tail = ctx.saved_tensors
        rwcat, rw1c, rw2c, mods = tail[:nr], tail[nr:2 * nr], tail[2 * nr:3 * nr], tail[3 * nr:]
        p1, p2, b1, b2 = ctx.refs
        L, B, P, H, Mn, n1, N = ctx.dims
        dride = dpass[Mn:]                      # gradients of the riders' outputs
        dpass = dpass[:Mn]
        dev = rank.device
        dS = dS.contiguous() if dS is not None else torch.zeros(L, Mn * B * P, N, dtype=torch.float32, device=dev)
        dG = torch.empty(Mn, L * B, N, dtype=torch.float32, device=dev)
        want_db = b1 is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        # the column sums stay slab stacks; the step's last reduction launch sums them into ONE (N,) buffer of which the two
        # biases' .grad are the halves (views handed over here, filled at the end of the backward pass)
        queue_db = (want_db and ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and slab_reduce_queueable(None, [b1, b2])
                    and b1.grad is None and b2.grad is None)
        lib = _hip.lib()
        if queue_db:
            # the scatter and the column-sum slabs read the same dS: one launch
            ws = torch.empty(int(lib.mmdfn_colsum_workspace(N)), dtype=torch.float32, device=dev)
            nsl = lib.mmdfn_party_gather_bwd_colsum(Mn, _hip.ptr(dS), _hip.ptr(rank), _hip.ptr_array([dG[m] for m in range(Mn)]),
                                                    None, L, B, P, N, _hip.ptr(ws), _hip.stream())
            if nsl <= 0:
                raise _hip.HipLibraryError("mmdfn_party_gather_bwd_colsum rejected the operands (%d)" % nsl)
            _queue_bias_halves(dS.view(-1, N), b1, b2, n1, slabs=(ws, int(nsl)))
        else:
            rc = lib.mmdfn_party_gather_bwd(Mn, _hip.ptr(dS), _hip.ptr(rank), _hip.ptr_array([dG[m] for m in range(Mn)]),
                                            None, L, B, P, N, _hip.stream())
            _hip.check(rc, "mmdfn_party_gather_bwd")
        db1 = db2 = None
        if want_db and not queue_db:
            db = colsum(dS.view(-1, N))
            db1, db2 = db[:n1], db[n1:]
        # input gradients: dX_m = dG_m [W1; W2] (+ the gradient that reached X_m's alias), one grouped launch
        dXs = [None] * Mn
        need = [m for m in range(Mn) if ctx.needs_input_grad[8 + 5 * nr + m]]
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
        # riders: their input gradient is added onto the source modality's (second launch: two workgroups of one launch may not
        # write the same rows), their weight / bias gradients are queued like any projection's
        rgrads = []
        for i, (src, (rp1, rp2, rb1, rb2)) in enumerate(ctx.riders):
            dyr = dride[i] if i < len(dride) else None
            if dyr is None:
                rgrads += [None, None, None, None, None]
                continue
            dy2 = dyr.reshape(L * B, -1).contiguous()
            rn1 = rp1.shape[0]
            if ctx.needs_input_grad[8 + 5 * nr + src]:
                if dXs[src] is None:
                    dXs[src] = torch.zeros(L, B, H, dtype=torch.float32, device=dev)
                tgt = dXs[src].view(L * B, H)
                if rwcat[i] is not None:
                    linear_group_raw([dict(x=dy2, wk=rwcat[i], out=tgt, accumulate=True)])
                else:
                    linear_group_raw([dict(x=dy2[:, :rn1], wk=rw1c[i], out=tgt, accumulate=True)])
                    linear_group_raw([dict(x=dy2[:, rn1:], wk=rw2c[i], out=tgt, accumulate=True)])
            xs2 = mods[src].view(L * B, H)
            dwa, dba = _wgrad(dy2[:, :rn1], xs2, rp1, rb1)
            dwb, dbb = _wgrad(dy2[:, rn1:], xs2, rp2, rb2)
            rgrads += [dwa, dwb, dba, dbb, None]
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
        return (None, dw1, dw2, db1, db2, None, None, None) + tuple(rgrads) + tuple(dXs)


def project_gather(Xs, qmask, w1, w2, b1, b2, wcat=None, bcat=None, riders=()):
    """(gi_p, rank, X_0', .., [rider outputs]): see _ProjectGather.  ``wcat`` / ``bcat``: optional stacked views of [w1; w2] /
    [b1; b2] (no gradient flows through them; without wcat the input gradient takes two launches per modality, without bcat the
    bias is concatenated per call).  ``riders``: [(index into Xs, w1, w2, b1, b2, wcat-or-None), ...] -- further projections
    X_index [w1; w2]^T + [b1; b2] computed in the same grouped launch, returned behind the identities as (L, B, .) tensors."""
    if w1.shape[1] % 4 or (w1.shape[0] + w2.shape[0]) % 4:
        raise ValueError("project_gather: widths must be multiples of 4")
    meta = [(int(r[0]), r[5] is not None) for r in riders]
    flat = [t for r in riders for t in r[1:6]]
    return _ProjectGather.apply(qmask, w1, w2, b1, b2, wcat, bcat, meta, *flat, *Xs)


class _PartyCombine(torch.autograd.Function):
    """out (Mn, N, H) = strip_pad(base_m + w_m * scatter(E)); E may be None (no speaker encoder), else it holds one
    (B*P)-column block per modality with a NON-ZERO weight, in modality order."""

    @staticmethod
    def forward(ctx, E, rank, flat_idx, weights, inv, *bases):
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
        ctx.save_for_backward(rank, flat_idx, inv)
        return out

    @staticmethod
    def backward(ctx, dout):
        rank, flat_idx, inv = ctx.saved_tensors
        L, B, P, H, Mn, N = ctx.dims
        dout = dout.contiguous()
        nact = sum(1 for w in ctx.weights[:Mn] if w != 0.0)
        nb, ne = Mn * L * B * H, (L * nact * B * P * H if ctx.has_E else 0)
        rc = -2
        if inv is not None:
            # destination-driven form: every element of dbase / dE is written by the kernel, no fill launch in front of it
            buf = torch.empty(nb + ne, dtype=torch.float32, device=dout.device)
            dbase = buf[:nb].view(Mn, L, B, H)
            dE = buf[nb:].view(L, nact * B * P, H) if ctx.has_E else None
            rc = _hip.lib().mmdfn_party_combine_bwd_dst(Mn, _hip.ptr(dout), _hip.ptr(rank), _hip.ptr(inv),
                                                        _hip.ptr_array([dbase[m] for m in range(Mn)]), _hip.ptr(dE),
                                                        _hip.float_array(ctx.weights), L, B, P, N, H, _hip.stream())
            if rc != -2:
                _hip.check(rc, "mmdfn_party_combine_bwd_dst")
        if rc == -2:
            zero = torch.zeros(nb + ne, dtype=torch.float32, device=dout.device)       # one fill for both (pad rows stay 0)
            dbase = zero[:nb].view(Mn, L, B, H)
            dE = zero[nb:].view(L, nact * B * P, H) if ctx.has_E else None
            rc = _hip.lib().mmdfn_party_combine_bwd(Mn, _hip.ptr(dout), _hip.ptr(rank), _hip.ptr(flat_idx),
                                                    _hip.ptr_array([dbase[m] for m in range(Mn)]), _hip.ptr(dE),
                                                    _hip.float_array(ctx.weights), L, B, P, N, H, _hip.stream())
            _hip.check(rc, "mmdfn_party_combine_bwd")
        return (dE, None, None, None, None) + tuple(dbase[m] for m in range(Mn))


def party_combine(bases, E, rank, flat_idx, weights, inv=None):
    """``inv`` (optional, (L * B) int64): the inverse of flat_idx (row of (t, b) in the stripped order or -1); with it the backward
    pass writes its outputs destination by destination and needs no zero fill."""
    return _PartyCombine.apply(E, rank, flat_idx, list(weights), inv, *bases)
