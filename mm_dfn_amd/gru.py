"""2-layer bidirectional GRU encoders on the fused HIP recurrence (csrc/gru.hip).

``bigru2(xs, grus, ...)`` evaluates several *independent* nn.GRU(200, 100, 2 layers, bidirectional)
modules (the reference's ``lstm_l`` and ``rnn_parties``, model.py:866,868) layer by layer:
per layer ONE dense input GEMM per module (all timesteps, both directions) and ONE persistent
recurrence launch shared by all modules.  The nn.GRU modules only hold the parameters (so the
state_dict keys stay the reference's); their own forward (MIOpen) is never called.
"""
import torch

from . import _hip, ops

H = 100


# Valid-length truncation of the speaker-party batch (csrc/gru.hip, "Valid-length truncation"; include/mmdfn_hip.h,
# mmdfn_gru_seq_fwd_seg): "auto" uses the segmented launches when the plain form would need more workgroups than the chip
# has CUs (every sequence-direction is one persistent workgroup; with a CU per chain the launch lasts as long as its longest
# chain whatever the others do, so truncation buys nothing below that), True / False force it (tests, A/B runs).
TRUNCATE = {"0": False, "1": True}.get(__import__("os").environ.get("MMDFN_GRU_TRUNCATE", "auto"), "auto")
CUS = 256
# Layer 1's reverse direction can be truncated as well, against the all-padding sequence (start_party_table: one more
# workgroup chain per step on a side stream).  Inside a captured step every fork / join of the side branch costs ~10 us of
# the main chain (measured, profiles/r05_gru_valid_length.md), which is what the shorter layer-1 launch saves at the
# BASELINE batch sizes: off unless asked for (MMDFN_GRU_TABLE=1, or gru.USE_TABLE = True).
USE_TABLE = __import__("os").environ.get("MMDFN_GRU_TABLE", "0") == "1"
# Without the table layer 1 can still run on the segmented kernels just to skip the silent rows (k = 0), at the price of their
# bookkeeping on every other row: with a quarter of the rows silent (synthetic cfg3) that is a loss (1.252 vs 1.225 ms per step),
# so layer 1 stays on the plain launch by default; corpora whose dialogues involve few of many speakers (MELD: 2-3 of 9)
# are the case for MMDFN_GRU_L1_SEG=1.
L1_SKIPS_SILENT = __import__("os").environ.get("MMDFN_GRU_L1_SEG", "0") == "1"


MFMA_MIN_CHAINS = int(__import__("os").environ.get("MMDFN_GRU_MFMA_MIN", "1024"))    # (csrc/gru.hip: mfma_min_chains)


def wants_truncation(n_rows_total):
    """Valid-length launches pay when the chains they remove change the number of rounds of the one-sequence-per-workgroup
    kernels; launches with more than MFMA_MIN_CHAINS sequence-directions run the MFMA form (csrc/gru_mfma.hip: 16 sequences per
    workgroup, launch time independent of the batch) at full length instead."""
    if TRUNCATE == "auto":
        return CUS < 2 * n_rows_total <= MFMA_MIN_CHAINS
    return bool(TRUNCATE)


class _Seg:
    """Per-launch description of the party group for the segmented kernels: ``group`` index, the gather's ``rank`` array
    (L, B, P) int32, the truncated direction, whether the all-padding sequence's outputs are an input (layer 1)."""

    def __init__(self, group, rank, tdir):
        self.group, self.rank, self.tdir = group, rank, tdir
        self.P = rank.shape[2]
        self.BP = rank.shape[1] * rank.shape[2]

    def arrays(self, n):
        ranks = [self.rank if g == self.group else None for g in range(n)]
        P = [self.P if g == self.group else 1 for g in range(n)]
        BP = [self.BP if g == self.group else 1 for g in range(n)]
        tdir = [self.tdir if g == self.group else -1 for g in range(n)]
        return _hip.ptr_array(ranks), _hip.int_array(P), _hip.int_array(BP), _hip.int_array(tdir)


_FLAG_RIDER_P = [None]     # bigru2 -> _GruRecurrence.forward: the dropout rate whose flags the NEXT plain forward launch may carry


class _GruRecurrence(torch.autograd.Function):
    """args = (seg, ytab, then per group (gi, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev), flattened) -> (y_0, y_1, ...).  The
    recurrent weights are the module's own parameters (no stack / cat per step): the kernels take one pointer per direction
    and the gradients come back per parameter.  ``seg`` (a _Seg or None) selects the segmented launch; ``ytab`` (T, 1, 2H) or
    None: the all-padding sequence's outputs the truncated reverse direction starts from (its gradient is returned)."""

    @staticmethod
    def forward(ctx, seg, ytab, *args):
        n = len(args) // 5
        gis = [args[5 * g].contiguous() for g in range(n)]
        whh = [args[5 * g + 1 + d].contiguous() for g in range(n) for d in range(2)]
        bhh = [args[5 * g + 3 + d].contiguous() for g in range(n) for d in range(2)]
        _hip.require_cuda(*gis)
        _hip.require_f32(*gis, *whh, *bhh)
        ys, gates, rows, Ts = [], [], [], []
        for gi in gis:
            T, R = gi.shape[0], gi.shape[1]
            ys.append(torch.empty(T, R, 2 * H, dtype=torch.float32, device=gi.device))
            gates.append(torch.empty(T, R, 2, 4, H, dtype=torch.float32, device=gi.device))
            rows.append(R)
            Ts.append(T)
        if seg is None:
            # (the first layer of a training-mode bigru2: the step's dropout flags are drawn as riders of this launch)
            flag_p, _FLAG_RIDER_P[0] = _FLAG_RIDER_P[0], None
            if flag_p is not None:
                ops.stage_flag_draw(flag_p, gis[0].device, rows)
            try:
                rc = _hip.lib().mmdfn_gru_seq_fwd(n, _hip.ptr_array(gis), _hip.ptr_array(whh), _hip.ptr_array(bhh),
                                                  _hip.ptr_array(ys), _hip.ptr_array(gates), _hip.int_array(rows),
                                                  _hip.int_array(Ts), H, _hip.stream())
            finally:
                if flag_p is not None:
                    ops.finish_flag_draw()
            _hip.check(rc, "mmdfn_gru_seq_fwd")
        else:
            if ytab is not None:
                ytab = ytab.contiguous()
                _hip.require_f32(ytab)
                if tuple(ytab.shape) != (Ts[seg.group], 1, 2 * H):
                    raise ValueError("ytab must be (T, 1, 2H)")
            ytabs = [ytab if g == seg.group else None for g in range(n)]
            rk, P, BP, tdir = seg.arrays(n)
            rc = _hip.lib().mmdfn_gru_seq_fwd_seg(n, _hip.ptr_array(gis), _hip.ptr_array(whh), _hip.ptr_array(bhh),
                                                  _hip.ptr_array(ys), _hip.ptr_array(gates), _hip.int_array(rows),
                                                  _hip.int_array(Ts), H, rk, P, BP, tdir, _hip.ptr_array(ytabs), _hip.stream())
            _hip.check(rc, "mmdfn_gru_seq_fwd_seg")
        ctx.n = n
        ctx.seg = seg
        ctx.has_tab = ytab is not None
        ctx.refs = [args[5 * g + 1 + k] for g in range(n) for k in range(4)]   # the parameter objects (leaf test)
        ctx.save_for_backward(*ys, *gates, *whh)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        seg = ctx.seg
        saved = ctx.saved_tensors
        ys, gates, whh = saved[:n], saved[n:2 * n], saved[2 * n:]
        dys = [dy.contiguous() if dy is not None else torch.zeros_like(y) for dy, y in zip(dys, ys)]
        dgi = [torch.empty(y.shape[0], y.shape[1], 6 * H, dtype=torch.float32, device=y.device) for y in ys]
        dgh = [torch.empty_like(t) for t in dgi]
        rows = [y.shape[1] for y in ys]
        Ts = [y.shape[0] for y in ys]
        dyt = None
        if seg is None:
            # weight gradients queued so far ride on the CUs this recurrence leaves idle (ops_wgrad.stage_riders)
            ops.stage_riders(rows, Ts)
            try:
                rc = _hip.lib().mmdfn_gru_seq_bwd(n, _hip.ptr_array(dys), _hip.ptr_array(ys), _hip.ptr_array(gates),
                                                  _hip.ptr_array(whh), _hip.ptr_array(dgi), _hip.ptr_array(dgh),
                                                  _hip.int_array(rows), _hip.int_array(Ts), H, _hip.stream())
            finally:
                ops.finish_riders()
            _hip.check(rc, "mmdfn_gru_seq_bwd")
        else:
            g0 = seg.group
            dhinit = kout = None
            if ctx.has_tab:
                dhinit = torch.empty(rows[g0], H, dtype=torch.float32, device=ys[g0].device)
                kout = torch.empty(rows[g0], dtype=torch.int32, device=ys[g0].device)
            rk, P, BP, tdir = seg.arrays(n)
            rc = _hip.lib().mmdfn_gru_seq_bwd_seg(n, _hip.ptr_array(dys), _hip.ptr_array(ys), _hip.ptr_array(gates),
                                                  _hip.ptr_array(whh), _hip.ptr_array(dgi), _hip.ptr_array(dgh),
                                                  _hip.int_array(rows), _hip.int_array(Ts), H, rk, P, BP, tdir,
                                                  _hip.ptr_array([dhinit if g == g0 else None for g in range(n)]),
                                                  _hip.ptr_array([kout if g == g0 else None for g in range(n)]), _hip.stream())
            _hip.check(rc, "mmdfn_gru_seq_bwd_seg")
            if ctx.has_tab and ctx.needs_input_grad[1]:
                # what reaches the all-padding sequence from the rows truncated against it (its own backward follows)
                dyt = torch.empty(Ts[g0], 1, 2 * H, dtype=torch.float32, device=ys[g0].device)
                rc = _hip.lib().mmdfn_gru_tab_reduce(_hip.ptr(dys[g0]), _hip.ptr(kout), _hip.ptr(dhinit), _hip.ptr(dyt),
                                                     rows[g0], Ts[g0], H, seg.tdir, _hip.stream())
                _hip.check(rc, "mmdfn_gru_tab_reduce")
        # recurrent-weight gradients of every group and direction join the step's weight-gradient batch:
        #   dW_hh = sum_t dgh_t (x) h_{t-1}   forward: h_{t-1} = y[t-1] (row shift -R), reverse: y[t+1] (+R)
        #   db_hh = sum_t dgh_t                = the column sums of the same operand
        out = []
        for g in range(n):
            y, d = ys[g], dgh[g]
            T, R = y.shape[0], y.shape[1]
            d2, y2 = d.view(T * R, 6 * H), y.view(T * R, 2 * H)
            wf, wr, bf, br = ctx.refs[4 * g: 4 * g + 4]
            parts = ((d2[:, :3 * H], y2[:, :H], wf, bf, -R), (d2[:, 3 * H:], y2[:, H:], wr, br, R))
            res = []
            for A, B, w, b, shift in parts:
                if ops._queueable(w, [b], 3 * H, H):
                    ops.queue_wgrad(A, B, w, [b], shift)
                    res.append((None, None))
                else:
                    dw = torch.empty(3 * H, H, dtype=torch.float32, device=y.device)
                    db = torch.empty(3 * H, dtype=torch.float32, device=y.device)
                    ops.gemm_tn_grouped([dict(A=A, B=B, C=dw, colsum=db, shift=shift)])
                    res.append((dw, db))
            out += [dgi[g], res[0][0], res[1][0], res[0][1], res[1][1]]
        return (None, dyt) + tuple(out)


# ---------------------------------------------------------------------------------------------------
# The all-padding ("silent") party sequence: input zeros -> gi = b_ih at every step.  Its reverse-direction states are what
# every party sequence's reverse pass goes through before it reaches its own data (model.py:1076-1087 pads BEHIND the data),
# so layer 1 computes them ONCE per step: one extra one-row group on the unchanged kernels, launched on a side stream at the
# start of the encoders (it depends on the weights only: a single workgroup, T dependent steps, hidden behind the projections
# and the gather) and joined where the first recurrence launch needs it.  Its backward (the same kernels again, fed by
# mmdfn_gru_tab_reduce) runs on the side stream as well, next to the rest of the encoder backward.
# ---------------------------------------------------------------------------------------------------
_SIDE = {"stream": None, "active": False}


class _SideStream:
    """Fork: the side stream continues from ``after`` (an event recorded on the caller's stream; default: from everything
    the caller's stream has been given so far)."""

    def __init__(self, after=None):
        self.after = after

    def __enter__(self):
        if _SIDE["stream"] is None:
            _SIDE["stream"] = torch.cuda.Stream()
        self.side = _SIDE["stream"]
        if self.after is not None:
            self.side.wait_event(self.after)
        else:
            self.side.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        _SIDE["active"] = True
        return self.side

    def __exit__(self, *exc):
        _SIDE["active"] = False
        return self.ctx.__exit__(*exc)


class _GruTable(torch.autograd.Function):
    """(T, b_ih_fwd, b_ih_rev, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev) -> y_tab (T, 1, 2H): the all-padding sequence of the
    layer these parameters belong to, on the plain one-row kernels.  Runs on whatever stream is current (start_party_table:
    the side stream; autograd then runs the backward there too).  Only the reverse direction's states are ever read, so
    only its parameters receive gradients -- W_hh_rev, b_hh_rev and b_ih_rev -- and they do not travel through autograd:
    they are small in-line contractions here, handed to ops.add_grad_addends, which adds them to ``.grad`` at the end of the
    backward pass (the step's weight-gradient batch does not wait for this one-workgroup chain)."""

    @staticmethod
    def forward(ctx, T, params, bif, bir, wf, wr, bf, br):
        _hip.require_cuda(bif, bir, wf, wr, bf, br)
        bcat = _stacked_view(bif, bir) if not torch.cuda.is_current_stream_capturing() or _adjacent(bif, bir) else None
        src = bcat if bcat is not None else torch.cat([bif.detach(), bir.detach()])
        gi = src.expand(T, 1, 6 * H).contiguous()
        whh = [wf.detach().contiguous(), wr.detach().contiguous()]
        bhh = [bf.detach().contiguous(), br.detach().contiguous()]
        y = torch.empty(T, 1, 2 * H, dtype=torch.float32, device=gi.device)
        gates = torch.empty(T, 1, 2, 4, H, dtype=torch.float32, device=gi.device)
        rc = _hip.lib().mmdfn_gru_seq_fwd(1, _hip.ptr_array([gi]), _hip.ptr_array(whh), _hip.ptr_array(bhh),
                                          _hip.ptr_array([y]), _hip.ptr_array([gates]), _hip.int_array([1]),
                                          _hip.int_array([T]), H, _hip.stream())
        _hip.check(rc, "mmdfn_gru_seq_fwd")
        ctx.T = T
        ctx.params = params            # (b_ih_rev, w_hh_rev, b_hh_rev): the parameter objects themselves
        ctx.save_for_backward(y, gates, *whh)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, gates, wf, wr = ctx.saved_tensors
        T = ctx.T
        dy = dy.contiguous()
        dgi = torch.empty(T, 1, 6 * H, dtype=torch.float32, device=y.device)
        dgh = torch.empty_like(dgi)
        rc = _hip.lib().mmdfn_gru_seq_bwd(1, _hip.ptr_array([dy]), _hip.ptr_array([y]), _hip.ptr_array([gates]),
                                          _hip.ptr_array([wf, wr]), _hip.ptr_array([dgi]), _hip.ptr_array([dgh]),
                                          _hip.int_array([1]), _hip.int_array([T]), H, _hip.stream())
        _hip.check(rc, "mmdfn_gru_seq_bwd")
        bir, w, b = ctx.params
        d2, g2, y2 = dgh.view(T, 6 * H)[:, 3 * H:], dgi.view(T, 6 * H)[:, 3 * H:], y.view(T, 2 * H)[:, H:]
        dw = torch.empty(3 * H, H, dtype=torch.float32, device=y.device)
        scratch = torch.empty_like(dw)
        dbh = torch.empty(3 * H, dtype=torch.float32, device=y.device)
        dbi = torch.empty_like(dbh)
        # dW_hh = sum_t dgh_t (x) y[t+1] (reverse direction: h_{t-1} of step t is the output at t+1), db_hh / db_ih = the column
        # sums of dgh / dgi (the second problem's product is scratch: only its column sum is wanted)
        ops.gemm_tn_grouped([dict(A=d2, B=y2, C=dw, colsum=dbh, shift=1), dict(A=g2, B=y2, C=scratch, colsum=dbi, shift=1)])
        ops.add_grad_addends([(w, dw), (b, dbh), (bir, dbi)])
        return (None,) * 8


class PartyTable:
    """Handle of the all-padding sequence of one forward pass: ``y`` (T, 1, 2H) and the event behind its launch."""

    def __init__(self, y, event):
        self.y, self.event = y, event

    def join(self):
        """Called on the consumer's stream right before the launch that reads ``y``."""
        cur = torch.cuda.current_stream()
        cur.wait_event(self.event)
        self.y.record_stream(cur)
        return self.y


def start_party_table(gru, T, after=None):
    """Launch the all-padding sequence of ``gru``'s first layer (T steps) on the side stream, which continues from ``after``
    (an event on the caller's stream; default: from the work given to it so far); returns a PartyTable."""
    _, b_ih, hh = _layer_params(gru, 0)
    for prm in (b_ih[1], hh[1], hh[3]):
        if prm.requires_grad and not prm.is_leaf:
            raise NotImplementedError("the valid-length party launches write the all-padding sequence's gradients into .grad: "
                                      "leaf parameters only")
    # the node's tensor inputs are aliases made on the CALLER's stream: autograd creates a parameter's gradient accumulator
    # where the parameter is first used and runs it on the stream that was current there -- made under the side stream, the
    # accumulators of these six parameters would run on it, and the end of every backward pass would wait for the side stream
    alias = [prm.view_as(prm) for prm in (b_ih[0], b_ih[1], *hh)]
    with _SideStream(after) as side:
        y = _GruTable.apply(T, (b_ih[1], hh[1], hh[3]), *alias)
        ev = torch.cuda.Event()
        ev.record(side)
    return PartyTable(y, ev)


def _layer_params(gru, layer):
    """((w_ih_fwd, w_ih_rev), (b_ih_fwd, b_ih_rev), [w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev]): the module's own
    parameters, never concatenated."""
    sfx = "_l%d" % layer
    w_ih = (getattr(gru, "weight_ih" + sfx), getattr(gru, "weight_ih" + sfx + "_reverse"))
    b_ih = (getattr(gru, "bias_ih" + sfx), getattr(gru, "bias_ih" + sfx + "_reverse"))
    hh = [getattr(gru, "weight_hh" + sfx), getattr(gru, "weight_hh" + sfx + "_reverse"),
          getattr(gru, "bias_hh" + sfx), getattr(gru, "bias_hh" + sfx + "_reverse")]
    return w_ih, b_ih, hh


def _adjacent(wf, wr):
    return (wf.is_contiguous() and wr.is_contiguous() and wf.shape == wr.shape and wf.dtype == wr.dtype
            and wr.data_ptr() == wf.data_ptr() + wf.numel() * wf.element_size()
            and wf.untyped_storage().data_ptr() == wr.untyped_storage().data_ptr())


def _stacked_view(wf, wr):
    """[wf; wr] (2 * 3H, K) without a copy, or None.  Two leaf parameters that are not adjacent yet are moved into one
    buffer first (``.data`` re-pointed, values kept: anything that aliased the OLD storage -- an EMA copy made with
    ``.data`` views, a foreign flat-parameter optimizer, an earlier captured graph -- goes stale; ``pair_gru_weights(model)``
    does this once, explicitly, at model / optimizer construction time, which is where it belongs) unless an owner of
    their storage forbids it (FlatAdam marks the
    parameters it has laid out -- in an order that keeps the pairs adjacent, distributed.bucket_order) or the stream is
    being captured."""
    if not _adjacent(wf, wr):
        if (not wf.is_cuda or not wf.is_leaf or not wr.is_leaf or getattr(wf, "_mmdfn_flat", False)
                or getattr(wr, "_mmdfn_flat", False) or torch.cuda.is_current_stream_capturing()):
            return None
        with torch.no_grad():
            buf = torch.cat([wf.detach().reshape(-1), wr.detach().reshape(-1)])
            wf.data = buf[:wf.numel()].view(wf.shape)
            wr.data = buf[wf.numel():].view(wr.shape)
    if wf.dim() == 1:
        return torch.as_strided(wf.detach(), (2 * wf.shape[0],), (1,))
    return torch.as_strided(wf.detach(), (2 * wf.shape[0], wf.shape[1]), (wf.shape[1], 1))


def pair_gru_weights(module):
    """Lay the two directions' ``weight_ih`` / ``bias_ih`` of every fused nn.GRU of ``module`` out adjacently NOW (what the
    first training forward would otherwise do lazily as a side effect): call it once after building / loading the model
    and before anything takes aliases of the parameter storage.  Values are unchanged.  Returns the number of pairs."""
    n = 0
    for sub in module.modules():
        if isinstance(sub, torch.nn.GRU) and sub.bidirectional and sub.num_layers == 2 and sub.hidden_size == H:
            for layer in range(2):
                w_ih, b_ih, _ = _layer_params(sub, layer)
                n += int(_stacked_view(*w_ih) is not None) + int(_stacked_view(*b_ih) is not None)
    return n


def bigru2(xs, grus, dropout=0.0, training=False, gi0=None, party=None):
    """xs[g]: (T, rows_g, 200) -> ys[g]: (T, rows_g, 200); grus[g]: the nn.GRU holding group g's weights.
    gi0[g] (optional): the first layer's gate pre-activations X W_ih^T + b_ih (T, rows_g, 600) computed by the caller
    (the party encoder projects the L*B utterances once and gathers the result instead of projecting the L*P*B
    party rows); xs[g] is then ignored.
    party = (group index, rank (L, B, P) int32, PartyTable or None): run that group with the valid-length launches (layer 1:
    reverse direction truncated against the all-padding sequence when one is given; layer 2: forward direction truncated;
    silent rows skipped)."""
    for gru in grus:
        if gru.hidden_size != H or gru.num_layers != 2 or not gru.bidirectional or gru.batch_first:
            raise NotImplementedError("fused GRU path supports nn.GRU(*, 100, num_layers=2, bidirectional=True)")
        if gru.input_size != 2 * H:
            # (both layers then share one (600, 200) stacked-weight shape: the copy fallback below relies on it)
            raise NotImplementedError("fused GRU path supports input_size == 2 * hidden_size (200) only, got %d" % gru.input_size)
    cur = list(xs)
    # [W_ih_fwd; W_ih_rev] of every (module, layer) as ONE (600, K) operand for the input-gradient GEMMs of the backward
    # pass.  The two parameters are kept adjacent in memory (_stacked_view), so this is a view; parameters some
    # other owner has laid out differently are copied instead (one multi-tensor copy launch per step).
    wcat = bcat = None
    if torch.is_grad_enabled():
        pairs = [_layer_params(gru, layer)[0] for layer in range(2) for gru in grus]
        wcat = [_stacked_view(wf, wr) for wf, wr in pairs]
        # the stacked input biases [b_ih; b_ih_reverse] the same way (views only)
        bcat = [_stacked_view(bf, br) for bf, br in (_layer_params(gru, layer)[1] for layer in range(2) for gru in grus)]
        if any(w is None for w in wcat):
            halves = [w for pair in pairs for w in pair]
            with torch.no_grad():
                wcat = torch.empty(len(pairs), 2 * halves[0].shape[0], halves[0].shape[1], dtype=halves[0].dtype,
                                   device=halves[0].device)
                torch._foreach_copy_([wcat[i // 2, (i % 2) * halves[0].shape[0]:(i % 2 + 1) * halves[0].shape[0]]
                                      for i in range(len(halves))], halves)
    pending = None                  # (keep flags, scale) of the dropout in front of the next layer
    for layer in range(2):
        prm = [_layer_params(gru, layer) for gru in grus]
        # hoisted input contractions (all t, both directions) of every group: one launch per group on the two
        # directions' own weight_ih / bias_ih parameters
        pre = gi0 if (layer == 0 and gi0 is not None) else [None] * len(grus)
        todo = [g for g in range(len(grus)) if pre[g] is None]
        gis = list(pre)
        # the groups' contractions do not depend on each other: ONE launch against the weights' piece planes when there are
        # several (the context and the party encoder's second layer: 1 760 + 7 040 rows at cfg2), each way
        joint = None
        if len(todo) >= 2:
            # (the dropout between the layers rides along: its forward is one launch inside the node, its backward the input
            # gradient launch's epilogue)
            pm = None if pending is None or len(todo) != len(grus) else pending[0]
            joint = ops.linear2_group([(cur[g], prm[g][0][0], prm[g][0][1], prm[g][1][0], prm[g][1][1]) for g in todo],
                                      masks=pm, scale=1.0 if pm is None else pending[1])
            if joint is not None and pm is not None:
                pending = None
        if pending is not None:
            cur = list(ops.mask_scale(cur, pending[0], pending[1]))
            pending = None
        for k, g in enumerate(todo):
            gis[g] = joint[k] if joint is not None else ops.linear2(
                cur[g], prm[g][0][0], prm[g][0][1], prm[g][1][0], prm[g][1][1],
                None if wcat is None else wcat[layer * len(grus) + g], None if bcat is None else bcat[layer * len(grus) + g])
        args = []
        for gi, p in zip(gis, prm):
            args += [gi] + p[2]
        _FLAG_RIDER_P[0] = dropout if (layer == 0 and training and dropout > 0) else None
        if party is None:
            cur = list(_GruRecurrence.apply(None, None, *args))
        elif layer == 0 and party[2] is None and not L1_SKIPS_SILENT:
            cur = list(_GruRecurrence.apply(None, None, *args))
        elif layer == 0 and party[2] is None:
            # no all-padding sequence at hand: layer 1 runs both directions at full length, silent rows are skipped
            cur = list(_GruRecurrence.apply(_Seg(party[0], party[1], -1), None, *args))
        elif layer == 0:
            cur = list(_GruRecurrence.apply(_Seg(party[0], party[1], 1), party[2].join(), *args))
        else:
            cur = list(_GruRecurrence.apply(_Seg(party[0], party[1], 0), None, *args))
        _FLAG_RIDER_P[0] = None
        if layer == 0 and training and dropout > 0:
            # nn.GRU's dropout between the layers: 0 / 1 keep flags from the step's flag pool (no generator launch of its
            # own) applied to every group's output by ONE launch each way -- at the head of the next layer (see there)
            pending = ([ops.keep_flags(y.numel(), dropout, y.device) for y in cur], ops.keep_scale(dropout))
    return cur
