"""Weight gradients: dW = dY^T X as hand-written kernels, and the per-step QUEUE that issues all of them as one launch pair.

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import ctypes
import torch

from . import _hip
from .ops_pad import pad4, padded_grad_like, row_operand, row_padded_view


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
# Riders (MMDFN_WGRAD_RIDERS=0 turns them off): what is queued when a GRU backward recurrence is about to be launched -- the
# graph side's weight gradients in front of the second GRU layer's recurrence, that layer's own in front of the first layer's --
# does not wait for the end of the pass: its tiles run as extra workgroups of the recurrence launch, on the CUs the recurrence
# leaves idle (csrc/gru.hip gru_seq_bwd_riders_kernel, include/mmdfn_hip.h mmdfn_wgrad_riders_stage).
RIDERS = __import__("os").environ.get("MMDFN_WGRAD_RIDERS", "1") == "1"
_RIDER_MAXSEG = 16  # MMDFN_RIDER_MAXSEG of csrc/mmdfn_internal.h
RIDER_LOG = None    # a list: stage_riders appends what it found queued and what it took (tools/rider_log.py)


class wgrad_batch:
    """``with ops.wgrad_batch(): loss.backward()`` -- weight gradients of leaf parameters are batched into one launch pair
    and written to ``.grad`` directly (see the comment above).  Re-entrant; exception-safe."""

    def __enter__(self):
        if _WGQ["scope"] == 0 and (_WGQ["outs"] or _WGQ["ext"] or _WGQ["armed"]):
            _WGQ["outs"], _WGQ["ext"], _WGQ["armed"] = {}, [], False        # stale entries of a backward that raised
        if _WGQ["scope"] == 0:
            drop_grad_addends()                                             # (same: its callback never ran)
            if _WGQ.get("rider_keep") or _WGQ.get("rider_held"):            # (a backward that raised with riders under way)
                _drain_riders(discard=True)
        _WGQ["scope"] += 1
        return self

    def __exit__(self, exc_type, exc, tb):
        _WGQ["scope"] -= 1
        if _WGQ["scope"] == 0:
            if exc_type is None:
                if _WGQ["outs"] or _WGQ["ext"]:
                    flush_queued_wgrads()                  # a backward driven without the engine callback
                _drain_riders()
            else:
                _WGQ["outs"], _WGQ["ext"], _WGQ["armed"] = {}, [], False    # the callback never ran: drop the half-built batch
                drop_grad_addends()
                _drain_riders(discard=True)
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


# what a CU delivers on the bf16-piece weight-gradient tiles (the cfg2 batch: 10.4 GFLOP in 96 us on 256 CUs) and what a
# recurrence step takes: the rider batch is sized so that its tiles end with the recurrence (tiles that outlast it run on a
# chip whose other CUs have nothing left to do: cfg2's first-layer launch took 100 us instead of 84 with everything aboard)
_RIDER_FLOPS_PER_CU_S = 4.2e11
RIDER_BUDGET = float(__import__("os").environ.get("MMDFN_RIDER_BUDGET", "0.8"))    # (0.65 .. 0.9 measure the same, 1.0 .. 1.5 +0.4 %)


def stage_riders(rows, T):
    """In front of the plain GRU backward recurrence launch of groups with ``rows`` sequences and ``T`` steps: weight gradients
    queued so far are staged as riders of that launch -- if it is of the kind that takes riders, as many (in queue order, at
    most 16 segments) as the CUs it leaves idle can finish while it runs; the rest stays queued for the next launch / the
    end-of-backward flush.  ``finish_riders()`` must follow the launch."""
    if not (RIDERS and _WGQ["scope"] > 0 and _WGQ["outs"]) or _WGQ.get("rider_keep"):
        return
    idle = _hip.lib().mmdfn_gru_seq_bwd_idle_cus(len(rows), _hip.int_array(rows))
    if idle <= 0:
        return
    step_s = 1e-9 * _hip.lib().mmdfn_gru_seq_bwd_step_ns(len(rows), _hip.int_array(rows))     # (0.75 us; the MFMA form 2.3)
    budget = RIDER_BUDGET * idle * max(T) * step_s * _RIDER_FLOPS_PER_CU_S
    take, nseg, work = [], 0, 0.0
    for key, o in list(_WGQ["outs"].items()):
        f = sum(2.0 * a.shape[0] * o["M"] * o["N"] for a, _, _ in o["segs"])
        if nseg + len(o["segs"]) > _RIDER_MAXSEG or work + f > 1.1 * budget:
            continue
        take.append(key)
        nseg += len(o["segs"])
        work += f
    if RIDER_LOG is not None:
        RIDER_LOG.append(dict(idle=idle, T=max(T), budget_gflop=budget / 1e9, taken_gflop=work / 1e9, taken_segments=nseg,
                              queued=[(o["M"], o["N"], len(o["segs"]), sum(a.shape[0] for a, _, _ in o["segs"]), k in take)
                                      for k, o in _WGQ["outs"].items()]))
    if not take:
        return
    outs = [_WGQ["outs"].pop(k) for k in take]      # 'armed' stays set: the end-of-backward callback flushes the rest
    _flush_outs(outs, None, (), stage=True)


def finish_riders():
    """Behind the GRU backward launch: a staged batch the launch did not take (another kernel form) is issued now; the
    references that kept its operands and slabs alive are dropped (the launches are in the stream)."""
    if _WGQ.get("rider_keep"):
        rc = _hip.lib().mmdfn_wgrad_riders_flush(_hip.stream())
        # the slabs are read again by the reduction launch that takes them in (the end-of-backward batch's): held until then
        _WGQ.setdefault("rider_held", []).append(_WGQ["rider_keep"])
        _WGQ["rider_keep"] = None
        _hip.check(rc, "mmdfn_wgrad_riders_flush")


def _drain_riders(discard=False):
    """End of the backward pass: the rider batches' slab stacks that no reduction launch took in are reduced now."""
    if _WGQ.get("rider_held") or _WGQ.get("rider_keep"):
        if _WGQ.get("rider_keep") and not discard:
            finish_riders()
        rc = _hip.lib().mmdfn_wgrad_riders_drain(_hip.stream(), 1 if discard else 0)
        _WGQ["rider_held"], _WGQ["rider_keep"] = [], None
        _hip.check(rc, "mmdfn_wgrad_riders_drain")


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
    _drain_riders()


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


def _flush_outs(outs, side, ext=(), stage=False):
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
    prepared = [_prepare_wgrad_batch(b, ext_items if (ride and b is batches[-1]) else None, stage=stage and b is batches[-1])
                for b in batches]          # allocations (workspace) on the current stream
    if stage:
        _WGQ["rider_keep"] = (outs, prepared)
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


def _prepare_wgrad_batch(batch, ext_items=None, stage=False):
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
        fn = lib.mmdfn_wgrad_riders_stage if stage else lib.mmdfn_gemm_tn_batch
        rc = fn(len(A), pa(A), pa(B), ia(R), ia(lda), ia(ldb), ia(sh), ia(oi), len(C), pa(C), pa(cs1),
                pa(cs2), ia(M), ia(N), ia(ldc), ia(acc), _hip.ptr(ws), stream)
        _hip.check(rc, "mmdfn_wgrad_riders_stage" if stage else "mmdfn_gemm_tn_batch")
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
