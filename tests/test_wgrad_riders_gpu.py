"""Weight-gradient riders of the GRU backward recurrence launch (ops_wgrad.stage_riders, csrc/gru.hip
gru_seq_bwd_riders_kernel, include/mmdfn_hip.h mmdfn_wgrad_riders_stage): the tiles of the step's weight-gradient queue that do not
depend on a recurrence run as extra workgroups of its launch.  The recurrence's own results must not change by a bit, every
parameter gradient must be the one the end-of-backward batch produces (to fp32 summation order: the split of a batch into slabs
depends on what else is in the batch), and the rider launch must actually be the one that runs."""
import pytest
import torch

from mm_dfn_amd import ops, ops_wgrad, synthetic
from mm_dfn_amd import train as T
from mm_dfn_amd.loss import FocalLoss

pytestmark = pytest.mark.gpu
CFG = dict(P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)


def _step(m, b, flat, riders):
    prev = ops_wgrad.RIDERS
    ops_wgrad.RIDERS = riders
    try:
        m.zero_grad(set_to_none=True)
        for k in ("textf", "acouf", "visuf"):
            b[k].grad = None
        logp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
        T.backward(FocalLoss(gamma=0.5)(logp, flat))
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, \
               {k: b[k].grad.clone() for k in ("textf", "acouf", "visuf")}
    finally:
        ops_wgrad.RIDERS = prev


CFG3 = dict(P=9, C=7, nlayers=4, D_t=600, D_a=300, D_v=342)            # BASELINE cfg3 (MELD-like): the MFMA-form recurrence
RAGGED_CFG3 = (33, 3, 17, 31, 9, 27, 12, 33, 5, 21, 30, 8, 16, 2, 25, 11, 33, 7, 19, 29, 4, 14, 23, 10, 32, 6, 18, 28, 13, 22, 1, 26)


@pytest.mark.parametrize("lengths,CFG", [((20, 13, 7), CFG), (tuple([110] * 16), CFG), ((110, 64, 27, 110, 90, 33), CFG),
                                         (RAGGED_CFG3, CFG3)], ids=["small", "cfg2", "ragged", "cfg3"])
def test_riders_leave_every_gradient_as_the_batch_computes_it(lengths, CFG):
    from torch.profiler import ProfilerActivity, profile
    m = synthetic.build_model(dropout=0.0, **CFG)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 7))
    m = m.cuda().train()
    b = synthetic.make_batch(13, lengths=list(lengths), device="cuda", B=len(lengths), L=max(lengths), **CFG)
    for k in ("textf", "acouf", "visuf"):
        b[k] = b[k].detach().requires_grad_(True)
    flat = T.flatten_labels(b["label"], b["lengths"])
    want, wantx = _step(m, b, flat, False)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        got, gotx = _step(m, b, flat, True)
    names = [e.key for e in prof.key_averages()]
    if CFG is CFG3:
        assert any("gru_seq_bwd_mfma_riders_kernel" in n for n in names), names
    elif len(set(lengths)) == 1 or max(lengths) < 32:    # (ragged long dialogues may take the valid-length launches: no riders)
        assert any("gru_seq_bwd_riders_kernel" in n for n in names), names
    assert set(got) == set(want)
    for k in want:
        scale = float(want[k].abs().max()) + 1e-30
        assert float((got[k] - want[k]).abs().max()) <= 2e-5 * scale, k
    # the recurrence itself (and everything downstream of it) is bit-identical: the input gradients pass through both GRU layers
    for k in wantx:
        assert torch.equal(gotx[k], wantx[k]), k
    # a second pass with riders reproduces itself bit for bit
    again, _ = _step(m, b, flat, True)
    for k in got:
        assert torch.equal(again[k], got[k]), k


def _batch_call(A, Bm, C, cs, stage):
    o = dict(M=A.shape[1], N=Bm.shape[1])
    return ops_wgrad._prepare_wgrad_batch([(o, C, [cs], 0, [(A, Bm, 0)])], stage=stage)


def _gru_bwd(T_, rows, seed):
    from mm_dfn_amd import _hip
    H = 100
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    dy, y, gates = r(T_, rows, 2 * H), torch.tanh(r(T_, rows, 2 * H)), torch.sigmoid(r(T_, rows, 2, 4, H))
    whh = [0.1 * r(3 * H, H), 0.1 * r(3 * H, H)]
    dgi = torch.full((T_, rows, 6 * H), float("nan"), device="cuda")
    dgh = torch.full_like(dgi, float("nan"))
    rc = _hip.lib().mmdfn_gru_seq_bwd(1, _hip.ptr_array([dy]), _hip.ptr_array([y]), _hip.ptr_array([gates]), _hip.ptr_array(whh),
                                      _hip.ptr_array([dgi]), _hip.ptr_array([dgh]), _hip.int_array([rows]), _hip.int_array([T_]),
                                      H, _hip.stream())
    _hip.check(rc, "mmdfn_gru_seq_bwd")
    return dgi, dgh


@pytest.mark.parametrize("rows,T_", [(80, 110), (3, 17), (127, 40), (600, 20)])     # (600: the MFMA form)
def test_rider_launch_is_bit_equal_to_the_two_plain_launches(rows, T_):
    """C-ABI level: (stage a batch, GRU backward) == (GRU backward, mmdfn_gemm_tn_batch) bit for bit, for the recurrence's
    outputs and for the batch's; and a staged batch that no GRU launch takes is flushed by mmdfn_wgrad_riders_flush."""
    from mm_dfn_amd import _hip
    lib = _hip.lib()
    torch.manual_seed(3)
    A = torch.randn(4100, 300, device="cuda")
    Bm = torch.randn(4100, 200, device="cuda")
    C0, c0 = torch.empty(300, 200, device="cuda"), torch.empty(300, device="cuda")
    _batch_call(A, Bm, C0, c0, False)(_hip.stream())
    dgi0, dgh0 = _gru_bwd(T_, rows, 5)
    C1, c1 = torch.empty(300, 200, device="cuda"), torch.empty(300, device="cuda")
    call = _batch_call(A, Bm, C1, c1, True)
    call(_hip.stream())
    assert lib.mmdfn_wgrad_riders_staged() == 1
    dgi1, dgh1 = _gru_bwd(T_, rows, 5)
    assert lib.mmdfn_wgrad_riders_staged() == 0          # the launch took it
    _hip.check(lib.mmdfn_wgrad_riders_drain(_hip.stream(), 0), "mmdfn_wgrad_riders_drain")      # (its slab reduction)
    torch.cuda.synchronize()
    assert torch.equal(dgi0, dgi1) and torch.equal(dgh0, dgh1)
    assert torch.equal(C0, C1) and torch.equal(c0, c1)
    ref = A.double().t() @ Bm.double()
    assert float((C1.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # staged, no GRU launch: the flush launches it
    C2, c2 = torch.empty(300, 200, device="cuda"), torch.empty(300, device="cuda")
    call2 = _batch_call(A, Bm, C2, c2, True)
    call2(_hip.stream())
    assert lib.mmdfn_wgrad_riders_staged() == 1
    _hip.check(lib.mmdfn_wgrad_riders_flush(_hip.stream()), "mmdfn_wgrad_riders_flush")
    assert lib.mmdfn_wgrad_riders_staged() == 0
    _hip.check(lib.mmdfn_wgrad_riders_drain(_hip.stream(), 0), "mmdfn_wgrad_riders_drain")
    torch.cuda.synchronize()
    assert torch.equal(C0, C2) and torch.equal(c0, c2)
    # a later batch that writes the same gradient: the waiting reduction goes first, the later batch accumulates onto it
    C3, c3 = torch.empty(300, 200, device="cuda"), torch.empty(300, device="cuda")
    _batch_call(A, Bm, C3, c3, True)(_hip.stream())
    _gru_bwd(T_, rows, 5)
    o = dict(M=300, N=200)
    ops_wgrad._prepare_wgrad_batch([(o, C3, [c3], 1, [(A, Bm, 0)])])(_hip.stream())
    _hip.check(lib.mmdfn_wgrad_riders_drain(_hip.stream(), 0), "mmdfn_wgrad_riders_drain")
    torch.cuda.synchronize()
    assert float((C3 - 2 * C0).abs().max()) <= 1e-5 * float(C0.abs().max())
    assert float((c3 - 2 * c0).abs().max()) <= 1e-5 * float(c0.abs().max())
    # two rider batches in a row that write the same gradient: the first one's waiting reduction goes first as well
    C4, c4 = torch.empty(300, 200, device="cuda"), torch.empty(300, device="cuda")
    _batch_call(A, Bm, C4, c4, True)(_hip.stream())
    _gru_bwd(T_, rows, 5)
    ops_wgrad._prepare_wgrad_batch([(o, C4, [c4], 1, [(A, Bm, 0)])], stage=True)(_hip.stream())
    _gru_bwd(T_, rows, 5)
    assert lib.mmdfn_wgrad_riders_staged() == 0
    _hip.check(lib.mmdfn_wgrad_riders_drain(_hip.stream(), 0), "mmdfn_wgrad_riders_drain")
    torch.cuda.synchronize()
    assert float((C4 - 2 * C0).abs().max()) <= 1e-5 * float(C0.abs().max())
    assert float((c4 - 2 * c0).abs().max()) <= 1e-5 * float(c0.abs().max())


@pytest.mark.parametrize("lengths,CFG", [((20, 13, 7), CFG), (tuple([110] * 16), CFG), (RAGGED_CFG3, CFG3)], ids=["small", "cfg2", "cfg3"])
def test_dropout_flags_drawn_as_riders_of_the_gru_forward_launch_are_the_same_flags(lengths, CFG):
    """The step's keep flags drawn by rider workgroups of the first GRU layer's forward launch (ops_flags.stage_flag_draw,
    gru_seq_fwd_io_flags_kernel) are the flags the generator launch of its own draws -- same seed, same step: bit-identical
    log-probabilities and gradients --, the generator launch is gone from the trace, and a captured step keeps drawing fresh
    flags at every replay."""
    from torch.profiler import ProfilerActivity, profile
    from mm_dfn_amd import ops_flags
    from mm_dfn_amd.graphs import CapturedStep
    m = synthetic.build_model(dropout=0.5, **CFG)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 9))
    m = m.cuda().train()
    b = synthetic.make_batch(17, lengths=list(lengths), device="cuda", B=len(lengths), L=max(lengths), **CFG)
    flat = T.flatten_labels(b["label"], b["lengths"])
    loss_f = FocalLoss(gamma=0.5)

    def step():
        m.zero_grad(set_to_none=True)
        logp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
        T.backward(loss_f(logp, flat))
        return logp.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    step()                                          # (every later step finds the hint: one draw per step)

    def two_steps(rider):
        prev, ops_flags.FLAG_RIDER = ops_flags.FLAG_RIDER, rider
        try:
            torch.manual_seed(1234)
            step()                                  # (leaves the hint: how many flags a step uses)
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                out = step()
                torch.cuda.synchronize()
            return out, [e.key for e in prof.key_averages()]
        finally:
            ops_flags.FLAG_RIDER = prev
    (lp0, g0), names0 = two_steps(False)
    (lp1, g1), names1 = two_steps(True)
    carrier = "gru_seq_fwd_mfma_flags_kernel" if CFG is CFG3 else "gru_seq_fwd_io_flags_kernel"
    assert any("keep_flags_kernel" in n for n in names0) and not any("_flags_kernel" in n and "keep" not in n for n in names0), names0
    assert any(carrier in n for n in names1) and not any("keep_flags_kernel" in n for n in names1), names1
    assert torch.equal(lp0, lp1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # captured: replays differ from each other (fresh flags), and the flags are Bernoulli(0.5)
    out = {}

    def fn():
        out["logp"] = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
        loss = loss_f(out["logp"], flat)
        T.backward(loss)
        return loss
    cap = CapturedStep(m, fn, warmup=2)
    cap.replay(); a = out["logp"].clone()
    cap.replay(); c = out["logp"].clone()
    torch.cuda.synchronize()
    assert not torch.equal(a, c)
    cap.close()
