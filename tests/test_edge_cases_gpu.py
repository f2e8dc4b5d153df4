"""GPU parity on edge cases of the path: degenerate dialogue lengths, silent speakers, single-dialogue batches,
two-modality graphs, the optional speaker / modality embeddings, non-default modal_weight."""
import numpy as np
import pytest
import torch

import mmdfn_oracle as O
from mm_dfn_amd import MM_GCN, synthetic, ops
from util import abs_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run_model(cfg, lengths, seed, qmask_edit=None):
    m = synthetic.build_model(**cfg)
    sd = synthetic.seeded_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    b = synthetic.make_batch(seed + 1, lengths=lengths, **cfg)
    if qmask_edit is not None:
        qmask_edit(b["qmask"])
    logp = m(b["textf"].to(DEV), b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"], b["acouf"].to(DEV),
             b["visuf"].to(DEV))[0]
    w = torch.from_numpy(np.random.RandomState(seed).randn(*logp.shape).astype(np.float32))
    (logp * w.to(DEV)).sum().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want = O.forward(params, b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"],
                     O.default_cfg(cfg["nlayers"]), engine="aten")
    (want * w).sum().backward()
    return m, logp, params, want


@pytest.mark.parametrize("lengths", [[1], [1, 1, 2], [2, 1], [3], [17, 1, 9]])
def test_degenerate_dialogue_lengths(lengths):
    cfg = dict(B=len(lengths), L=max(lengths), P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    m, logp, params, want = _run_model(cfg, lengths, 31)
    assert abs_err(logp, want) < 1e-4
    for k in ("linear_l.weight", "rnn_parties.weight_hh_l0", "graph_model.graph_net.convs.1.weight", "smax_fc.weight"):
        g = dict(m.named_parameters())[k].grad
        assert rel_err(g, params[k].grad) < 1e-4, k


def test_silent_speaker_and_unflagged_utterance():
    """Speaker 1 never speaks in dialogue 0; one valid utterance carries no speaker flag at all (its party term is 0)."""
    cfg = dict(B=2, L=9, P=3, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)

    def edit(q):
        q[:, 0, 1] = 0.0
        q[:, 0, 0] = torch.maximum(q[:, 0, 0], (q[:, 0].sum(1) == 0).float() * (torch.arange(9) < 9).float())
        q[4, 1, :] = 0.0
    m, logp, params, want = _run_model(cfg, [9, 6], 32, edit)
    assert abs_err(logp, want) < 1e-4
    assert rel_err(m.rnn_parties.weight_ih_l0.grad, params["rnn_parties.weight_ih_l0"].grad) < 1e-4


@pytest.mark.parametrize("modals", ["av", "al", "vl"])
@pytest.mark.parametrize("modal_weight", [1.0, 0.6])
def test_two_modality_graph_module(modals, modal_weight):
    rs = np.random.RandomState(40)
    lengths = [11, 4, 7]
    N = sum(lengths)
    net = MM_GCN(200, 200, 200, 200, 3, 100, 6, 0.0, 0.5, 0.2, True, True, True, n_speakers=2, modals=list(modals),
                 use_speaker=False, use_modal=False, reason_flag=True, modal_weight=modal_weight)
    sd = synthetic.seeded_state_dict(net.state_dict(), 41)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    feats = {k: torch.from_numpy(rs.randn(N, 200).astype(np.float32)) for k in "avl"}
    q = torch.zeros(max(lengths), len(lengths), 2)
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in feats.items()}
    out = net(dev["a"] if "a" in modals else [], dev["v"] if "v" in modals else [], dev["l"] if "l" in modals else [],
              lengths, q.to(DEV))
    w = torch.from_numpy(rs.randn(*out.shape).astype(np.float32))
    (out * w.to(DEV)).sum().backward()
    cpu = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    params = {"graph_model." + k: v for k, v in sd.items()}
    cfg = O.default_cfg(3, modal_weight=modal_weight)
    want = O.mm_gcn([cpu[k] for k in modals], lengths, params, cfg)
    (want * w).sum().backward()
    assert abs_err(out, want) < 5e-5
    for k in modals:
        assert rel_err(dev[k].grad, cpu[k].grad) < 1e-4, k


def test_speaker_and_modality_embeddings_enabled():
    """use_speaker / use_modal (model_mm.py:78-93): l += speaker_embeddings[argmax qmask], x_m += modal_embeddings[m]."""
    rs = np.random.RandomState(50)
    lengths = [8, 5]
    N = sum(lengths)
    L, B, P = 8, 2, 2
    net = MM_GCN(200, 200, 200, 200, 2, 100, 6, 0.0, 0.5, 0.2, True, True, True, n_speakers=P, modals=list("avl"),
                 use_speaker=True, use_modal=True, reason_flag=True)
    sd = synthetic.seeded_state_dict(net.state_dict(), 51)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    a, v, l = (torch.from_numpy(rs.randn(N, 200).astype(np.float32)) for _ in range(3))
    spk = rs.randint(0, P, size=(L, B))
    q = torch.zeros(L, B, P)
    for b_, n in enumerate(lengths):
        q[np.arange(n), b_, spk[:n, b_]] = 1
    with torch.no_grad():
        got = net(a.clone().to(DEV), v.clone().to(DEV), l.clone().to(DEV), lengths, q.to(DEV))
        flat_spk = torch.cat([torch.from_numpy(spk[:n, b_]) for b_, n in enumerate(lengths)])
        l2 = l + sd["speaker_embeddings.weight"][flat_spk]
        emb = sd["modal_embeddings.weight"]
        feats = [a + emb[0], v + emb[1], l2 + emb[2]]
        want = O.mm_gcn(feats, lengths, {"graph_model." + k: t for k, t in sd.items()}, O.default_cfg(2))
    assert abs_err(got, want) < 5e-5


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_ragged_batches_against_oracle(seed):
    """Random batch shapes (B, lengths, speakers, depth, input dims): train-mode (dropout 0) log-probs and the
    gradients of every live parameter against the CPU oracle."""
    rs = np.random.RandomState(900 + seed)
    B = int(rs.randint(1, 7))
    lengths = [int(x) for x in rs.randint(1, 41, size=B)]
    cfg = dict(B=B, L=max(lengths), P=int(rs.randint(2, 10)), C=int(rs.choice([6, 7])), nlayers=int(rs.randint(1, 5)),
               D_t=4 * int(rs.randint(5, 160)), D_a=4 * int(rs.randint(5, 100)), D_v=4 * int(rs.randint(5, 140)))
    m, logp, params, want = _run_model(cfg, lengths, 950 + seed)
    assert abs_err(logp, want) < 1e-4
    checked = 0
    unstable = []
    ref64 = {}

    def grads_fp64():
        """The same oracle in float64: second opinion when a pre-activation sits within fp32 rounding of a ReLU kink
        and two fp32 evaluations land on different sides.  (Seen once in 46 random cases, seed 45 of
        tools/campaign.sh: the gradient of convs.0.weight moved by 2.4 % between the device / the build container's
        fp32 oracle on one side and the GPU box's fp32 oracle / fp64 on the other, log-probs equal to 5e-7; every other
        weight draw on the same shapes agrees to 3e-6.  Round 3: with the modality projections on the few-row grouped
        kernel -- another fp32 summation order -- the device lands on the other side of that kink while both oracles
        agree with each other, so the campaign reports seed 45 as a failure of this check; with ops.GROUP_ROWS = 0
        (library projections) the same model matches every gradient to 1e-4.  Log-probs agree to 6e-7 either way.  The flipped case moves with every
        change of an fp32 summation order upstream: with the LDS-staged few-row kernel it was seed 14 (7 of 38 400 first-layer ReLU
        inputs below 2e-6 in fp64, the smallest 1.1e-8; convs.0.weight moved by 0.3 %), and with the k-permuted MFMA fragments of the
        final round-3 stack kernels none of the seeds 6..45 flips (tools/campaign.sh: 40 seeds, 0 failures).)"""
        if not ref64:
            sd = synthetic.seeded_state_dict(m.state_dict(), 950 + seed)
            b = synthetic.make_batch(950 + seed + 1, lengths=lengths, **cfg)
            p64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
            out = O.forward(p64, b["textf"].double(), b["qmask"].double(), b["umask"].double(), b["lengths"],
                            b["acouf"].double(), b["visuf"].double(), O.default_cfg(cfg["nlayers"]), engine="manual")
            w = torch.from_numpy(np.random.RandomState(950 + seed).randn(*out.shape).astype(np.float32)).double()
            (out * w).sum().backward()
            ref64.update({k: v.grad for k, v in p64.items()})
        return ref64

    for k, p in m.named_parameters():
        if p.grad is None:
            assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k
            continue
        g_ref = params[k].grad
        assert g_ref is not None, k
        if float(g_ref.abs().max()) < 1e-6:
            assert float(p.grad.abs().max()) < 1e-4, k
        elif rel_err(p.grad, g_ref) >= 1e-4:                      # gradients to 1e-4 relative
            # second opinion in float64 at the SAME tolerance; only when the fp32 and fp64 oracles themselves disagree
            # (a pre-activation within rounding of a ReLU kink: the gradient is discontinuous there) is the case
            # recorded as unstable instead of failed
            g64 = grads_fp64()[k]
            if rel_err(p.grad.double().cpu(), g64) >= 1e-4:
                assert rel_err(g_ref.double(), g64) >= 1e-4, "%s: device gradient off by %.3g" % (k, rel_err(p.grad, g_ref))
                unstable.append(k)
        checked += 1
    assert checked >= 40
    assert len(unstable) <= 2, unstable


@pytest.mark.parametrize("weights", ["1-2-0.5", "0-0-0", "0-4-0"])
def test_other_speaker_weight_settings(weights):
    """speaker_weights other than the scripts' '3-0-1': every modality active, none (the party encoder is skipped
    altogether), only the visual one -- against the oracle, forward and the encoder gradients."""
    cfg = dict(B=3, L=12, P=3, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    lengths = [12, 5, 8]
    m = synthetic.build_model(speaker_weights=weights, **cfg)
    sd = synthetic.seeded_state_dict(m.state_dict(), 61)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    b = synthetic.make_batch(62, lengths=lengths, **cfg)
    logp = m(b["textf"].to(DEV), b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"], b["acouf"].to(DEV),
             b["visuf"].to(DEV))[0]
    R = torch.from_numpy(np.random.RandomState(63).randn(*logp.shape).astype(np.float32))
    (logp * R.to(DEV)).sum().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = O.default_cfg(2, speaker_weights=[float(x) for x in weights.split("-")])
    want = O.forward(params, b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"], ocfg, engine="aten")
    (want * R).sum().backward()
    assert abs_err(logp, want) < 1e-4
    named = dict(m.named_parameters())
    for k in ("linear_a.weight", "linear_v.weight", "linear_l.weight", "lstm_l.weight_ih_l0", "rnn_parties.weight_hh_l1"):
        g, gr = named[k].grad, params[k].grad
        if gr is None or float(gr.abs().max()) == 0.0:
            assert g is None or float(g.abs().max()) == 0.0, k
        else:
            assert rel_err(g, gr) < 1e-4, k
