"""GPU: BASELINE configs 5 (L=512, six modality streams, 8 GCN layers) and 4 (per-GPU shard B=32, L=110) checked
against the CPU oracle -- not against another HIP kernel.

cfg5 is beyond the reference's trimodal wiring (model_mm.py:97-106), so the checker is the oracle's M-stream
composition of reference-pinned functions (oracle/mmdfn_oracle.py::forward_streams, create_big_adj, gcnii_stack).
Dialogues never interact (block-diagonal adjacency incl. its normalisation, model_mm.py:137-178), so at full size the
oracle runs on ONE dialogue of the batch (dense (M L)^2 = 3072^2) and is compared with that dialogue's slice of the
device results; the size-independent properties cover the whole batch.
"""
import numpy as np
import pytest
import torch

import mmdfn_oracle as O
from mm_dfn_amd import synthetic, ops
from mm_dfn_amd.layout import pair_list
from util import abs_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stream_model(cfg, seed, dropout=0.0):
    m = synthetic.build_stream_model(dropout=dropout, **cfg)
    sd = synthetic.seeded_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    return m.to(DEV), sd


def _run_streams(m, b):
    return m([s.to(DEV) for s in b["streams"]], b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"])[0]


# ------------------------------------------------------------------------------------------------------------------
# (i) M = 6 streams, 8 layers, small ragged dialogues: log-probs and EVERY live gradient against the oracle
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lengths,dims", [([13, 7, 20], [24, 512, 36, 100, 64, 48]), ([1, 33, 2, 5], [512] * 6)])
def test_six_stream_eight_layer_model_against_oracle(lengths, dims):
    cfg = dict(B=len(lengths), L=max(lengths), P=2, C=6, nlayers=8, D_streams=dims)
    m, sd = _stream_model(cfg, 1201)
    m.train()                                              # dropout p = 0
    b = synthetic.make_stream_batch(1202, lengths=lengths, **cfg)
    xs = [s.to(DEV).requires_grad_(True) for s in b["streams"]]
    logp = m(xs, b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"])[0]
    w = torch.from_numpy(np.random.RandomState(1203).randn(*logp.shape).astype(np.float32))
    (logp * w.to(DEV)).sum().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = [s.clone().requires_grad_(True) for s in b["streams"]]
    want = O.forward_streams(params, xo, b["lengths"], O.default_cfg(8))
    (want * w).sum().backward()
    assert logp.shape == (sum(lengths), 6)
    assert abs_err(logp, want) < 1e-4
    checked = 0
    for k, p in m.named_parameters():
        gr = params[k].grad
        if p.grad is None:
            assert gr is None or float(gr.abs().max()) == 0.0, k
            continue
        assert rel_err(p.grad, gr) < 1e-4, k
        checked += 1
    assert checked == 2 * len(dims) + 8 + 2 + 4 + 2         # projections, convs, fcs.0, LSTM gate, head
    for x, y in zip(xs, xo):
        assert rel_err(x.grad, y.grad) < 1e-4


# ------------------------------------------------------------------------------------------------------------------
# (ii) cfg5-size kernels (L = 512, M = 6) against the oracle on one dialogue slice + whole-batch properties
# ------------------------------------------------------------------------------------------------------------------
def _cfg5_feats(B, seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.randn(6, B * 512, 200).astype(np.float32))


@pytest.mark.parametrize("B,j", [(8, 5), (32, 17)])
def test_cfg5_adjacency_propagate_and_their_gradients_against_oracle_slice(B, j):
    L, M, d = 512, 6, 100
    lengths = [L] * B
    N = L * B
    sl = slice(j * L, (j + 1) * L)
    feats_cpu = _cfg5_feats(B, 1300 + B)
    feats = feats_cpu.to(DEV).requires_grad_(True)
    adj = ops.build_adjacency(feats, lengths)
    lay = adj.layout
    rs = np.random.RandomState(1301)
    H_cpu = torch.from_numpy(rs.randn(M * N, d).astype(np.float32))
    dO_cpu = torch.from_numpy(rs.randn(M * N, d).astype(np.float32))
    H = H_cpu.to(DEV).requires_grad_(True)

    # ---- oracle on dialogue j alone (dense 3072 x 3072)
    fj = [feats_cpu[m_, sl].clone().requires_grad_(True) for m_ in range(M)]
    t_o, c_o, _ = O.adjacency_tiles(fj, [L])
    dense = O.tiles_to_dense(t_o, c_o, [L], M)
    Hj = torch.cat([H_cpu[m_ * N + j * L: m_ * N + (j + 1) * L] for m_ in range(M)], 0).requires_grad_(True)
    dOj = torch.cat([dO_cpu[m_ * N + j * L: m_ * N + (j + 1) * L] for m_ in range(M)], 0)
    out_o = dense @ Hj

    # K5 forward: adjacency tiles and cross-modal diagonals of dialogue j
    base = int(lay.tile_base_host[j])
    t_d = adj.tiles[base: base + M * L * L]
    # off the diagonal to 2e-5 of the largest entry; ON the diagonal (self-similarity: acos at 0.99999, slope -224,
    # model_mm.py:149) one ulp of the cosine already moves the entry by 1.3e-5 relative, and it is the largest entry
    T_d, T_o = t_d.view(M, L, L).cpu(), t_o.detach().view(M, L, L)
    off = ~torch.eye(L, dtype=torch.bool).unsqueeze(0).expand(M, L, L)
    e = float((T_d - T_o)[off].abs().max() / T_o[off].abs().max())
    assert e < 2e-5, "adjacency tiles (off-diagonal) rel err %.3g" % e
    e = float((T_d - T_o)[~off].abs().max() / T_o[~off].abs().max())
    assert e < 2e-4, "adjacency tiles (diagonal) rel err %.3g" % e
    e = rel_err(adj.cross[:, sl], c_o.detach())
    assert e < 2e-5, "cross-modal diagonals rel err %.3g" % e

    # K6 forward through the autograd op (the bf16-piece kernel at this size)
    out = ops.propagate(adj, H)
    got_j = torch.cat([out[m_ * N + j * L: m_ * N + (j + 1) * L] for m_ in range(M)], 0)
    e = rel_err(got_j, out_o.detach())
    assert e < 1e-5, "propagate rel err %.3g" % e

    # K6 backward: dH = A^T dO, dA = dO H^T on the tile pattern, then K5 backward down to the features
    out.backward(dO_cpu.to(DEV))
    out_o.backward(dOj)
    dH_j = torch.cat([H.grad[m_ * N + j * L: m_ * N + (j + 1) * L] for m_ in range(M)], 0)
    e = rel_err(dH_j, Hj.grad)
    assert e < 1e-5, "dH rel err %.3g" % e
    want_df = torch.stack([f.grad for f in fj], 0)
    e = rel_err(feats.grad[:, sl], want_df)
    assert e < 1e-4, "dfeats (through K6', K5 backward) rel err %.3g" % e

    # tile_outer alone (dtiles / dcross of dialogue j) against the dense outer product on the pattern
    dt, dc = ops.tile_outer_raw(dO_cpu.to(DEV), H_cpu.to(DEV), lay)
    G = dOj @ Hj.detach().t()                                  # (ML, ML): dense dL/dA of dialogue j
    want_t = torch.cat([G[m_ * L:(m_ + 1) * L, m_ * L:(m_ + 1) * L].reshape(-1) for m_ in range(M)])
    e = rel_err(dt[base: base + M * L * L], want_t)
    assert e < 1e-5, "tile_outer tiles rel err %.3g" % e
    ar = torch.arange(L)
    for k, (m_, n_) in enumerate(pair_list(M)):
        want_c = G[m_ * L + ar, n_ * L + ar] + G[n_ * L + ar, m_ * L + ar]
        e = rel_err(dc[k, sl], want_c)
        assert e < 1e-5, "tile_outer cross (%d,%d) rel err %.3g" % (m_, n_, e)


def test_cfg5_whole_batch_properties():
    """B = 32, L = 512, M = 6: linearity of K6 and the fixed point A_hat . deg^(1/2) = deg^(1/2) of the normalised
    adjacency (A_hat = R A R with R = deg^-1/2, so A_hat (1/r) = R A 1 = R deg = 1/r) on every row of the batch."""
    L, M, B = 512, 6, 32
    lengths = [L] * B
    N = L * B
    feats = _cfg5_feats(B, 1400).to(DEV)
    adj = ops.build_adjacency(feats, lengths)
    H1 = torch.randn(M * N, 100, device=DEV)
    H2 = torch.randn(M * N, 100, device=DEV)
    lhs = ops.propagate(adj, 2.0 * H1 - 0.5 * H2)
    rhs = 2.0 * ops.propagate(adj, H1) - 0.5 * ops.propagate(adj, H2)
    assert rel_err(lhs, rhs) < 1e-5
    # r from the tile diagonals: A_hat[p, p] = sim(|u_p|^2) r_p^2
    tiles = adj.tiles.view(B, M, L, L)
    r2 = tiles.diagonal(dim1=2, dim2=3).permute(1, 0, 2).reshape(M, N)
    unit = feats / feats.norm(dim=2, keepdim=True)
    spp = 1.0 - torch.acos((unit * unit).sum(2) * 0.99999) / np.pi
    inv_r = torch.sqrt(spp / r2).reshape(M * N, 1).repeat(1, 4).contiguous()
    assert rel_err(ops.propagate(adj, inv_r), inv_r) < 1e-4
    # symmetry of the stored tiles (the backward pass relies on it: dH = A_hat dO)
    assert abs_err(tiles, tiles.transpose(2, 3)) < 1e-7


# ------------------------------------------------------------------------------------------------------------------
# (iii) the cfg5 module stack at full dialogue length
# ------------------------------------------------------------------------------------------------------------------
def test_cfg5_stack_two_long_dialogues_all_gradients_against_oracle():
    """L = 512, M = 6, 8 layers, B = 2: log-probs and every live parameter gradient against the oracle (dense 6144^2).

    The model holds ~11 M ReLU pre-activations (projections + 8 layers x 6 144 x 200); where one lies within fp32 rounding of
    zero two correct fp32 summation orders disagree on its side and every gradient upstream of that unit moves (1e-2
    relative on one parameter; of the batch seeds 1502..1511 five have such a unit against this oracle's summation order,
    1502 -- used here, the first of them -- with |pre| = 8.5e-8 in stream 4).  Instead of picking a seed that happens to
    agree the test reads the device's own ReLU decisions, requires every disagreement to be a pre-activation
    below 1e-5 and differentiates the oracle on the linear piece the device is on (tests/util.relu_flips_from_tap)."""
    from mm_dfn_amd import gcn_stack
    from util import relu_flips_from_tap
    cfg = dict(synthetic.STREAM_CONFIGS["cfg5"], B=2)
    m, sd = _stream_model(cfg, 1501)
    m.train()
    b = synthetic.make_stream_batch(1502, **cfg)
    gcn_stack.TAP = []
    try:
        logp = _run_streams(m, b)
    finally:
        tap, gcn_stack.TAP = gcn_stack.TAP, None
    w = torch.from_numpy(np.random.RandomState(1502).randn(*logp.shape).astype(np.float32))
    (logp * w.to(DEV)).sum().backward()

    def oracle(flips=None):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        probe = O.ReluProbe(flips)
        prev = O.set_relu_probe(probe)
        try:
            want = O.forward_streams(params, b["streams"], b["lengths"], O.default_cfg(8))
            (want * w).sum().backward()
        finally:
            O.set_relu_probe(prev)
        return want.detach(), {k: v.grad for k, v in params.items()}, probe

    want, grads, probe = oracle()
    assert abs_err(logp, want) < 1e-4
    named = [(k, p) for k, p in m.named_parameters() if p.grad is not None]
    if not all(rel_err(p.grad, grads[k]) < 1e-4 for k, p in named):
        assert len(tap) == 1
        flips = relu_flips_from_tap(tap[0], probe, "graph_model.graph_net.", 6, sum(b["lengths"]))
        assert flips, "gradients differ although device and oracle agree on every ReLU"
        print("ReLU units evaluated on the device's side of the kink: %s"
              % {k: [(int(r), int(c), float(probe.pre[k][r, c])) for r, c in v] for k, v in flips.items()})
        _, grads, _ = oracle(flips)
    assert len(named) >= 20
    for k, p in named:
        assert rel_err(p.grad, grads[k]) < 1e-4, k


def test_cfg5_stack_full_batch_eval_against_oracle_slice():
    """The BASELINE cfg5 batch (B = 8 dialogues of 512 utterances): eval log-probs of one dialogue vs the oracle run on
    that dialogue alone, and the input gradient of that dialogue (train mode, dropout 0)."""
    cfg = dict(synthetic.STREAM_CONFIGS["cfg5"])
    j, L = 3, cfg["L"]
    m, sd = _stream_model(cfg, 1601)
    b = synthetic.make_stream_batch(1602, **cfg)
    m.eval()
    with torch.no_grad():
        logp = _run_streams(m, b)
        want = O.forward_streams(sd, [s[:, j:j + 1] for s in b["streams"]], [L], O.default_cfg(8))
    assert abs_err(logp[j * L:(j + 1) * L], want) < 1e-4
    m.train()
    xs = [s.to(DEV).requires_grad_(True) for s in b["streams"]]
    logp = m(xs, b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"])[0]
    w = torch.from_numpy(np.random.RandomState(1603).randn(*logp.shape).astype(np.float32))
    (logp * w.to(DEV)).sum().backward()
    xo = [s[:, j:j + 1].clone().requires_grad_(True) for s in b["streams"]]
    wo = O.forward_streams(sd, xo, [L], O.default_cfg(8))
    (wo * w[j * L:(j + 1) * L]).sum().backward()
    for x, y in zip(xs, xo):
        assert rel_err(x.grad[:, j:j + 1], y.grad) < 1e-4


# ------------------------------------------------------------------------------------------------------------------
# (iv) cfg4 per-GPU shard (B = 32, L = 110) through the trimodal model
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ragged", [False, True])
def test_cfg4_shard_eval_logits_against_oracle(ragged):
    cfg = dict(synthetic.CONFIGS["cfg4"])
    b = synthetic.make_batch(1701, ragged=ragged, **cfg)
    m = synthetic.build_model(**cfg)
    sd = synthetic.seeded_state_dict(m.state_dict(), 1700)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    with torch.no_grad():
        got = m(b["textf"].to(DEV), b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"], b["acouf"].to(DEV),
                b["visuf"].to(DEV))[0]
        want = O.forward(sd, b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"],
                         O.default_cfg(cfg["nlayers"]), engine="aten")
    assert got.shape == want.shape
    assert abs_err(got, want) < 1e-4
