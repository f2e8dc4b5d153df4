"""GPU: the drop-in modules (HIP path) against the golden vectors of the real reference
and against the CPU oracle.  Logits within 1e-4 abs (north_star), gradients 1e-4..2e-4 rel."""
import os

import numpy as np
import pytest
import torch

import mmdfn_oracle as O
from mm_dfn_amd import FocalLoss, GCNII_lyc, synthetic, train, ops
from test_oracle_golden import E2E, GOLD, _digest, load
from util import abs_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def hip_model(cfg, seed, dropout=0.0):
    m = synthetic.build_model(dropout=dropout, **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), seed))
    return m.to(DEV)


def run(m, b):
    return m(b["textf"].to(DEV), b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"], b["acouf"].to(DEV),
             b["visuf"].to(DEV))[0]


@pytest.mark.parametrize("name", sorted(E2E))
def test_end_to_end_against_reference_golden(name):
    cfg, seed, lengths = E2E[name]
    g = load("e2e_%s.npz" % name)
    b = synthetic.make_batch(seed + 1, lengths=lengths, **cfg)
    m = hip_model(cfg, seed).eval()
    with torch.no_grad():
        logp = run(m, b)
    assert np.abs(logp.cpu().numpy() - g["log_prob"]).max() < 1e-4
    m.train()  # dropout p = 0
    logp = run(m, b)
    label = train.flatten_labels(b["label"].to(DEV), b["lengths"])
    loss = FocalLoss(gamma=0.5)(logp, label)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    live = [str(x) for x in g["live_params"]]
    for k in live:
        want = g["gd/" + k]
        assert grads[k] is not None, k
        got = _digest(grads[k])
        # all three components of the reference's digest: sum |g| and sum g^2 relative to themselves, the signed sum (which may
        # cancel to nothing) relative to sum |g|
        assert abs(got[1] - want[1]) / (want[1] + 1e-12) < 2e-4, k
        assert abs(got[2] - want[2]) / (want[2] + 1e-30) < 4e-4, k
        assert abs(got[0] - want[0]) / (want[1] + 1e-12) < 2e-4, k
    for k, gr in grads.items():
        if k not in live:
            assert gr is None or float(gr.abs().max()) == 0.0, k
    for k in [x[2:] for x in g.files if x.startswith("g/")]:
        want = g["g/" + k]
        assert np.abs(grads[k].cpu().numpy() - want).max() / np.abs(want).max() < 2e-4, k


@pytest.mark.parametrize("table", ["l1plain", "l1seg", "table"])
@pytest.mark.parametrize("name", sorted(E2E))
def test_end_to_end_golden_with_valid_length_party_launches(name, table):
    """The same reference goldens with the party encoder on the valid-length launches (gru.TRUNCATE forced on: at these batch
    sizes "auto" keeps the full-length form): log-probabilities, loss and every gradient digest, through the queued
    weight-gradient batch (the all-padding sequence's segments are queued from the side stream)."""
    from mm_dfn_amd import gru as fused
    cfg, seed, lengths = E2E[name]
    g = load("e2e_%s.npz" % name)
    b = synthetic.make_batch(seed + 1, lengths=lengths, **cfg)
    prev, fused.TRUNCATE = fused.TRUNCATE, True
    prev_tab, fused.USE_TABLE = fused.USE_TABLE, table == "table"
    prev_l1, fused.L1_SKIPS_SILENT = fused.L1_SKIPS_SILENT, table != "l1plain"
    try:
        m = hip_model(cfg, seed).eval()
        with torch.no_grad():
            logp = run(m, b)
        assert np.abs(logp.cpu().numpy() - g["log_prob"]).max() < 1e-4
        m.train()
        logp = run(m, b)
        label = train.flatten_labels(b["label"].to(DEV), b["lengths"])
        loss = FocalLoss(gamma=0.5)(logp, label)
        assert abs(loss.item() - float(g["loss"])) < 1e-5
        train.backward(loss)
    finally:
        fused.TRUNCATE, fused.USE_TABLE, fused.L1_SKIPS_SILENT = prev, prev_tab, prev_l1
    grads = {k: p.grad for k, p in m.named_parameters()}
    for k in [str(x) for x in g["live_params"]]:
        want = g["gd/" + k]
        assert grads[k] is not None, k
        assert abs(_digest(grads[k])[1] - want[1]) / (want[1] + 1e-12) < 2e-4, k
    for k in [x[2:] for x in g.files if x.startswith("g/")]:
        want = g["g/" + k]
        assert np.abs(grads[k].cpu().numpy() - want).max() / np.abs(want).max() < 2e-4, k


def test_gcnii_module_against_golden():
    g = load("gcnii.npz")
    for ci in range(3):
        nl, reason = [int(x) for x in g["cfg%d" % ci]]
        rs = np.random.RandomState(300 + ci)
        lengths = [9, 4]
        N = sum(lengths)
        net = GCNII_lyc(nfeat=200, nlayers=nl, nhidden=100, nclass=6, dropout=0.0, lamda=0.5, alpha=0.2, variant=True,
                        return_feature=True, use_residue=True, reason_flag=bool(reason))
        net.load_state_dict(synthetic.seeded_state_dict(net.state_dict(), 300 + ci))
        net = net.to(DEV).train()
        x = torch.from_numpy(rs.randn(3 * N, 200).astype(np.float32)).to(DEV).requires_grad_(True)
        feats = torch.stack([torch.from_numpy(rs.randn(N, 200).astype(np.float32)) for _ in range(3)], 0).to(DEV)
        adj = ops.build_adjacency(feats, lengths)
        R = torch.from_numpy(rs.randn(3 * N, 300).astype(np.float32)).to(DEV)
        y = net(x, lengths, None, adj)
        assert np.abs(y.detach().cpu().numpy() - g["y%d" % ci]).max() < 2e-5
        (y * R).sum().backward()
        assert rel_err(x.grad, torch.from_numpy(g["dx%d" % ci])) < 1e-4
        assert rel_err(net.convs[0].weight.grad, torch.from_numpy(g["dW0_%d" % ci])) < 1e-4
        # a dense adjacency tensor is accepted too (drop-in signature)
        from torch.profiler import ProfilerActivity, profile
        dense = adj.to_dense()
        xd = x.detach().clone().requires_grad_(True)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            y2 = net(xd, lengths, None, dense)
            (y2 * R).sum().backward()
            torch.cuda.synchronize()
        assert abs_err(y2, y) < 2e-5
        assert rel_err(xd.grad, torch.from_numpy(g["dx%d" % ci])) < 1e-4
        # ... and it stays on this package's MFMA kernels (round 5: the op-by-op composition no longer calls torch.mm / F.linear)
        names = [e.key for e in prof.key_averages()]
        assert not [n for n in names if n.startswith("Cijk_")], names


def test_adjacency_against_golden():
    g = load("adjacency.npz")
    m = hip_model(E2E["deep16"][0], 1)
    for ci in range(3):
        lengths = [int(x) for x in g["lengths%d" % ci]]
        rs = np.random.RandomState(200 + ci)
        N = sum(lengths)
        feats = [torch.from_numpy(rs.randn(N, 200).astype(np.float32)).to(DEV).requires_grad_(True) for _ in range(3)]
        R = torch.from_numpy(rs.randn(3 * N, 3 * N).astype(np.float32)).to(DEV)
        adj = m.graph_model.create_big_adj(feats[0], feats[1], feats[2], lengths, ['a', 'v', 'l'])
        dense = adj.to_dense()
        assert np.abs(dense.detach().cpu().numpy() - g["adj%d" % ci]).max() < 2e-5
        (dense * R).sum().backward()
        got = np.stack([f.grad.cpu().numpy() for f in feats], 0)
        want = g["dfeats%d" % ci]
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-4


@pytest.mark.parametrize("cfgname,ragged", [("cfg2", False), ("cfg2", True), ("cfg3", True)])
def test_full_size_against_oracle_forward(cfgname, ragged):
    """BASELINE.json sizes, eval logits vs the CPU oracle (aten GRU engine)."""
    cfg = dict(synthetic.CONFIGS[cfgname])
    b = synthetic.make_batch(41, ragged=ragged, **cfg)
    m = hip_model(cfg, 40).eval()
    with torch.no_grad():
        got = run(m, b)
        want = O.forward({k: v.cpu() for k, v in m.state_dict().items()}, b["textf"], b["qmask"], b["umask"],
                         b["lengths"], b["acouf"], b["visuf"], O.default_cfg(cfg["nlayers"]), engine="aten")
    assert abs_err(got, want) < 1e-4


def test_three_step_training_trace_against_golden():
    g = load("train_trace.npz")
    cfg = dict(B=3, L=24, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    m = hip_model(cfg, 500)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4, weight_decay=1e-4)
    loss_f = FocalLoss(gamma=0.5)
    names = ["c%d" % i for i in range(6)]
    losses, preds = [], []
    for s, lengths in enumerate([[24, 11, 17], [9, 24, 2], [13, 13, 20]]):
        b = synthetic.make_batch(600 + s, lengths=lengths, **cfg)
        data = [b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"], ["v%d" % s]]
        res = train.train_or_eval_graph_model(m, loss_f, [data], 0, True, opt, True, 'avl', names)
        losses.append(res[2])
        preds.append(res[5])
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-4
    assert (np.concatenate(preds) == g["preds"]).mean() > 0.99
    assert np.abs(m.smax_fc.weight.detach().cpu().numpy() - g["smax_fc.weight"]).max() < 1e-5
    assert np.abs(m.graph_model.graph_net.convs[1].weight.detach().cpu().numpy() - g["convs1"]).max() < 1e-5
    assert np.abs(m.linear_a.bias.detach().cpu().numpy() - g["linear_a.bias"]).max() < 1e-5


def test_size_independent_properties_at_full_size():
    """cfg4 shard size: propagate is linear in H; every normalised-adjacency row block reproduces
    D^-1/2 A D^-1/2 (rows of A_hat . sqrt(deg) == sqrt(deg))."""
    cfg = dict(synthetic.CONFIGS["cfg4"])
    rs = np.random.RandomState(9)
    lengths = synthetic.make_lengths(rs, cfg["B"], cfg["L"], True)
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(3, N, 200).astype(np.float32)).to(DEV)
    adj = ops.build_adjacency(feats, lengths)
    H1 = torch.randn(3 * N, 100, device=DEV)
    H2 = torch.randn(3 * N, 100, device=DEV)
    lhs = ops.propagate(adj, 2.0 * H1 - 0.5 * H2)
    rhs = 2.0 * ops.propagate(adj, H1) - 0.5 * ops.propagate(adj, H2)
    assert rel_err(lhs, rhs) < 1e-5
    # A_hat = R A R with R = deg^-1/2, so A_hat . (1/r) = R . (A . 1) = R . deg = 1/r
    unit = feats / feats.norm(dim=2, keepdim=True)
    # recover r from the diagonal: A_hat[p,p] = S_pp r_p^2, S_pp = sim(|u|^2)
    lay = adj.layout
    diag = []
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        for m_ in range(3):
            diag.append(adj.tiles[base + m_ * L * ld: base + (m_ + 1) * L * ld].view(L, ld).diagonal())
    # reorder (dialogue-major, modality) -> modality-major rows
    r2 = torch.zeros(3, N, device=DEV)
    k = 0
    start = 0
    for i, L in enumerate(lengths):
        for m_ in range(3):
            r2[m_, start:start + L] = diag[k]; k += 1
        start += L
    spp = 1.0 - torch.acos((unit * unit).sum(2) * 0.99999) / np.pi
    inv_r = torch.sqrt(spp / r2).reshape(3 * N, 1).repeat(1, 4)
    out = ops.propagate(adj, inv_r.contiguous())
    assert rel_err(out, inv_r) < 1e-4


def test_mfn_fusion_variant_against_reference_golden():
    """--mm_fusion_mthd mfn after the GDF graph (model.py:1303-1326), eval logits vs the reference."""
    g = np.load(os.path.join(GOLD, "fusion_modules.npz"))
    cfg = dict(B=3, L=12, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    m = synthetic.build_model(att_type='mfn', **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 702))
    m = m.to(DEV).eval()
    b = synthetic.make_batch(703, lengths=[12, 5, 9], **cfg)
    with torch.no_grad():
        logp = run(m, b)
    assert np.abs(logp.cpu().numpy() - g["e2e_mfn_log_prob"]).max() < 1e-4
    # gradients through the memory fusion on the device: the module against the reference's own gradients (G8) ...
    from mm_dfn_amd import MFN
    rs = np.random.RandomState(700)
    mfn = MFN()
    mfn.load_state_dict(synthetic.seeded_state_dict(mfn.state_dict(), 700))
    mfn = mfn.to(DEV).eval()
    x = torch.from_numpy(rs.randn(9, 2, 900).astype(np.float32)).to(DEV).requires_grad_(True)
    R = torch.from_numpy(rs.randn(9, 2, 400).astype(np.float32)).to(DEV)
    y = mfn(x)
    (y * R).sum().backward()
    assert np.abs(y.detach().cpu().numpy() - g["mfn_y"]).max() < 1e-5
    assert rel_err(x.grad, torch.from_numpy(g["mfn_dx"])) < 1e-4
    assert rel_err(mfn.gamma1_fc1.weight.grad, torch.from_numpy(g["mfn_dW"])) < 1e-4
    # ... and gradients reach every stage of the 'mfn' model (re-pad -> MFN -> strip -> head) on the device
    m.eval()                                                 # dropouts inactive, gradients still flow
    logp = run(m, b)
    W = torch.from_numpy(np.random.RandomState(704).randn(*logp.shape).astype(np.float32)).to(DEV)
    m.zero_grad(set_to_none=True)
    (logp * W).sum().backward()
    assert m.mfn.gamma1_fc1.weight.grad is not None and torch.isfinite(m.mfn.gamma1_fc1.weight.grad).all()
    assert float(m.mfn.lstm_l.weight_hh.grad.abs().max()) > 0 and float(m.linear_l.weight.grad.abs().max()) > 0


def test_fused_flat_adam_matches_torch_adam_and_reference_trace():
    """FlatAdam (one HIP launch per step) vs torch.optim.Adam(weight_decay=l2) and vs the reference's 3-step trace."""
    from mm_dfn_amd.optim import FlatAdam
    g = load("train_trace.npz")
    cfg = dict(B=3, L=24, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    m_ref = hip_model(cfg, 500)
    m_fus = hip_model(cfg, 500)
    o_ref = torch.optim.Adam(m_ref.parameters(), lr=3e-4, weight_decay=1e-4)
    o_fus = FlatAdam(m_fus, lr=3e-4, weight_decay=1e-4)
    loss_f = FocalLoss(gamma=0.5)
    losses = []
    for s, lengths in enumerate([[24, 11, 17], [9, 24, 2], [13, 13, 20]]):
        b = synthetic.make_batch(600 + s, lengths=lengths, **cfg)
        label = train.flatten_labels(b["label"].to(DEV), b["lengths"])
        for m_, o_ in ((m_ref, o_ref), (m_fus, o_fus)):
            m_.train()
            o_.zero_grad(set_to_none=True)
            loss = loss_f(run(m_, b), label)
            loss.backward()
            o_.step()
        losses.append(round(float(loss.detach()), 4))
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-4
    for (k, p_ref), (_, p_fus) in zip(m_ref.named_parameters(), m_fus.named_parameters()):
        assert abs_err(p_fus, p_ref) < 2e-6, k
    assert np.abs(m_fus.smax_fc.weight.detach().cpu().numpy() - g["smax_fc.weight"]).max() < 1e-5
    # untouched parameters (no gradient on this path) stay bit-identical to their initial values
    init = synthetic.seeded_state_dict(m_fus.state_dict(), 500)
    assert abs_err(m_fus.gatedatt.transform_l.weight, init["gatedatt.transform_l.weight"]) == 0.0


def test_deepgcn_sibling_against_reference_golden_and_oracle():
    """graph_type='DeepGCN' on the HIP kernels (M = 1 adjacency tiles, same layer kernels): reference goldens for
    the eval log-probs of both fusions and the gate-less gradients; the gated gradients against the oracle
    (the reference itself cannot back-propagate them)."""
    from test_oracle_golden import DEEP_CFG, DEEP_LENGTHS, deep_state
    g = load("deepgcn.npz")
    b = synthetic.make_batch(801, lengths=DEEP_LENGTHS, **DEEP_CFG)
    for att in ("concat_subsequently", "gated"):
        m, sd = deep_state(att)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        with torch.no_grad():
            assert np.abs(run(m, b).cpu().numpy() - g["logp_" + att]).max() < 1e-4, att
    R = torch.from_numpy(np.random.RandomState(802).randn(*g["logp_nogate"].shape).astype(np.float32))
    m, sd = deep_state(reason_flag=False)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    logp = run(m, b)
    assert np.abs(logp.detach().cpu().numpy() - g["logp_nogate"]).max() < 1e-4
    (logp * R.to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    for k in [f[5:] for f in g.files if f.startswith("grad_")]:
        assert rel_err(named[k].grad, torch.from_numpy(g["grad_" + k])) < 5e-4, k
    # gate on: oracle autograd (functional restatement, no in-place update) is the checker
    m, sd = deep_state()
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    logp = run(m, b)
    (logp * R.to(DEV)).sum().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want = O.forward_deepgcn(params, b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"],
                             O.default_cfg(DEEP_CFG["nlayers"]), engine="aten")
    (want * R).sum().backward()
    assert abs_err(logp, want) < 1e-4
    named = dict(m.named_parameters())
    for k in ("linear_l.weight", "lstm_l.weight_hh_l0", "graph_net_a.rnn.weight_hh_l0", "graph_net_l.convs.1.weight",
              "graph_net_v.fcs.0.bias", "smax_fc.weight"):
        assert rel_err(named[k].grad, params[k].grad) < 1e-4, k


@pytest.mark.parametrize("gamma,use_alpha,size_average", [(0.0, False, True), (0.5, False, True), (1.0, True, True),
                                                          (2.0, True, False), (0.5, False, False)])
def test_fused_focal_loss_kernels_against_oracle(gamma, use_alpha, size_average):
    """K10 (one launch each way) against the oracle's restatement of loss.py:14-34, incl. pt == 1 rows (log-prob 0),
    class weights, sum reduction, and an upstream gradient that is not 1."""
    rs = np.random.RandomState(77)
    N, C = 1237, 7
    logp = torch.log_softmax(torch.from_numpy(rs.randn(N, C).astype(np.float32)) * 3, 1)
    tgt = torch.from_numpy(rs.randint(0, C, size=N))
    logp[5] = torch.tensor([0.0] + [-100.0] * (C - 1))
    tgt[5] = 0                                              # pt = 1 exactly: (1 - pt)^gamma = 0 (or 1 at gamma 0)
    alpha = torch.from_numpy(rs.uniform(0.5, 2.0, size=C).astype(np.float32)) if use_alpha else None
    lo = logp.clone().requires_grad_(True)
    want = O.focal_loss(lo, tgt, gamma, alpha, size_average)
    (want * 1.7).backward()
    lg = logp.clone().to(DEV).requires_grad_(True)
    f = FocalLoss(gamma=gamma, alpha=None if alpha is None else alpha.tolist(), size_average=size_average)
    got = f(lg, tgt.to(DEV))
    (got * 1.7).backward()
    assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    assert abs_err(lg.grad, lo.grad) <= 1e-6 * max(1.0, float(lo.grad.abs().max()))
    got2 = f(lg.detach(), tgt.to(DEV))
    assert got2.item() == got.item()                        # fixed reduction order


@pytest.mark.parametrize("name", ["l15_p3", "l33_p9", "l110_p2"])
def test_encoder_stack_on_device_against_reference_golden(name):
    """G4 on the HIP path: DialogueGNNModel.encode (fused projections, party gather, both BiGRUs in shared launches,
    combine + pad strip) against the features captured from the reference."""
    from test_oracle_golden import enc_setup
    g = load("encoders_graphconv.npz")
    cfg, sd, b, m = enc_setup(name)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    with torch.no_grad():
        feats = m.encode(b["textf"].to(DEV), b["qmask"].to(DEV), b["lengths"], b["acouf"].to(DEV), b["visuf"].to(DEV))
    assert np.abs(feats.cpu().numpy() - g["enc_" + name]).max() < 2e-5


def test_graph_convolution_module_against_reference_golden():
    """G2 on the device: the drop-in GraphConvolution with a dense adjacency tensor."""
    from mm_dfn_amd import GraphConvolution
    g = load("encoders_graphconv.npz")
    rs = np.random.RandomState(610)
    n = 37
    conv = GraphConvolution(100, 100, variant=True)
    conv.load_state_dict(synthetic.seeded_state_dict(conv.state_dict(), 611))
    conv = conv.to(DEV)
    x = torch.from_numpy(rs.randn(n, 100).astype(np.float32)).to(DEV)
    h0 = torch.from_numpy(rs.randn(n, 100).astype(np.float32)).to(DEV)
    adj = torch.from_numpy(rs.uniform(0, 1, size=(n, n)).astype(np.float32))
    adj = (adj / adj.sum(1, keepdim=True)).to(DEV)
    for l in (1, 2, 16):
        with torch.no_grad():
            assert np.abs(conv(x, adj, h0, 0.5, 0.2, l).cpu().numpy() - g["gconv_l%d" % l]).max() < 1e-5


def test_test_label_dumps_the_reference_activation_files(tmp_path, capsys):
    """--test_label (reference model_GCN.py:474-480, model.py:1297-1301): per-layer outputs and the fused (N, 900)
    features are saved under the reference's file names; the pass itself returns the same log-probabilities."""
    cfg = dict(B=3, L=20, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    model = synthetic.build_model(**cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 3))
    model = model.to(DEV).eval()
    model.graph_model.graph_net.test_output_dir = str(tmp_path) + "/"
    b = synthetic.make_batch(4, lengths=[20, 11, 5], device=DEV, **cfg)
    args = (b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])
    with torch.no_grad():
        plain = model(*args)[0]
        dumped = model(*args, True)[0]
    assert float((plain - dumped).abs().max()) < 1e-5
    N = sum(b["lengths"])
    multi = np.load(str(tmp_path / "1080_v2_test_output_multi_15.npy"))
    assert multi.shape == (N, 900)
    for i in range(2):
        layer = np.load(str(tmp_path / ("1080_v1_test_output_layer_%d.npy" % i)))
        assert layer.shape == (3 * N, 100)
    # the last layer's output is the graph part of the fused features: columns 200:300 of every modality block
    for m in range(3):
        assert np.allclose(multi[:, 300 * m + 200:300 * (m + 1)], layer[m * N:(m + 1) * N], atol=1e-6)
    assert "# deepGCN layer 0" in capsys.readouterr().out
    # model.py:1331-1335: the class scores behind smax_fc; their log-softmax is what the pass returned
    scores = np.load(str(tmp_path / "1080_v3_test_output_multi_after_relu-fc_15.npy"))
    assert scores.shape == (N, 6)
    assert np.allclose(torch.log_softmax(torch.from_numpy(scores), 1).numpy(), dumped.cpu().numpy(), atol=1e-6)
    # through the pass loop with a captured-step cache: the dumping pass is not captured (it calls .cpu() / np.save)
    from mm_dfn_amd import FocalLoss, train
    batch = [b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]]
    cache = train.StepGraphCache(model, FocalLoss(gamma=0.5))
    out = train.train_or_eval_graph_model(model, FocalLoss(gamma=0.5), [batch], train_flag=False, test_label=True,
                                          graph_cache=cache)
    assert cache.misses == 0 and len(out[5]) == N


def test_fusion_modules_run_without_library_gemms():
    """MFN and MMGatedAttention forward + backward launch only this package's kernels for their dense products: no Tensile
    (`Cijk_*`, hipBLASLt / rocBLAS) kernel in the device trace."""
    from torch.profiler import ProfilerActivity, profile
    from mm_dfn_amd import MFN, MMGatedAttention
    rs = np.random.RandomState(5)
    t = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32)).to(DEV).requires_grad_(True)
    mfn = MFN().to(DEV).train()
    gat = MMGatedAttention(300, 300).to(DEV).train()
    x, a, v, l = t(7, 3, 900), t(50, 300), t(50, 300), t(50, 300)

    def step():
        with ops.wgrad_batch():
            (mfn(x).sum() + gat(a, v, l).sum()).backward()
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("linear_lds_kernel" in n for n in names) and any("gated_pair_fwd_kernel" in n for n in names)
    assert any("softmax_scale_bwd_kernel" in n for n in names) and any("mfn_mem_fwd_kernel" in n for n in names)
    assert not [n for n in names if n.startswith("Cijk_") or "gemm" in n.lower() and "gemm_tn" not in n], names
    assert mfn.gamma2_fc2.weight.grad is not None and gat.transform_vl.weight.grad is not None


def test_mfn_and_gated_attention_modules_against_reference_golden():
    """SURVEY 8 a-13 / a-14 at module level against fixtures generated from the reference modules
    (tests/golden/fusion_modules.npz, tests/golden/make_golden.py): outputs, input gradient and a weight gradient of MFN
    (model_fusion.py:10-120), outputs of MMGatedAttention 'general' for three and two modalities (model.py:718-781)."""
    import os
    from mm_dfn_amd import MFN, MMGatedAttention, synthetic
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fusion_modules.npz"))
    rs = np.random.RandomState(700)
    mfn = MFN()
    mfn.load_state_dict(synthetic.seeded_state_dict(mfn.state_dict(), 700))
    mfn = mfn.to(DEV).eval()
    x = torch.from_numpy(rs.randn(9, 2, 900).astype(np.float32)).to(DEV).requires_grad_(True)
    R = torch.from_numpy(rs.randn(9, 2, 400).astype(np.float32)).to(DEV)
    y = mfn(x)
    (y * R).sum().backward()
    assert np.abs(y.detach().cpu().numpy() - g["mfn_y"]).max() < 1e-5
    assert np.abs(x.grad.cpu().numpy() - g["mfn_dx"]).max() / np.abs(g["mfn_dx"]).max() < 1e-4
    assert np.abs(mfn.gamma1_fc1.weight.grad.cpu().numpy() - g["mfn_dW"]).max() / np.abs(g["mfn_dW"]).max() < 1e-4
    ga = MMGatedAttention(300, 100, att_type='general')
    ga.load_state_dict(synthetic.seeded_state_dict(ga.state_dict(), 701))
    ga = ga.to(DEV).eval()
    a, v, l = (torch.from_numpy(rs.randn(11, 300).astype(np.float32)).to(DEV) for _ in range(3))
    with torch.no_grad():
        assert np.abs(ga(a, v, l, ['a', 'v', 'l']).cpu().numpy() - g["gated_avl"]).max() < 2e-6
        assert np.abs(ga(a, [], l, ['a', 'l']).cpu().numpy() - g["gated_al"]).max() < 2e-6
