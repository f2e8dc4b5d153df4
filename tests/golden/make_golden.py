"""Export golden vectors from the REAL reference (run only in the build container).

    python tests/golden/make_golden.py

Imports /root/reference through oracle/ref_shim.py and writes small .npz
fixtures next to this file.  Inputs and weights are NOT stored: both sides
regenerate them from np.random.RandomState seeds through
mm_dfn_amd/synthetic.py, so only reference OUTPUTS are committed.

Gradients come from the reference in train() mode with dropout=1e-12: that is
numerically "dropout off" (mask all ones, scale 1/(1-1e-12) == 1.0f) but avoids
the reference's autograd error in eval()/p=0, where ``layer_inner += q``
(model_GCN.py:472) modifies the ReLU output in place.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_shim  # noqa: E402
from mm_dfn_amd import synthetic  # noqa: E402

TINY = 1e-12

E2E_CASES = {
    # name: (cfg, seed, lengths)
    "iemocap_b1": (dict(B=1, L=110, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512), 101, [110]),
    "iemocap_ragged_refdims": (dict(B=4, L=60, P=2, C=6, nlayers=2, D_t=100, D_a=1582, D_v=342), 102, [60, 27, 41, 33]),
    "meld_like": (dict(B=6, L=33, P=9, C=7, nlayers=4, D_t=600, D_a=300, D_v=342), 103, [33, 3, 17, 9, 24, 1]),
    "deep16": (dict(B=3, L=20, P=2, C=6, nlayers=16, D_t=100, D_a=100, D_v=512), 104, [20, 13, 7]),
}


def ref_model(cfg, seed, dropout):
    m = ref_shim.build_reference_model(cfg["D_t"], cfg["D_a"], cfg["D_v"], cfg["P"], cfg["C"], cfg["nlayers"],
                                       dropout=dropout)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), seed))
    return m


def grad_digest(g):
    g = g.detach().double().reshape(-1)
    return np.array([g.sum().item(), g.abs().sum().item(), (g * g).sum().item()], dtype=np.float64)


def export_e2e():
    _, _, _, ref_loss = ref_shim.modules()
    for name, (cfg, seed, lengths) in E2E_CASES.items():
        batch = synthetic.make_batch(seed + 1, lengths=lengths, **cfg)
        args = (batch["textf"], batch["qmask"], batch["umask"], batch["lengths"], batch["acouf"], batch["visuf"])
        m = ref_model(cfg, seed, 0.0).eval()
        with torch.no_grad():
            logp = m(*args)[0]
        out = {"log_prob": logp.numpy()}
        m = ref_model(cfg, seed, TINY).train()
        logp_t = m(*args)[0]
        label = torch.cat([batch["label"][j][:n] for j, n in enumerate(batch["lengths"])])
        loss = ref_loss.FocalLoss(gamma=0.5)(logp_t, label)
        loss.backward()
        out["loss"] = np.array(loss.item(), dtype=np.float64)
        out["train_log_prob_maxdiff"] = np.array((logp_t.detach() - logp).abs().max().item())
        names = []
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            out["gd/" + k] = grad_digest(p.grad)
        for k in ("smax_fc.weight", "graph_model.graph_net.convs.0.weight", "graph_model.graph_net.rnn.weight_hh_l0",
                  "rnn_parties.weight_hh_l0", "lstm_l.weight_hh_l1_reverse", "linear_l.weight"):
            out["g/" + k] = dict(m.named_parameters())[k].grad.numpy()
        out["live_params"] = np.array(names)
        np.savez_compressed(os.path.join(HERE, "e2e_%s.npz" % name), **out)
        print(name, "N=%d" % sum(lengths), "loss", loss.item(), "train-vs-eval", out["train_log_prob_maxdiff"])


VARIANT_CASES = {
    # name: (modals, av_using_lstm, speaker_weights, cfg, seed, lengths) -- the two flags of the same GDF path the reference driver
    # exposes next to the MM-DFN scripts' setting (run_train_erc.py --modals, --av_using_lstm; model.py:851-868,1067,1096)
    "av": ("av", False, "3-0-1", dict(B=3, L=20, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512), 111, [20, 13, 7]),
    "al": ("al", False, "3-0-1", dict(B=3, L=20, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512), 112, [20, 13, 7]),
    "vl": ("vl", False, "1-2-1", dict(B=4, L=17, P=3, C=7, nlayers=3, D_t=100, D_a=100, D_v=342), 113, [17, 9, 1, 12]),
    "avl_lstm": ("avl", True, "3-0-1", dict(B=3, L=20, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512), 114, [20, 13, 7]),
    "al_lstm": ("al", True, "1-1-1", dict(B=2, L=15, P=2, C=6, nlayers=2, D_t=100, D_a=1582, D_v=342), 115, [15, 6]),
}


def export_variants():
    """Bimodal graphs and av_using_lstm: eval log-probabilities, loss, gradient digests of every live parameter, a few
    full gradients, and the state_dict key list (the modules of absent modalities do not exist in the reference)."""
    _, _, _, ref_loss = ref_shim.modules()
    out = {}
    for name, (modals, lstm, sw, cfg, seed, lengths) in VARIANT_CASES.items():
        batch = synthetic.make_batch(seed + 1, lengths=lengths, **cfg)
        args = (batch["textf"], batch["qmask"], batch["umask"], batch["lengths"], batch["acouf"], batch["visuf"])

        def build(dropout):
            m = ref_shim.build_reference_model(cfg["D_t"], cfg["D_a"], cfg["D_v"], cfg["P"], cfg["C"], cfg["nlayers"],
                                               dropout=dropout, speaker_weights=sw, modals=modals, av_using_lstm=lstm)
            m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), seed))
            return m
        m = build(0.0).eval()
        with torch.no_grad():
            out[name + "/log_prob"] = m(*args)[0].numpy()
        with open(os.path.join(HERE, "state_dict_keys_variant_%s.txt" % name), "w") as f:
            for k, v in m.state_dict().items():
                f.write("%s %s\n" % (k, " ".join(str(d) for d in v.shape)))
        m = build(TINY).train()
        logp = m(*args)[0]
        label = torch.cat([batch["label"][j][:n] for j, n in enumerate(batch["lengths"])])
        loss = ref_loss.FocalLoss(gamma=0.5)(logp, label)
        loss.backward()
        out[name + "/loss"] = np.array(loss.item(), dtype=np.float64)
        live = []
        for k, p in m.named_parameters():
            if p.grad is not None and float(p.grad.abs().max()) > 0:
                live.append(k)
                out[name + "/gd/" + k] = grad_digest(p.grad)
        out[name + "/live_params"] = np.array(live)
        for k in ("smax_fc.weight", "rnn_parties.weight_hh_l0", "graph_model.graph_net.convs.0.weight") + \
                (("lstm_a.weight_hh_l1_reverse",) if lstm and "a" in modals else ()):
            out[name + "/g/" + k] = dict(m.named_parameters())[k].grad.numpy()
        print("variant", name, "N=%d" % sum(lengths), "loss", loss.item(), "live", len(live), "keys", len(m.state_dict()))
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)


def export_adjacency():
    ref = ref_shim.build_reference_model(100, 100, 512, 2, 6, 2)
    out = {}
    for ci, lengths in enumerate([[5], [7, 3, 1], [20, 13]]):
        rs = np.random.RandomState(200 + ci)
        N = sum(lengths)
        feats = [torch.from_numpy(rs.randn(N, 200).astype(np.float32)).requires_grad_(True) for _ in range(3)]
        R = torch.from_numpy(rs.randn(3 * N, 3 * N).astype(np.float32))
        A = ref.graph_model.create_big_adj(feats[0], feats[1], feats[2], lengths, ['a', 'v', 'l'])
        (A * R).sum().backward()
        out["adj%d" % ci] = A.detach().numpy()
        out["dfeats%d" % ci] = np.stack([f.grad.numpy() for f in feats], 0)
        out["lengths%d" % ci] = np.array(lengths)
    np.savez_compressed(os.path.join(HERE, "adjacency.npz"), **out)
    print("adjacency ok")


def export_gcnii():
    _, _, ref_gcn, _ = ref_shim.modules()
    out = {}
    for ci, (nl, reason) in enumerate([(2, True), (4, True), (3, False)]):
        rs = np.random.RandomState(300 + ci)
        lengths = [9, 4]
        N = sum(lengths)
        g = ref_gcn.GCNII_lyc(nfeat=200, nlayers=nl, nhidden=100, nclass=6, dropout=TINY, lamda=0.5, alpha=0.2,
                              variant=True, return_feature=True, use_residue=True, reason_flag=reason)
        g.load_state_dict(synthetic.seeded_state_dict(g.state_dict(), 300 + ci))
        g.train()
        x = torch.from_numpy(rs.randn(3 * N, 200).astype(np.float32)).requires_grad_(True)
        feats = [torch.from_numpy(rs.randn(N, 200).astype(np.float32)) for _ in range(3)]
        ref = ref_shim.build_reference_model(100, 100, 512, 2, 6, 2)
        adj = ref.graph_model.create_big_adj(feats[0], feats[1], feats[2], lengths, ['a', 'v', 'l']).detach()
        R = torch.from_numpy(rs.randn(3 * N, 300).astype(np.float32))
        y = g(x, lengths, None, adj)
        (y * R).sum().backward()
        out["y%d" % ci] = y.detach().numpy()
        out["dx%d" % ci] = x.grad.numpy()
        out["dW0_%d" % ci] = g.convs[0].weight.grad.numpy()
        out["cfg%d" % ci] = np.array([nl, int(reason)])
    np.savez_compressed(os.path.join(HERE, "gcnii.npz"), **out)
    print("gcnii ok")


def export_focal():
    _, _, _, ref_loss = ref_shim.modules()
    rs = np.random.RandomState(400)
    logp = torch.log_softmax(torch.from_numpy(rs.randn(50, 6).astype(np.float32)), 1).requires_grad_(True)
    tgt = torch.from_numpy(rs.randint(0, 6, size=50))
    alpha = torch.from_numpy(rs.uniform(0.5, 2.0, size=6).astype(np.float32))
    out = {}
    for tag, a, gamma in (("plain", None, 0.0), ("g05", None, 0.5), ("alpha_g1", alpha, 1.0)):
        if logp.grad is not None:
            logp.grad = None
        l = ref_loss.FocalLoss(gamma=gamma, alpha=a)(logp, tgt)
        l.backward()
        out["loss_" + tag] = np.array(l.item())
        out["grad_" + tag] = logp.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "focal.npz"), **out)
    print("focal ok")


def export_train_trace():
    """Three optimisation steps through the reference's own train_or_eval_graph_model."""
    ref_shim.install()
    import run_train_erc as ref_train
    cfg = dict(B=3, L=24, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    ref_train.args = argparse.Namespace(seed=2021, multi_modal=True, mm_fusion_mthd='concat_subsequently')
    m = ref_model(cfg, 500, TINY)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4, weight_decay=1e-4)
    loss_f = ref_train.FocalLoss(gamma=0.5)
    loader = []
    for s, lengths in enumerate([[24, 11, 17], [9, 24, 2], [13, 13, 20]]):
        b = synthetic.make_batch(600 + s, lengths=lengths, **cfg)
        loader.append([b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"], ["v%d" % s]])
    names = ["c%d" % i for i in range(6)]
    losses = []
    preds_all = []
    for step in range(3):
        res = ref_train.train_or_eval_graph_model(m, loss_f, [loader[step]], 0, True, opt, False, 'avl', names)
        losses.append(res[2])
        preds_all.append(res[5])
    out = {"losses": np.array(losses), "preds": np.concatenate(preds_all),
           "smax_fc.weight": m.smax_fc.weight.detach().numpy(),
           "convs1": m.graph_model.graph_net.convs[1].weight.detach().numpy(),
           "linear_a.bias": m.linear_a.bias.detach().numpy()}
    np.savez_compressed(os.path.join(HERE, "train_trace.npz"), **out)
    print("train trace", losses)


def export_fusion_modules():
    """G8: MFN (model_fusion.py:10-120) and MMGatedAttention('general') (model.py:718-781) at module level, and the
    GDF model with --mm_fusion_mthd mfn end to end (needs torch.cuda.is_available faked True: model.py:1307,1325)."""
    ref_shim.install()
    import model_fusion
    import model as ref_model_mod
    out = {}
    rs = np.random.RandomState(700)
    mfn = model_fusion.MFN()
    mfn.load_state_dict(synthetic.seeded_state_dict(mfn.state_dict(), 700))
    mfn.eval()
    x = torch.from_numpy(rs.randn(9, 2, 900).astype(np.float32)).requires_grad_(True)
    R = torch.from_numpy(rs.randn(9, 2, 400).astype(np.float32))
    y = mfn(x)
    (y * R).sum().backward()
    out["mfn_y"] = y.detach().numpy()
    out["mfn_dx"] = x.grad.numpy()
    out["mfn_dW"] = mfn.gamma1_fc1.weight.grad.numpy()
    g = ref_model_mod.MMGatedAttention(300, 100, att_type='general')
    g.load_state_dict(synthetic.seeded_state_dict(g.state_dict(), 701))
    g.eval()
    a, v, l = (torch.from_numpy(rs.randn(11, 300).astype(np.float32)) for _ in range(3))
    out["gated_avl"] = g(a, v, l, ['a', 'v', 'l']).detach().numpy()
    out["gated_al"] = g(a, [], l, ['a', 'l']).detach().numpy()
    real = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        cfg = dict(B=3, L=12, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
        m = ref_shim.build_reference_model(100, 100, 512, 2, 6, 2, att_type='mfn')
        m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 702))
        m.eval()
        b = synthetic.make_batch(703, lengths=[12, 5, 9], **cfg)
        with torch.no_grad():
            out["e2e_mfn_log_prob"] = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0].numpy()
    finally:
        torch.cuda.is_available = real
    np.savez_compressed(os.path.join(HERE, "fusion_modules.npz"), **out)
    print("fusion modules ok")


def export_deepgcn():
    """Sibling graph type 'DeepGCN' (model.py:922-941, 1242-1290): eval log-probs for both fusions the build supports,
    and gradients of sum(logp * R) in train() with dropout 1e-12 (see module docstring)."""
    _, _, _, _ = ref_shim.modules()
    out = {}
    cfg = dict(B=3, L=14, P=2, C=6, nlayers=3, D_t=100, D_a=100, D_v=512)
    lengths = [14, 5, 9]
    batch = synthetic.make_batch(801, lengths=lengths, **cfg)
    args = (batch["textf"], batch["qmask"], batch["umask"], batch["lengths"], batch["acouf"], batch["visuf"])
    for att in ("concat_subsequently", "gated"):
        build = lambda p, rf=True: ref_shim.build_reference_model(cfg["D_t"], cfg["D_a"], cfg["D_v"], cfg["P"], cfg["C"],
                                                                  cfg["nlayers"], dropout=p, graph_type="DeepGCN",
                                                                  att_type=att, reason_flag=rf)
        m = build(0.0)
        sd = synthetic.seeded_state_dict(m.state_dict(), 800)
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            out["logp_" + att] = m(*args)[0].numpy()
        if att == "concat_subsequently":
            with open(os.path.join(HERE, "state_dict_keys_deepgcn.txt"), "w") as fh:
                fh.write("\n".join("%s %s" % (k, "x".join(map(str, v.shape))) for k, v in m.state_dict().items()) + "\n")
            # gradients with reason_flag=False: with the gate on, the reference's own backward fails in every mode
            # ("layer_inner += q" (model_GCN.py:277) overwrites the ReLU output autograd saved; GCNII has no
            # dropout between the two that would make a fresh tensor, unlike GCNII_lyc)
            mt = build(TINY, False)
            mt.load_state_dict(sd)
            mt.eval()
            with torch.no_grad():
                out["logp_nogate"] = mt(*args)[0].numpy()
            mt.train()
            logp = mt(*args)[0]
            R = torch.from_numpy(np.random.RandomState(802).randn(*logp.shape).astype(np.float32))
            (logp * R).sum().backward()
            named = dict(mt.named_parameters())
            for k in ("linear_a.weight", "rnn_parties.weight_hh_l0", "graph_net_a.convs.2.weight", "graph_net_v.fcs.0.weight",
                      "graph_net_l.convs.0.weight", "smax_fc.weight"):
                out["grad_" + k] = named[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "deepgcn.npz"), **out)
    print("deepgcn ok")


ENC_CASES = {  # name: (L, P, lengths): SURVEY 8c G4
    "l15_p3": (15, 3, [15, 6, 11]),
    "l33_p9": (33, 9, [33, 9, 1, 20]),
    "l110_p2": (110, 2, [110, 47]),
}


def export_encoders_and_graphconv():
    """G4: the speaker-aware encoder stack in isolation (what DialogueGNNModel hands to MM_GCN: features_a / _v / _l,
    model.py:1062-1209), captured with a forward pre-hook on graph_model.  G2: GraphConvolution.forward at layer
    indices 1, 2, 16 (model_GCN.py:176-189)."""
    _, _, ref_gcn, _ = ref_shim.modules()
    out = {}
    for name, (L, P, lengths) in ENC_CASES.items():
        cfg = dict(B=len(lengths), L=L, P=P, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
        m = ref_model(cfg, 600, 0.0).eval()
        batch = synthetic.make_batch(601, lengths=lengths, **cfg)
        grabbed = []
        h = m.graph_model.register_forward_pre_hook(lambda mod, args: grabbed.append([t.detach().clone() for t in args[:3]]))
        with torch.no_grad():
            m(batch["textf"], batch["qmask"], batch["umask"], batch["lengths"], batch["acouf"], batch["visuf"])
        h.remove()
        a, v, l = grabbed[0]
        out["enc_%s" % name] = torch.stack([a, v, l], 0).numpy()
    rs = np.random.RandomState(610)
    n = 37
    conv = ref_gcn.GraphConvolution(100, 100, variant=True)
    conv.load_state_dict(synthetic.seeded_state_dict(conv.state_dict(), 611))
    x = torch.from_numpy(rs.randn(n, 100).astype(np.float32))
    h0 = torch.from_numpy(rs.randn(n, 100).astype(np.float32))
    adj = torch.from_numpy(rs.uniform(0, 1, size=(n, n)).astype(np.float32))
    adj = adj / adj.sum(1, keepdim=True)
    for l in (1, 2, 16):
        with torch.no_grad():
            out["gconv_l%d" % l] = conv(x, adj, h0, 0.5, 0.2, l).numpy()
    np.savez_compressed(os.path.join(HERE, "encoders_graphconv.npz"), **out)
    print("encoders + graphconv ok")


def export_state_keys():
    m = ref_shim.build_reference_model(100, 1582, 342, 2, 6, 2)
    keys = ["%s %s" % (k, "x".join(map(str, v.shape))) for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys_iemocap.txt"), "w") as fh:
        fh.write("\n".join(keys) + "\n")
    print("keys", len(keys))


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--variants-only" in sys.argv:          # (round 5: the bimodal / av_using_lstm fixtures, without touching the others)
        export_variants()
        sys.exit(0)
    export_variants()
    export_state_keys()
    export_fusion_modules()
    export_encoders_and_graphconv()
    export_deepgcn()
    export_focal()
    export_adjacency()
    export_gcnii()
    export_e2e()
    export_train_trace()
