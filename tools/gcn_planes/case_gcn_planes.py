"""(Not collected by pytest: the parity cases that were run against the experiment, see README.md.)  GPU: the many-row forms of K7 forward / backward and K8 backward on bf16 piece planes (csrc/gcn_planes.hip) against fp64 and
against the exact-f32 kernels of csrc/gcn_stack.hip they stand in for (same operands, same layouts)."""
import numpy as np
import pytest
import torch

from mm_dfn_amd import _hip, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).to(DEV)


def _keep(rs, *shape, p=0.5):
    return torch.from_numpy((rs.uniform(size=shape) > p).astype(np.float32)).to(DEV)


def rel_err(got, want):
    want = want.double().cpu()
    return float((got.double().cpu() - want).abs().max()) / (float(want.abs().max()) + 1e-30)


@pytest.mark.parametrize("R,H,masked,has_q,ldo", [(5280, 100, True, True, 100), (200, 100, False, False, 300), (16384 + 77, 100, True, True, 300),
                                                   (70, 36, True, False, 36), (3000, 64, False, True, 64)])
def test_gcnii_layer_plane_kernels(R, H, masked, has_q, ldo):
    lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
    rs = np.random.RandomState(183)
    hi, h0 = _rnd(rs, R, H), _rnd(rs, R, H)
    W = torch.nn.Parameter(_rnd(rs, 2 * H, H, scale=0.1))
    q = _rnd(rs, R, H) if has_q else None
    m = _keep(rs, R, H) if masked else None
    theta, alpha, ms = 0.405, 0.2, 2.0
    wide = torch.full((R, ldo), 3.0, device=DEV)
    out = wide[:, ldo - H:]
    gmask = torch.empty(R, H, device=DEV)
    pf = ops.weight_planes(W, mode=1)
    assert lib.mmdfn_gcnii_layer_fwd_planes(P(hi), P(h0), P(pf.buf), P(q), P(m), P(out), P(gmask), theta, alpha, R, H, ldo, ms, st()) == 0
    d = lambda t: None if t is None else t.detach().double().cpu()
    hid, h0d = d(hi).requires_grad_(True), d(h0).requires_grad_(True)
    pre = theta * (torch.cat([hid, h0d], 1) @ d(W)) + (1 - theta) * ((1 - alpha) * hid + alpha * h0d)
    want = torch.relu(pre) * (d(m) * ms if masked else 1.0) + (d(q) if has_q else 0.0)
    assert rel_err(out, want) < 2e-6
    assert ldo == H or float(wide[:, :ldo - H].min()) == 3.0
    # the exact-f32 kernel on the same operands: same output to rounding, same ReLU decisions except within rounding of the kink
    out0, gmask0 = torch.empty(R, H, device=DEV), torch.empty(R, H, device=DEV)
    assert lib.mmdfn_gcnii_layer_fwd(P(hi), P(h0), P(W), P(q), P(m), P(out0), P(gmask0), theta, alpha, R, H, H, ms, st()) == 0
    assert rel_err(out, out0) < 2e-6
    differ = (gmask != gmask0).cpu()
    assert float(pre.detach().abs()[differ].max() if differ.any() else 0.0) < 1e-5
    assert torch.equal((gmask.cpu() != 0) | differ, (((pre > 0) & ((d(m) != 0) if masked else torch.ones_like(pre, dtype=torch.bool))) | differ))
    # backward (on the plane kernel's own mask)
    dwide = _rnd(rs, R, ldo)
    dout = dwide[:, ldo - H:]
    gm = gmask.double().cpu()
    dP, dhi_w = torch.empty(R, H, device=DEV), torch.full((R, 2 * H), 7.0, device=DEV)
    dhi = dhi_w[:, H:]
    dh0 = _rnd(rs, R, H)
    old = dh0.clone()
    pb = ops.weight_planes(W, mode=0)
    assert lib.mmdfn_gcnii_layer_bwd_planes(P(dout), P(gmask), P(pb.buf), P(dP), P(dhi), P(dh0), theta, alpha, R, H, ldo, 1, 2 * H, st()) == 0
    gg = d(dout) * gm
    wantP = theta * gg
    Wd = d(W)
    want_hi = wantP @ Wd[:H].t() + (1 - theta) * (1 - alpha) * gg
    want_h0 = wantP @ Wd[H:].t() + (1 - theta) * alpha * gg
    assert rel_err(dP, wantP) < 1e-6
    assert rel_err(dhi, want_hi) < 3e-6 and float(dhi_w[:, :H].min()) == 7.0
    assert rel_err(dh0, d(old) + want_h0) < 3e-6
    assert lib.mmdfn_gcnii_layer_bwd_planes(P(dout), P(gmask), P(pb.buf), P(dP), P(dhi), P(dh0), theta, alpha, R, H, ldo, 0, 2 * H, st()) == 0
    assert rel_err(dh0, want_h0) < 3e-6
    # ... and against the exact-f32 kernel
    dP0, dhi0, dh00 = torch.empty(R, H, device=DEV), torch.empty(R, H, device=DEV), torch.empty(R, H, device=DEV)
    assert lib.mmdfn_gcnii_layer_bwd(P(dout), P(gmask), P(W), P(dP0), P(dhi0), P(dh00), theta, alpha, R, H, ldo, 0, st()) == 0
    assert rel_err(dP, dP0) < 1e-6 and rel_err(dhi, dhi0) < 3e-6 and rel_err(dh0, dh00) < 3e-6


@pytest.mark.parametrize("R,H,has_h,two_dh,has_dc,lddres", [(5280, 100, True, True, True, 300), (16384 + 5, 100, True, False, True, 100),
                                                           (777, 100, False, False, False, 100), (130, 36, True, True, False, 0),
                                                           (2000, 64, True, True, True, 64)])
def test_lstm_gate_backward_plane_kernel(R, H, has_h, two_dh, has_dc, lddres):
    lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
    rs = np.random.RandomState(184)
    gates = torch.cat([torch.sigmoid(_rnd(rs, R, H)), torch.sigmoid(_rnd(rs, R, H)), torch.tanh(_rnd(rs, R, H)),
                       torch.sigmoid(_rnd(rs, R, H))], 1).contiguous()
    c_prev = _rnd(rs, R, H) if has_h else None
    c_new = _rnd(rs, R, H)
    dh_a = _rnd(rs, R, H)
    dh_b = _rnd(rs, R, H) if two_dh else None
    dc_next = _rnd(rs, R, H) if has_dc else None
    w_ih = torch.nn.Parameter(_rnd(rs, 4 * H, H, scale=0.1))
    w_hh = torch.nn.Parameter(_rnd(rs, 4 * H, H, scale=0.1))
    dres_w = _rnd(rs, R, lddres) if lddres else None
    dres = dres_w[:, lddres - H:] if lddres else None

    def run(planes):
        dG, dq = torch.empty(R, 4 * H, device=DEV), torch.empty(R, H, device=DEV)
        dcp = torch.empty(R, H, device=DEV) if has_h else None
        dhp = torch.empty(R, H, device=DEV) if has_h else None
        if planes:
            pl = ops.weight_planes(w_ih, w_hh if has_h else None, mode=3)
            rc = lib.mmdfn_lstm_gate_bwd_planes(P(gates), P(c_prev), P(c_new), P(dh_a), P(dh_b), P(dc_next), P(pl.buf), P(dres), P(dG),
                                                P(dcp), P(dq), P(dhp), R, H, 1 if has_h else 0, lddres or H, st())
        else:
            rc = lib.mmdfn_lstm_gate_bwd(P(gates), P(c_prev), P(c_new), P(dh_a), P(dh_b), P(dc_next), P(w_ih), P(w_hh), P(dres), P(dG),
                                         P(dcp), P(dq), P(dhp), R, H, 1 if has_h else 0, lddres or H, st())
        assert rc == 0
        return dG, dcp, dq, dhp

    got, ref = run(True), run(False)
    d = lambda t: t.detach().double().cpu()
    gi, gf, gg, go = [d(gates)[:, i * H:(i + 1) * H] for i in range(4)]
    tc = torch.tanh(d(c_new))
    dhv = d(dh_a) + (d(dh_b) if two_dh else 0.0)
    dc = (d(dc_next) if has_dc else 0.0) + dhv * go * (1 - tc * tc)
    dG = torch.cat([dc * gg * gi * (1 - gi), dc * (d(c_prev) if has_h else 0.0) * gf * (1 - gf), dc * gi * (1 - gg * gg),
                    dhv * tc * go * (1 - go)], 1)
    assert rel_err(got[0], dG) < 3e-6 and rel_err(got[0], ref[0]) < 1e-6
    want_q = dG @ d(w_ih) + (d(dres) if lddres else 0.0)
    assert rel_err(got[2], want_q) < 3e-6 and rel_err(got[2], ref[2]) < 3e-6
    if has_h:
        assert rel_err(got[1], dc * gf) < 3e-6 and rel_err(got[1], ref[1]) < 1e-6
        assert rel_err(got[3], dG @ d(w_hh)) < 3e-6 and rel_err(got[3], ref[3]) < 3e-6


def test_fused_stack_on_plane_kernels_matches_the_exact_kernels(monkeypatch):
    """The fused GCN stack node with K7 / K8' forced onto the plane kernels (thresholds lowered) against the same node on the
    exact-f32 kernels: output, feature gradient and every parameter gradient."""
    from mm_dfn_amd import GCNII_lyc, gcn_stack, synthetic
    rs = np.random.RandomState(185)
    lengths = [9, 4, 17, 30]
    N = sum(lengths)
    feats0 = _rnd(rs, 3, N, 200)
    Rw = _rnd(rs, 3 * N, 300)
    res = []
    for rows in (1 << 30, 1):
        monkeypatch.setattr(gcn_stack, "K7_PLANES_ROWS", rows)
        monkeypatch.setattr(gcn_stack, "K8_PLANES_ROWS", rows)
        net = GCNII_lyc(nfeat=200, nlayers=3, nhidden=100, nclass=6, dropout=0.0, lamda=0.5, alpha=0.2, variant=True,
                        return_feature=True, use_residue=True, reason_flag=True)
        net.load_state_dict(synthetic.seeded_state_dict(net.state_dict(), 85))
        net = net.to(DEV).train()
        feats = feats0.clone().requires_grad_(True)
        adj = ops.build_adjacency(feats, lengths)
        y = gcn_stack.gcn_stack(adj.stacked_feats.reshape(3 * N, 200), adj, None, 1.0, net.lamda, net.alpha, True, True,
                                net.fcs[0].weight, net.fcs[0].bias, net.rnn, [c.weight for c in net.convs])
        from mm_dfn_amd import train
        train.backward((y * Rw).sum())
        res.append((y.detach(), feats.grad.clone(), {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}))
    (y0, g0, p0), (y1, g1, p1) = res
    assert rel_err(y1, y0) < 5e-6 and rel_err(g1, g0) < 2e-5
    assert p0.keys() == p1.keys()
    for k in p0:
        assert rel_err(p1[k], p0[k]) < 2e-5, k
