def test_lstm_gate_backward_bf16_piece_form_against_the_exact_f32_form(kernel_variants):
    """The many-row K8 backward (csrc/lstm_gate_split.hip, round 5: gate gradients formed in the contraction's A layout, the
    index walked as (unit block, gate), six bf16-piece MFMA products) against the exact-f32 producer / consumer kernel on the
    same inputs, both against fp64: dG / dc_prev are the same arithmetic, dq / dh_prev at most 4x the exact-f32 kernel's
    error; forced below the row threshold too (ragged last row block, narrow H, first layer, absent optional inputs)."""
    from mm_dfn_amd import _hip
    P, st = _hip.ptr, _hip.stream
    for R, H, first, full in ((24576, 100, False, True), (301, 100, False, False), (1000, 36, True, True), (130, 96, False, True),
                              (515, 12, False, True)):
        rs = np.random.RandomState(86)
        gates = torch.sigmoid(_rnd(rs, R, 4 * H))
        gates[:, 2 * H:3 * H] = gates[:, 2 * H:3 * H] * 2 - 1                   # the g gate is a tanh
        c, c_new = _rnd(rs, R, H), _rnd(rs, R, H)
        dh_a, dh_b, dc_n, dres_w = _rnd(rs, R, H), _rnd(rs, R, H), _rnd(rs, R, H), _rnd(rs, R, H + 12)
        dres = dres_w[:, 4:4 + H]
        Wih, Whh = _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H, H, scale=0.2)
        if not full:
            dh_b = dc_n = dres = None
        res = {}
        for mode in ("0", "1"):
            kernel_variants.setenv("MMDFN_GATE_BWD_SPLIT", mode)
            dG, dq = torch.full((R, 4 * H), float("nan"), device=DEV), torch.full((R, H), float("nan"), device=DEV)
            dcp = None if first else torch.full((R, H), float("nan"), device=DEV)
            dhp = None if first else torch.full((R, H), float("nan"), device=DEV)
            assert _hip.lib().mmdfn_lstm_gate_bwd(P(gates), None if first else P(c), P(c_new), P(dh_a), P(dh_b), P(dc_n), P(Wih),
                                                  P(Whh), P(dres), P(dG), P(dcp), P(dq), P(dhp), R, H, 0 if first else 1,
                                                  H + 12 if dres is not None else 0, st()) == 0
            res[mode] = (dG, dq, dcp, dhp)
        d = lambda t: None if t is None else t.double().cpu()
        gi, gf, gg, go = (d(gates)[:, k * H:(k + 1) * H] for k in range(4))
        tc = torch.tanh(d(c_new))
        dhv = d(dh_a) + (d(dh_b) if dh_b is not None else 0)
        dc = (d(dc_n) if dc_n is not None else 0) + dhv * go * (1 - tc * tc)
        dG_w = torch.cat([dc * gg * gi * (1 - gi), (dc * d(c) * gf * (1 - gf)) if not first else torch.zeros_like(dc),
                          dc * gi * (1 - gg * gg), dhv * tc * go * (1 - go)], 1)
        dq_w = dG_w @ d(Wih) + (d(dres) if dres is not None else 0)
        want = (dG_w, dq_w, None if first else dc * gf, None if first else dG_w @ d(Whh))
        for k in range(4):
            if want[k] is None:
                continue
            e32 = float((res["0"][k].double().cpu() - want[k]).abs().max())
            esp = float((res["1"][k].double().cpu() - want[k]).abs().max())
            assert torch.isfinite(res["1"][k]).all()
            assert esp <= 4 * e32 + 2e-6 * float(want[k].abs().max()), (R, H, k, esp, e32)
        assert float((res["1"][1] - res["0"][1]).abs().max()) > 0.0             # it really was a different kernel


