"""GPU parity of the fused GRU recurrence (K2) against the oracle's explicit GRU equations."""
import numpy as np
import pytest
import torch

import mmdfn_oracle as O
import mmdfn_vectorised as V
from mm_dfn_amd import gru as fused
from mm_dfn_amd import synthetic
from util import rel_err, abs_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_gru(seed):
    g = torch.nn.GRU(200, 100, num_layers=2, bidirectional=True)
    g.load_state_dict(synthetic.seeded_state_dict(g.state_dict(), seed, scale=1.5))
    return g


@pytest.mark.parametrize("shapes", [[(7, 3)], [(1, 5)], [(110, 16), (110, 96)], [(33, 40), (33, 700)], [(20, 300)],
                                    [(12, 1), (5, 2), (9, 130)]])
def test_bigru2_forward_backward(shapes):
    rs = np.random.RandomState(len(shapes) * 100 + shapes[0][0])
    grus = [make_gru(50 + i) for i in range(len(shapes))]
    xs = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)) for T, R in shapes]
    ws = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)) for T, R in shapes]
    # oracle: explicit GRU equations on CPU
    want, wgrads, xgrads = [], [], []
    for g, x, w in zip(grus, xs, ws):
        params = {"g." + k: v.detach().clone().requires_grad_(True) for k, v in g.state_dict().items()}
        xo = x.clone().requires_grad_(True)
        y = O.bigru2(xo, params, "g.", engine="manual")
        (y * w).sum().backward()
        want.append(y.detach())
        wgrads.append({k[2:]: v.grad for k, v in params.items()})
        xgrads.append(xo.grad)
    gd = [g.to(DEV) for g in grus]
    xg = [x.to(DEV).requires_grad_(True) for x in xs]
    ys = fused.bigru2(xg, gd, 0.0, True)
    sum((y * w.to(DEV)).sum() for y, w in zip(ys, ws)).backward()
    for i in range(len(shapes)):
        assert abs_err(ys[i], want[i]) < 2e-6
        assert rel_err(xg[i].grad, xgrads[i]) < 2e-5
        for k, p in gd[i].named_parameters():
            assert rel_err(p.grad, wgrads[i][k]) < 5e-5, k


@pytest.mark.parametrize("shapes", [[(7, 3)], [(110, 16), (110, 64)], [(12, 1), (5, 2), (9, 120)]])
def test_one_sequence_per_workgroup_backward_kernels_agree(shapes, kernel_variants):
    """Batches small enough for one sequence per workgroup in ONE round (<= 128 sequences) run their backward pass on the wave-partitioned kernel (gate rows
    split over 8 waves, operands through v_readlane); the lane-pair kernel it replaced stays in the library for the tuning
    build.  Same gradients to fp32 summation-order noise."""
    rs = np.random.RandomState(31)
    xs = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)).to(DEV) for T, R in shapes]
    ws = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)).to(DEV) for T, R in shapes]
    res = {}
    for mode in ("1", "0"):
        kernel_variants.setenv("MMDFN_GRU_KPART_BWD", mode)
        grus = [make_gru(70 + i).to(DEV) for i in range(len(shapes))]
        xg = [x.clone().requires_grad_(True) for x in xs]
        ys = fused.bigru2(xg, grus, 0.0, True)
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        res[mode] = ([x.grad for x in xg], [dict((k, p.grad) for k, p in g.named_parameters()) for g in grus])
    for i in range(len(shapes)):
        assert rel_err(res["1"][0][i], res["0"][0][i]) < 2e-5
        assert float((res["1"][0][i] - res["0"][0][i]).abs().max()) > 0.0      # really a different kernel
        for k in res["1"][1][i]:
            assert rel_err(res["1"][1][i][k], res["0"][1][i][k]) < 5e-5, k


@pytest.mark.parametrize("R", [2, 4])
def test_rows_per_workgroup_variants_agree(R, kernel_variants):
    """The dispatcher gives every sequence its own workgroup up to 2048 sequences; the R = 2 / 4 instantiations (several
    sequences behind each other in one workgroup: only for very many short sequences) are forced here and must give the
    same results and gradients."""
    rs = np.random.RandomState(41)
    shapes = [(17, 9), (33, 70)]
    xs = [torch.from_numpy(rs.randn(T, n, 200).astype(np.float32)).to(DEV) for T, n in shapes]
    ws = [torch.from_numpy(rs.randn(T, n, 200).astype(np.float32)).to(DEV) for T, n in shapes]
    res = {}
    for mode in ("1", str(R)):
        kernel_variants.setenv("MMDFN_GRU_R", mode)
        grus = [make_gru(80 + i).to(DEV) for i in range(len(shapes))]
        xg = [x.clone().requires_grad_(True) for x in xs]
        ys = fused.bigru2(xg, grus, 0.0, True)
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        res[mode] = ([y.detach() for y in ys], [x.grad for x in xg], [dict((k, p.grad) for k, p in g.named_parameters()) for g in grus])
    a, b = res["1"], res[str(R)]
    for i in range(len(shapes)):
        assert abs_err(a[0][i], b[0][i]) < 2e-6
        assert rel_err(a[1][i], b[1][i]) < 2e-5
        for k in a[2][i]:
            assert rel_err(a[2][i][k], b[2][i][k]) < 5e-5, k


@pytest.mark.parametrize("shapes", [[(7, 3)], [(1, 5)], [(33, 40), (33, 100)], [(12, 1), (5, 2), (9, 130)], [(110, 17)]])
def test_mfma_form_agrees_with_the_one_sequence_per_workgroup_kernels(shapes, kernel_variants):
    """Round 5: launches with more than 1 024 sequence-directions (BASELINE cfg3) run 16 sequences per workgroup with the
    recurrent products on bf16-piece MFMAs (csrc/gru_mfma.hip).  Forced here at small sizes (rows that are not multiples of 16,
    one step, several groups, 110 steps) against the scalar kernels: outputs, input gradients and every parameter gradient to
    fp32 summation-order noise -- and really a different kernel."""
    rs = np.random.RandomState(77)
    xs = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)).to(DEV) for T, R in shapes]
    ws = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)).to(DEV) for T, R in shapes]
    res = {}
    for mode in ("0", "100000000"):
        kernel_variants.setenv("MMDFN_GRU_MFMA_MIN", mode)
        grus = [make_gru(90 + i).to(DEV) for i in range(len(shapes))]
        xg = [x.clone().requires_grad_(True) for x in xs]
        ys = fused.bigru2(xg, grus, 0.0, True)
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        res[mode] = ([y.detach() for y in ys], [x.grad for x in xg], [dict((k, p.grad) for k, p in g.named_parameters()) for g in grus])
    a, b = res["0"], res["100000000"]
    differs = False
    for i in range(len(shapes)):
        assert abs_err(a[0][i], b[0][i]) < 2e-6
        assert rel_err(a[1][i], b[1][i]) < 2e-5
        differs = differs or float((a[1][i] - b[1][i]).abs().max()) > 0.0
        for k in a[2][i]:
            assert rel_err(a[2][i][k], b[2][i][k]) < 5e-5, k
    assert differs


def test_matches_torch_gru_module_eval():
    g = make_gru(3)
    x = torch.randn(40, 9, 200)
    with torch.no_grad():
        want = g(x)[0]
        got = fused.bigru2([x.to(DEV)], [g.to(DEV)], 0.0, False)[0]
    assert abs_err(got, want) < 2e-6


@pytest.mark.parametrize("passthrough", [False, True])
@pytest.mark.parametrize("w", [[3.0, 0.0, 1.0], [1.0, 2.0, 0.5], [0.0, 0.0, 2.0]])
@pytest.mark.parametrize("P,lengths", [(2, [15, 9, 1, 6]), (9, [12, 12, 3]), (3, [1]), (2, [110, 64, 80])])
def test_party_gather_combine_kernels_match_index_composition(P, lengths, w, passthrough):
    """K3/K4 kernels vs the torch index-op composition of oracle/mmdfn_vectorised.py (itself checked against the oracle
    on CPU in tests/test_host_logic.py), forward and backward, including a non-one-hot qmask row."""
    from mm_dfn_amd import ops
    from mm_dfn_amd.dialogue_model import _flat_index
    cfg = dict(B=len(lengths), L=max(lengths), P=P, C=6, nlayers=2, D_t=100, D_a=32, D_v=64)
    m = synthetic.build_model(**cfg)
    b = synthetic.make_batch(4, lengths=lengths, **cfg)
    q = b["qmask"].clone()
    if max(lengths) > 2 and P > 1:
        q[1, 0, :2] = 1.0   # two speakers flagged on one utterance: last one wins on scatter
    q = q.to(DEV)
    L, B = max(lengths), len(lengths)
    rs = np.random.RandomState(5)
    Xs = [torch.from_numpy(rs.randn(L, B, 200).astype(np.float32)).to(DEV) for _ in range(3)]
    idx = _flat_index(lengths, L, B, DEV)
    Wg = torch.from_numpy(rs.randn(3, sum(lengths), 200).astype(np.float32)).to(DEV)
    # reference composition (torch index ops)
    Xr = [x.clone().requires_grad_(True) for x in Xs]
    plan = V.party_plan(q)
    Sr = V.party_gather(torch.stack(Xr, 0), plan)
    Er = torch.tanh(Sr * 0.7 + 0.1)      # stand-in for the party GRU (any differentiable row-wise map)
    Ur = V.party_scatter(Er, plan, 3)
    outr = torch.stack([Xr[i] + w[i] * Ur[i] for i in range(3)], 0).reshape(3, L * B, 200).index_select(1, idx)
    (outr * Wg).sum().backward()
    # kernels
    Xk = [x.clone().requires_grad_(True) for x in Xs]
    # the party encoder only sees the modalities with a non-zero weight (E: one column block per such modality)
    act = [i for i in range(3) if w[i] != 0.0]
    bases = list(Xk)
    if passthrough:
        # the gathered modalities return as identities; the combine stage reads those, so both gradient paths of X_m
        # meet inside the gather's backward kernel (no autograd accumulation)
        Sk, rank, *passed = ops.party_gather([Xk[i] for i in act], q, passthrough=True)
        for slot, i in enumerate(act):
            assert passed[slot].data_ptr() == Xk[i].data_ptr()
            bases[i] = passed[slot]
    else:
        Sk, rank = ops.party_gather([Xk[i] for i in act], q)
    Ek = torch.tanh(Sk * 0.7 + 0.1)
    outk = ops.party_combine(bases, Ek, rank, idx, w)
    (outk * Wg).sum().backward()
    BP = B * P
    for slot, i in enumerate(act):
        assert abs_err(Sk[:, slot * BP:(slot + 1) * BP], Sr[:, i * BP:(i + 1) * BP]) == 0.0
    assert abs_err(outk, outr) < 1e-6
    for i in range(3):
        assert rel_err(Xk[i].grad, Xr[i].grad) < 1e-6


@pytest.mark.parametrize("stacked_views", [True, False])
@pytest.mark.parametrize("P,lengths,nact", [(2, [15, 9, 1, 6], 2), (9, [12, 12, 3], 3), (3, [1], 1), (2, [110, 64, 80], 2)])
def test_project_gather_node_matches_projection_of_gathered_rows(P, lengths, nact, stacked_views):
    """The first party-GRU layer as the reference computes it -- gather the party rows (padding = zeros), then
    X [W_ih; W_ih_reverse]^T + b on EVERY row (model.py:1076-1082) -- against ops.project_gather (projection of the
    utterances, gather, bias on every row): gate pre-activations, the gradients of both weight blocks and both biases (the
    bias gradient is the column sum over ALL party rows, padding included), and the input gradients including the part
    that arrives through the passthrough aliases."""
    from mm_dfn_amd import ops
    cfg = dict(B=len(lengths), L=max(lengths), P=P, C=6, nlayers=2, D_t=100, D_a=32, D_v=64)
    q = synthetic.make_batch(4, lengths=lengths, **cfg)["qmask"].to(DEV)
    L, B = max(lengths), len(lengths)
    rs = np.random.RandomState(9)
    t = lambda *sh: torch.from_numpy(rs.randn(*sh).astype(np.float32)).to(DEV)
    Xs = [t(L, B, 200) for _ in range(nact)]
    wbuf, bbuf = t(600, 200) * 0.1, t(600)
    Wg, Wp = t(L, nact * B * P, 600), [t(L, B, 200) for _ in range(nact)]
    # reference composition: index-op gather, then the projection over every party row
    Xr = [x.clone().requires_grad_(True) for x in Xs]
    w1r, w2r = wbuf[:300].clone().requires_grad_(True), wbuf[300:].clone().requires_grad_(True)
    b1r, b2r = bbuf[:300].clone().requires_grad_(True), bbuf[300:].clone().requires_grad_(True)
    Sr = V.party_gather(torch.stack(Xr, 0), V.party_plan(q))
    gr = torch.nn.functional.linear(Sr, torch.cat([w1r, w2r]), torch.cat([b1r, b2r]))
    ((gr * Wg).sum() + sum((x * w).sum() for x, w in zip(Xr, Wp))).backward()
    # the node
    Xk = [x.clone().requires_grad_(True) for x in Xs]
    wk, bk = wbuf.clone(), bbuf.clone()
    w1, w2 = wk[:300].requires_grad_(True), wk[300:].requires_grad_(True)
    b1, b2 = bk[:300].requires_grad_(True), bk[300:].requires_grad_(True)
    views = (wk, bk) if stacked_views else (None, None)
    gk, rank, *passed = ops.project_gather(Xk, q, w1, w2, b1, b2, *views)
    assert all(p_.data_ptr() == x.data_ptr() for p_, x in zip(passed, Xk))
    ((gk * Wg).sum() + sum((x * w).sum() for x, w in zip(passed, Wp))).backward()
    assert abs_err(gk, gr) < 2e-5
    for a, b in ((w1, w1r), (w2, w2r), (b1, b1r), (b2, b2r)):
        assert rel_err(a.grad, b.grad) < 2e-5
    for a, b in zip(Xk, Xr):
        assert rel_err(a.grad, b.grad) < 2e-5


@pytest.mark.parametrize("stacked_views", [True, False])
@pytest.mark.parametrize("P,lengths,nact,src", [(2, [15, 9, 1, 6], 2, 1), (2, [110, 64, 80], 3, 2), (3, [1], 1, 0)])
def test_project_gather_riders(P, lengths, nact, src, stacked_views):
    """riders: another projection of one of the gathered inputs (the context GRU's first-layer input contraction, model.py:1132)
    computed in the node's grouped launch -- against the node without riders plus a linear2 of the same input: the rider's
    output, every weight / bias gradient, and the source input's gradient, which now meets INSIDE the node."""
    from mm_dfn_amd import ops
    from mm_dfn_amd import train as T
    cfg = dict(B=len(lengths), L=max(lengths), P=P, C=6, nlayers=2, D_t=100, D_a=32, D_v=64)
    q = synthetic.make_batch(4, lengths=lengths, **cfg)["qmask"].to(DEV)
    L, B = max(lengths), len(lengths)
    rs = np.random.RandomState(19)
    t = lambda *sh: torch.from_numpy(rs.randn(*sh).astype(np.float32)).to(DEV)
    Xs = [t(L, B, 200) for _ in range(nact)]
    wbuf, bbuf, rwbuf, rbbuf = t(600, 200) * 0.1, t(600), t(600, 200) * 0.1, t(600)
    Wg, Wp, Wr = t(L, nact * B * P, 600), [t(L, B, 200) for _ in range(nact)], t(L, B, 600)

    def run(with_rider):
        Xk = [x.clone().requires_grad_(True) for x in Xs]
        wk, bk, rwk, rbk = wbuf.clone(), bbuf.clone(), rwbuf.clone(), rbbuf.clone()
        prm = [wk[:300].requires_grad_(True), wk[300:].requires_grad_(True), bk[:300].requires_grad_(True), bk[300:].requires_grad_(True)]
        rprm = [rwk[:300].requires_grad_(True), rwk[300:].requires_grad_(True), rbk[:300].requires_grad_(True), rbk[300:].requires_grad_(True)]
        views = (wk, bk) if stacked_views else (None, None)
        if with_rider:
            gk, rank, *rest = ops.project_gather(Xk, q, *prm, *views, riders=[(src, *rprm, rwk if stacked_views else None)])
            passed, (yr,) = rest[:nact], rest[nact:]
        else:
            gk, rank, *passed = ops.project_gather(Xk, q, *prm, *views)
            yr = ops.linear2(passed[src], *rprm, rwk if stacked_views else None, None)
        loss = (gk * Wg).sum() + (yr * Wr).sum() + sum((x * w).sum() for i, (x, w) in enumerate(zip(passed, Wp)) if i != src)
        T.backward(loss)
        return gk.detach(), yr.detach(), [x.grad for x in Xk], [p_.grad for p_ in prm + rprm]

    g1, y1, dx1, dp1 = run(True)
    g0, y0, dx0, dp0 = run(False)
    assert torch.equal(g1, g0)
    assert rel_err(y1, y0) < 2e-6
    for a, b in zip(dx1 + dp1, dx0 + dp0):
        assert rel_err(a, b) < 3e-6


@pytest.mark.parametrize("R,H", [(7040, 600), (33, 600), (1, 4), (19008, 600), (100, 68), (64, 64)])
def test_column_sum_kernel(R, H):
    from mm_dfn_amd import ops
    A = torch.from_numpy(np.random.RandomState(R + H).randn(R, H).astype(np.float32)).to(DEV)
    got = ops.colsum(A)
    assert torch.equal(got, ops.colsum(A))                        # fixed summation order
    want = A.double().sum(0)
    assert float((got.double() - want).abs().max()) < 1e-5 * max(1.0, float(A.abs().sum(0).max()))
    part = ops.colsum(A[:, : H // 2 // 4 * 4 or 4])               # a column slice: row stride > width
    assert float((part.double() - want[: part.numel()]).abs().max()) < 1e-5 * max(1.0, float(A.abs().sum(0).max()))


@pytest.mark.parametrize("shapes", [[(9, 3, 200)], [(110, 16, 200), (110, 64, 200)], [(1, 1, 4), (3, 5, 8), (2, 2, 12), (7, 1, 4)]])
def test_multi_tensor_mask_scale_is_dropout_on_given_flags(shapes):
    """nn.GRU(dropout=p) between the layers (model.py:866,868) for every encoder group in one launch each way:
    out = x * flags / (1-p), exactly torch's multiply on the same keep flags, and so is its backward."""
    from mm_dfn_amd import ops
    rs = np.random.RandomState(17 + len(shapes))
    p = 0.5
    xs = [torch.from_numpy(rs.randn(*sh).astype(np.float32)).to(DEV).requires_grad_(True) for sh in shapes]
    masks = [torch.from_numpy((rs.uniform(size=int(np.prod(sh))) > p).astype(np.float32)).to(DEV) for sh in shapes]
    ws = [torch.from_numpy(rs.randn(*sh).astype(np.float32)).to(DEV) for sh in shapes]
    got = ops.mask_scale(xs, masks, 1.0 / (1.0 - p))
    sum((y * w).sum() for y, w in zip(got, ws)).backward()
    for x, m, w, y in zip(xs, masks, ws, got):
        assert torch.equal(y, x.detach() * m.view_as(x) * 2.0)
        assert torch.equal(x.grad, w * m.view_as(x) * 2.0)


def test_train_mode_dropout_through_bigru2_is_unbiased_and_reaches_the_gradients():
    """bigru2(training=True, dropout=p): keep flags from the step's pool, no torch dropout op in the graph."""
    from mm_dfn_amd import ops
    g = make_gru(8).to(DEV)
    x = torch.randn(30, 24, 200, device=DEV, requires_grad=True)
    with ops.flag_pool("t"):
        y_tr = fused.bigru2([x], [g], 0.5, True)[0]
    with torch.no_grad():
        y_ev = fused.bigru2([x.detach()], [g], 0.5, False)[0]
    assert float((y_tr - y_ev).abs().max()) > 1e-3              # the second layer saw a dropped-out input
    node_names = set()
    stack = [y_tr.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in node_names:
            continue
        node_names.add(fn)
        stack += [f for f, _ in fn.next_functions]
    assert not any("Dropout" in type(fn).__name__ for fn in node_names)
    y_tr.sum().backward()
    assert bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0
    for p in g.parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all())


def test_direction_weight_pairs_become_one_stacked_operand_without_a_copy():
    """[W_ih; W_ih_reverse] for the input-gradient GEMM: the first training forward moves each pair into one buffer
    (values kept, .data re-pointed); afterwards the stacked operand is a view -- no per-step packing launch -- and the
    gradients equal those of the copying path (parameters an optimizer has laid out itself)."""
    g = make_gru(9).to(DEV)
    before = {k: v.detach().clone() for k, v in g.state_dict().items()}
    x = torch.randn(12, 5, 200, device=DEV, requires_grad=True)
    w = torch.randn(12, 5, 200, device=DEV)
    (fused.bigru2([x], [g], 0.0, True)[0] * w).sum().backward()
    for layer in range(2):
        wf, wr = getattr(g, "weight_ih_l%d" % layer), getattr(g, "weight_ih_l%d_reverse" % layer)
        assert fused._adjacent(wf, wr)
        sv = fused._stacked_view(wf, wr)
        assert sv.data_ptr() == wf.data_ptr() and torch.equal(sv, torch.cat([wf, wr], 0))
    for k, v in g.state_dict().items():
        assert torch.equal(v, before[k])
    # an in-place update (what an optimizer does) is seen through the view
    with torch.no_grad():
        g.weight_ih_l0_reverse.add_(1.0)
    assert torch.equal(fused._stacked_view(g.weight_ih_l0, g.weight_ih_l0_reverse)[300:], g.weight_ih_l0_reverse)
    with torch.no_grad():
        g.weight_ih_l0_reverse.sub_(1.0)
    grads = {k: p.grad.clone() for k, p in g.named_parameters()}
    xg = x.grad.clone()
    # copying path: parameters flagged as laid out by someone else and NOT adjacent
    g2 = make_gru(9).to(DEV)
    for p in g2.parameters():
        p._mmdfn_flat = True
    x2 = x.detach().clone().requires_grad_(True)
    (fused.bigru2([x2], [g2], 0.0, True)[0] * w).sum().backward()
    assert not fused._adjacent(g2.weight_ih_l0, g2.weight_ih_l0_reverse)
    assert rel_err(x2.grad, xg) < 1e-6
    for k, p in g2.named_parameters():
        assert rel_err(p.grad, grads[k]) < 1e-6, k


def test_keep_flag_generator_statistics_seeding_and_graph_replay():
    """ops.draw_flags (csrc/encoder_glue.hip, Philox4x32-10 with device-resident state): the keep rate, independence of
    consecutive draws, torch.manual_seed reproducibility (also with another torch random op in between), the degenerate rates,
    and fresh flags at every replay of a captured graph."""
    from mm_dfn_amd import ops
    n = 1 << 20
    torch.manual_seed(1234)
    a = ops.draw_flags(n, 0.3, DEV)
    b = ops.draw_flags(n, 0.3, DEV)
    assert set(torch.unique(a).tolist()) == {0.0, 1.0}
    assert abs(float(a.mean()) - 0.7) < 3e-3 and abs(float(b.mean()) - 0.7) < 3e-3
    assert abs(float((a * b).mean()) - 0.49) < 3e-3                     # consecutive draws are independent
    assert abs(float((a[:-1] * a[1:]).mean()) - 0.49) < 3e-3           # and so are neighbours
    torch.manual_seed(1234)
    a2 = ops.draw_flags(n, 0.3, DEV)
    r1 = torch.rand(5, device=DEV)
    b2 = ops.draw_flags(n, 0.3, DEV)
    assert torch.equal(a, a2) and not torch.equal(b, b2)               # the torch.rand in between moved the stream
    torch.manual_seed(1234)
    ops.draw_flags(n, 0.3, DEV)
    assert torch.equal(torch.rand(5, device=DEV), r1)
    assert torch.equal(ops.draw_flags(n, 0.3, DEV), b2)
    assert float(ops.draw_flags(64, 0.0, DEV).min()) == 1.0 and float(ops.draw_flags(64, 1.0, DEV).max()) == 0.0
    # captured: the state advances on the device, replay after replay
    out = torch.empty(4096, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.draw_flags(4096, 0.5, DEV)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out.copy_(ops.draw_flags(4096, 0.5, DEV))
    torch.cuda.current_stream().wait_stream(s)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append(out.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    assert all(abs(float(x.mean()) - 0.5) < 0.05 for x in seen)


def _party_case(L, lengths, P, seed, silent=None, solo=None):
    """Speaker masks of a ragged batch: one-hot per utterance, zero beyond each dialogue's length; ``silent`` = (b, p): that
    speaker never talks in dialogue b; ``solo`` = b: one speaker holds the whole dialogue (k_bp = L_b)."""
    rs = np.random.RandomState(seed)
    B = len(lengths)
    q = np.zeros((L, B, P), np.float32)
    for b, n in enumerate(lengths):
        spk = rs.randint(0, P, size=n)
        if silent is not None and silent[0] == b:
            spk = np.where(spk == silent[1], (silent[1] + 1) % P, spk)
        if solo is not None and solo == b:
            spk[:] = P - 1
        q[np.arange(n), b, spk] = 1.0
    return torch.from_numpy(q)


def _run_party(mode, L, lengths, P, seed, qmask, dropout=0.0, batch=True, n_mod=2, table=True, l1seg=True):
    """ctx + party encoders through bigru2 with / without the valid-length launches; loss reads only what the model reads:
    the scattered rows [:k_bp] of the party output and the context rows of real utterances."""
    from mm_dfn_amd import ops
    from mm_dfn_amd import train as T
    rs = np.random.RandomState(seed)
    B = len(lengths)
    torch.manual_seed(seed)
    g_ctx, g_par = make_gru(seed + 1).to(DEV), make_gru(seed + 2).to(DEV)
    Xs = [torch.from_numpy(rs.randn(L, B, 200).astype(np.float32)).to(DEV).requires_grad_(True) for _ in range(n_mod)]
    Xc = torch.from_numpy(rs.randn(L, B, 200).astype(np.float32)).to(DEV).requires_grad_(True)
    q = qmask.to(DEV)
    prev, prev_l1 = fused.TRUNCATE, fused.L1_SKIPS_SILENT
    fused.TRUNCATE, fused.L1_SKIPS_SILENT = mode, l1seg
    try:
        tab = fused.start_party_table(g_par, L) if (mode and table) else None
        S, rank = ops.party_gather(Xs, q)
        with ops.flag_pool(("t", seed, bool(mode))):
            ctx, E = fused.bigru2([Xc, S], [g_ctx, g_par], dropout, True, party=(1, rank, tab) if mode else None)
        # weights: zero where the model never looks (t >= k of a party row)
        k = (rank.max(0).values + 1).to(torch.int64)                     # (B, P)
        kk = k.reshape(1, B * P).repeat(1, n_mod)                        # rows (m, b, p)
        live = (torch.arange(L, device=DEV).unsqueeze(1) < kk).unsqueeze(-1).to(torch.float32)
        wE = torch.from_numpy(np.random.RandomState(seed + 9).randn(*E.shape).astype(np.float32)).to(DEV) * live
        wc = torch.from_numpy(np.random.RandomState(seed + 8).randn(*ctx.shape).astype(np.float32)).to(DEV)
        loss = (E * wE).sum() + (ctx * wc).sum()
        if batch:
            T.backward(loss)
        else:
            loss.backward()
    finally:
        fused.TRUNCATE, fused.L1_SKIPS_SILENT = prev, prev_l1
    torch.cuda.synchronize()
    grads = {("ctx." if g is g_ctx else "par.") + n: p.grad.clone() for g in (g_ctx, g_par) for n, p in g.named_parameters()}
    return dict(E=E.detach() * live, ctx=ctx.detach(), dX=[x.grad.clone() for x in Xs], dXc=Xc.grad.clone(), grads=grads)


@pytest.mark.parametrize("table", ["table", "l1seg", "l1plain"])
@pytest.mark.parametrize("batch", [True, False])
@pytest.mark.parametrize("L,lengths,P,kw", [
    (15, [15, 9, 1], 3, {}),                       # the goldens' (L, B, P) sets
    (33, [33, 20, 7, 3], 9, {}),
    (110, [110, 64], 2, {}),
    (24, [24, 24, 11, 5, 17], 4, dict(silent=(1, 2), solo=0)),       # a silent speaker; one speaker holding a whole dialogue (k = L)
    (12, [12] * 40, 2, {}),                        # more chains than CUs in the plain form
])
def test_valid_length_launches_match_full_length(L, lengths, P, kw, batch, table):
    """Layer 1 reverse truncated against the all-padding sequence, layer 2 forward truncated, silent rows skipped, the P
    sequences of a (modality, dialogue) back to back in one workgroup: the scattered rows and the context are BIT-equal to
    the full-length launches, every gradient agrees to summation-order noise (the padding steps' contributions to the
    recurrent weights and biases arrive through the one all-padding sequence instead of once per row)."""
    q = _party_case(L, lengths, P, 7 + L, **kw)
    full = _run_party(False, L, lengths, P, 11, q, batch=batch)
    seg = _run_party(True, L, lengths, P, 11, q, batch=batch, table=table == "table", l1seg=table != "l1plain")
    assert torch.equal(seg["ctx"], full["ctx"])
    assert torch.equal(seg["E"], full["E"])
    assert float(full["E"].abs().max()) > 0
    for a, b in zip(seg["dX"] + [seg["dXc"]], full["dX"] + [full["dXc"]]):
        assert rel_err(a, b) < 2e-5
    for k in full["grads"]:
        assert rel_err(seg["grads"][k], full["grads"][k]) < 5e-5, k


def test_valid_length_launches_with_interlayer_dropout():
    """nn.GRU's dropout between the layers is applied to every row, visited or not: same flags, same result."""
    L, lengths, P = 21, [21, 13, 8], 3
    q = _party_case(L, lengths, P, 3)
    full = _run_party(False, L, lengths, P, 5, q, dropout=0.5)
    seg = _run_party(True, L, lengths, P, 5, q, dropout=0.5)
    assert torch.equal(seg["E"], full["E"]) and torch.equal(seg["ctx"], full["ctx"])
    for k in full["grads"]:
        assert rel_err(seg["grads"][k], full["grads"][k]) < 5e-5, k


@pytest.mark.parametrize("L,lengths,P,kw", [
    (15, [15, 9, 1], 3, {}),
    (33, [33, 20, 7, 3], 9, {}),
    (24, [24, 24, 11, 5, 17], 4, dict(silent=(1, 2), solo=0)),
])
@pytest.mark.parametrize("io,kpart", [("0", "0"), ("1", "1")])
def test_valid_length_launches_on_both_kernel_families(L, lengths, P, kw, io, kpart, kernel_variants):
    """The segmented launch picks the 5-wave / wave-partitioned kernels up to one chain per CU and the 4-wave lane-pair
    kernels beyond; both families are forced here on the same small cases."""
    kernel_variants.setenv("MMDFN_GRU_IO", io)
    kernel_variants.setenv("MMDFN_GRU_KPART_BWD", kpart)
    q = _party_case(L, lengths, P, 7 + L, **kw)
    seg = _run_party(True, L, lengths, P, 11, q)
    kernel_variants.delenv("MMDFN_GRU_IO")
    kernel_variants.delenv("MMDFN_GRU_KPART_BWD")
    full = _run_party(False, L, lengths, P, 11, q)
    assert torch.equal(seg["ctx"], full["ctx"]) and torch.equal(seg["E"], full["E"])
    for a, b in zip(seg["dX"] + [seg["dXc"]], full["dX"] + [full["dXc"]]):
        assert rel_err(a, b) < 2e-5
    for k in full["grads"]:
        assert rel_err(seg["grads"][k], full["grads"][k]) < 5e-5, k
