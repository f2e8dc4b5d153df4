"""GPU parity of the graph kernels (K5 adjacency, K6 propagate) against the CPU oracle.

Tolerances: forward 1e-5 relative (fp32 summation-order noise), gradients 1e-4
relative; the adjacency itself 2e-5 absolute because acos is
ill-conditioned near the unit diagonal (d/dx ~ 224 at x = 0.99999).
"""
import numpy as np
import pytest
import torch

import mmdfn_oracle as O
from mm_dfn_amd import GCNII_lyc, ops, synthetic
from mm_dfn_amd.layout import BlockTileAdjacency, DialogueLayout, pair_list
from util import abs_err, random_block_adjacency, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [
    ([5], 3, 100),
    ([7, 3, 1], 3, 100),
    ([20, 13], 3, 100),
    ([110, 64, 65, 27], 3, 100),
    ([33, 3, 17, 48, 16], 2, 100),
    ([130, 40], 3, 200),
    ([70], 6, 512),
    ([9, 31], 1, 36),
]


@pytest.mark.parametrize("lengths,M,d", CASES)
def test_propagate_forward_and_transpose(lengths, M, d):
    adj, dense, _, _ = random_block_adjacency(11, lengths, M, DEV)
    rs = np.random.RandomState(5)
    H = torch.from_numpy(rs.randn(M * sum(lengths), d).astype(np.float32))
    out = ops.propagate_raw(adj.tiles, adj.cross, H.to(DEV), adj.layout)
    assert rel_err(out, dense @ H) < 1e-5
    # dense built by to_dense() must agree with the independent CPU construction
    assert abs_err(adj.to_dense(), dense) == 0.0
    out_t = ops.propagate_raw(adj.tiles, adj.cross, H.to(DEV), adj.layout, transpose=True)
    assert rel_err(out_t, dense.t() @ H) < 1e-5


SPLIT_CASES = [
    ([5], 3, 100),
    ([7, 3, 1], 3, 100),
    ([32, 33, 31, 64], 3, 100),
    ([110, 64, 65, 27], 3, 100),
    ([129, 127, 128, 200], 2, 128),
    ([260, 40], 6, 64),
    ([513], 3, 100),
    ([9, 31], 1, 36),
    ([140, 77], 3, 200),
    ([70], 6, 512),
    ([33, 200], 2, 132),
    # 96 < d <= 112: three 32-column tiles + the 16-column tail tile (round 5), every width of that range
    ([130, 45], 3, 104),
    ([64, 300], 6, 108),
    ([200], 2, 112),
]


@pytest.mark.parametrize("lengths,M,d", SPLIT_CASES)
def test_propagate_bf16_piece_kernel(lengths, M, d, kernel_variants):
    """The large-launch variant (three exact bf16 pieces per operand, six MFMA products) forced on every
    shape: same fp32-level tolerance as the f32-MFMA kernel, ragged chunk tails, and tile padding columns
    poisoned with NaN (they are not data and must never reach a product)."""
    adj, dense, _, _ = random_block_adjacency(13, lengths, M, DEV)
    lay = adj.layout
    tiles = adj.tiles.clone()
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        if ld > L:
            tiles[base: base + M * L * ld].view(M * L, ld)[:, L:] = float("nan")
    rs = np.random.RandomState(7)
    H = torch.from_numpy(rs.randn(M * sum(lengths), d).astype(np.float32))
    kernel_variants.setenv("MMDFN_PROP_CFG", "8")
    out = ops.propagate_raw(tiles, adj.cross, H.to(DEV), lay)
    want = dense.double() @ H.double()
    assert rel_err(out, want) < 1e-5
    kernel_variants.setenv("MMDFN_PROP_CFG", "9")
    ref = ops.propagate_raw(tiles, adj.cross, H.to(DEV), lay)
    # both kernels sit at fp32 rounding level against the fp64 product
    e_split = float((out.double().cpu() - want).abs().max())
    e_f32 = float((ref.double().cpu() - want).abs().max())
    assert e_split <= 4 * e_f32 + 1e-7




def test_propagate_bf16_piece_kernel_is_the_large_launch_default(kernel_variants):
    """At the long-dialogue stress shape (L=512, M=6) the dispatcher picks the bf16-piece kernel; results agree
    with the f32-MFMA kernel to fp32 rounding."""
    kernel_variants.delenv("MMDFN_PROP_CFG", raising=False)
    lengths = [512] * 11 + [300]
    M, d = 6, 100
    rs = np.random.RandomState(8)
    N = sum(lengths)
    adj = ops.build_adjacency(torch.from_numpy(rs.randn(M, N, 200).astype(np.float32)).to(DEV), lengths)
    H = torch.from_numpy(rs.randn(M * N, d).astype(np.float32)).to(DEV)
    out = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    kernel_variants.setenv("MMDFN_PROP_CFG", "9")
    ref = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    assert float((out - ref).abs().max()) < 2e-6
    assert float((out - ref).abs().max()) > 0.0   # it really was a different kernel


@pytest.mark.parametrize("lengths,M,d", [([513, 140], 3, 100), ([77, 129], 6, 112)])
def test_propagate_tail_tile_against_four_full_tiles(lengths, M, d, kernel_variants):
    """The tail-tile form of the bf16-piece kernel (columns 96 .. on v_mfma_f32_16x16x32_bf16 with the A pieces regrouped by
    v_permlane16_swap) against the four-tile form it replaces for 96 < d <= 112: same products, fp32 summation-order noise only."""
    adj, dense, _, _ = random_block_adjacency(17, lengths, M, DEV)
    rs = np.random.RandomState(9)
    H = torch.from_numpy(rs.randn(M * sum(lengths), d).astype(np.float32)).to(DEV)
    kernel_variants.setenv("MMDFN_PROP_CFG", "8")
    outs = {}
    for tail in ("1", "0"):
        kernel_variants.setenv("MMDFN_SPLIT_TAIL", tail)
        outs[tail] = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    want = dense.double() @ H.double().cpu()
    assert rel_err(outs["1"], want) < 1e-5
    assert rel_err(outs["1"], outs["0"]) < 1e-6
    assert float((outs["1"][:, 96:] - outs["0"][:, 96:]).abs().max()) > 0.0   # (really a different instruction stream there)
    assert bool((outs["1"][:, :96] == outs["0"][:, :96]).all())        # the three full tiles are the same instructions


@pytest.mark.parametrize("lengths,M,d", CASES)
def test_propagate_backward(lengths, M, d):
    adj, dense, tiles, cross = random_block_adjacency(12, lengths, M, DEV)
    rs = np.random.RandomState(6)
    N = sum(lengths)
    H = torch.from_numpy(rs.randn(M * N, d).astype(np.float32))
    R = torch.from_numpy(rs.randn(M * N, d).astype(np.float32))
    # oracle: dense autograd, gradient read back on the stored pattern
    Ad = dense.clone().requires_grad_(True)
    Hd = H.clone().requires_grad_(True)
    ((Ad @ Hd) * R).sum().backward()
    # HIP
    t = adj.tiles.clone().requires_grad_(True)
    c = adj.cross.clone().requires_grad_(True)
    Hg = H.to(DEV).requires_grad_(True)
    out = ops._Propagate.apply(t, c, Hg, adj.layout, False)
    (out * R.to(DEV)).sum().backward()
    assert rel_err(Hg.grad, Hd.grad) < 1e-5
    gA = BlockTileAdjacency(adj.layout, t.grad, c.grad)
    lay = adj.layout
    start = 0
    worst = 0.0
    scale = float(Ad.grad.abs().max())
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        for m in range(M):
            got = t.grad[base + m * L * ld: base + (m + 1) * L * ld].view(L, ld).cpu()
            want = Ad.grad[m * N + start:m * N + start + L, m * N + start:m * N + start + L]
            worst = max(worst, float((got[:, :L] - want).abs().max()))
            assert float(got[:, L:].abs().max()) == 0.0 if ld > L else True
        start += L
    ar = torch.arange(N)
    for k, (m, n) in enumerate(pair_list(M)):
        want = Ad.grad[m * N + ar, n * N + ar] + Ad.grad[n * N + ar, m * N + ar]
        worst = max(worst, float((c.grad[k].cpu() - want).abs().max()))
    assert worst / scale < 1e-5


ADJ_CASES = [([5], 3, 200), ([7, 3, 1], 3, 200), ([20, 13], 3, 200), ([110, 64, 33], 3, 200), ([40, 9], 2, 200),
             ([24, 50], 6, 64)]


@pytest.mark.parametrize("lengths,M,D", ADJ_CASES)
def test_adjacency_build_forward(lengths, M, D):
    rs = np.random.RandomState(21)
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(M, N, D).astype(np.float32))
    adj = ops.build_adjacency(feats.to(DEV), lengths, 1.0)
    want = O.create_big_adj([feats[m] for m in range(M)], lengths, 1.0)
    assert abs_err(adj.to_dense(), want) < 2e-5
    t, c, _ = O.adjacency_tiles([feats[m] for m in range(M)], lengths, 1.0)
    assert abs_err(adj.cross, c) < 2e-5
    # symmetric by construction
    dense = adj.to_dense()
    assert abs_err(dense, dense.t()) < 1e-6


@pytest.mark.parametrize("lengths,M,D", ADJ_CASES)
@pytest.mark.parametrize("modal_weight", [1.0, 0.7])
def test_adjacency_build_backward(lengths, M, D, modal_weight):
    rs = np.random.RandomState(22)
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(M, N, D).astype(np.float32))
    R = torch.from_numpy(rs.randn(M * N, M * N).astype(np.float32))
    fo = feats.clone().requires_grad_(True)
    (O.create_big_adj([fo[m] for m in range(M)], lengths, modal_weight) * R).sum().backward()
    fg = feats.to(DEV).requires_grad_(True)
    adj = ops.build_adjacency(fg, lengths, modal_weight)
    (adj.to_dense() * R.to(DEV)).sum().backward()
    assert rel_err(fg.grad, fo.grad) < 1e-4


def test_adjacency_then_propagate_chain_gradient():
    """d/dfeats of sum((A(feats) . H) * R): exercises dA from tile_outer feeding adj_build_bwd."""
    lengths, M, D, d = [30, 17, 8], 3, 200, 100
    rs = np.random.RandomState(23)
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(M, N, D).astype(np.float32))
    H = torch.from_numpy(rs.randn(M * N, d).astype(np.float32))
    R = torch.from_numpy(rs.randn(M * N, d).astype(np.float32))
    fo = feats.clone().requires_grad_(True)
    Ho = H.clone().requires_grad_(True)
    ((O.create_big_adj([fo[m] for m in range(M)], lengths) @ Ho) * R).sum().backward()
    fg = feats.to(DEV).requires_grad_(True)
    Hg = H.to(DEV).requires_grad_(True)
    (ops.propagate(ops.build_adjacency(fg, lengths), Hg) * R.to(DEV)).sum().backward()
    # dH = A^T dO inherits the 2e-5 absolute acos noise of A itself
    assert rel_err(Hg.grad, Ho.grad) < 1e-4
    assert rel_err(fg.grad, fo.grad) < 1e-4


def test_strided_propagate_and_tile_outer():
    """Row-strided operands (column slices of a wider matrix) through the C ABI."""
    lengths, M, d = [40, 17], 3, 100
    adj, dense, _, _ = random_block_adjacency(31, lengths, M, DEV)
    rs = np.random.RandomState(32)
    N = sum(lengths)
    wide = torch.from_numpy(rs.randn(M * N, 3 * d).astype(np.float32)).to(DEV)
    H = wide[:, d:2 * d]
    out_wide = torch.zeros(M * N, 2 * d, device=DEV)
    ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=out_wide[:, :d])
    assert rel_err(out_wide[:, :d], dense @ H.cpu()) < 1e-5
    assert float(out_wide[:, d:].abs().max()) == 0.0
    X = wide[:, 2 * d:]
    dt, dc = ops.tile_outer_raw(X, H, adj.layout)
    dt2, dc2 = ops.tile_outer_raw(X.contiguous(), H.contiguous(), adj.layout)
    assert abs_err(dt, dt2) == 0.0 and abs_err(dc, dc2) == 0.0


def test_lstm_pointwise_and_gcnii_combine_against_oracle():
    rs = np.random.RandomState(33)
    R, Hd = 333, 100
    G = torch.from_numpy(rs.randn(R, 4 * Hd).astype(np.float32))
    c0 = torch.from_numpy(rs.randn(R, Hd).astype(np.float32))
    wh = torch.from_numpy(rs.randn(R, Hd).astype(np.float32))
    wc = torch.from_numpy(rs.randn(R, Hd).astype(np.float32))
    for prev in (True, False):
        Go = G.clone().requires_grad_(True)
        co = c0.clone().requires_grad_(True)
        z = torch.zeros(R, Hd)
        # oracle cell with identity "weights": feed G through lstm_cell by making x W + h W = G
        i, f, g, o = Go.chunk(4, 1)
        cprev = co if prev else z
        c2 = torch.sigmoid(f) * cprev + torch.sigmoid(i) * torch.tanh(g)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        ((h2 * wh).sum() + (c2 * wc).sum()).backward()
        Gg = G.to(DEV).requires_grad_(True)
        cg = c0.to(DEV).requires_grad_(True)
        h, c = ops.lstm_pointwise(Gg, cg if prev else None)
        ((h * wh.to(DEV)).sum() + (c * wc.to(DEV)).sum()).backward()
        assert abs_err(h, h2) < 1e-6 and abs_err(c, c2) < 1e-6
        assert rel_err(Gg.grad, Go.grad) < 1e-5
        if prev:
            assert rel_err(cg.grad, co.grad) < 1e-5
    d = 100
    P = torch.from_numpy(rs.randn(R, d).astype(np.float32))
    S2 = torch.from_numpy(rs.randn(R, 2 * d).astype(np.float32))
    q = torch.from_numpy(rs.randn(R, d).astype(np.float32))
    mask = torch.from_numpy((rs.rand(R, d) > 0.5).astype(np.float32) * 2.0)
    w = torch.from_numpy(rs.randn(R, d).astype(np.float32))
    theta, alpha = 0.405, 0.2
    for use_q, use_m in ((True, True), (False, False), (True, False)):
        Po, So, qo = P.clone().requires_grad_(True), S2.clone().requires_grad_(True), q.clone().requires_grad_(True)
        hi, h0 = So[:, :d], So[:, d:]
        ref = torch.relu(theta * Po + (1 - theta) * ((1 - alpha) * hi + alpha * h0))
        if use_m:
            ref = ref * mask
        if use_q:
            ref = ref + qo
        (ref * w).sum().backward()
        Pg, Sg, qg = (t.to(DEV).requires_grad_(True) for t in (P, S2, q))
        out = ops.gcnii_combine(Pg, Sg, qg if use_q else None, mask.to(DEV) if use_m else None, theta, alpha)
        (out * w.to(DEV)).sum().backward()
        assert abs_err(out, ref) < 1e-6
        assert rel_err(Pg.grad, Po.grad) < 1e-6 and rel_err(Sg.grad, So.grad) < 1e-6
        if use_q:
            assert rel_err(qg.grad, qo.grad) < 1e-6


@pytest.mark.parametrize("R,K,N", [(5280, 200, 100), (4100, 100, 400), (10560, 200, 600), (37, 12, 5), (1000, 900, 6),
                                   (4097, 256, 67)])
def test_mfma_linear_forward_backward(R, K, N):
    rs = np.random.RandomState(R + K + N)
    x = torch.from_numpy(rs.randn(R, K).astype(np.float32))
    w = torch.from_numpy((rs.randn(N, K) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rs.randn(N).astype(np.float32))
    g = torch.from_numpy(rs.randn(R, N).astype(np.float32))
    for act in (0, 1):
        xo, wo, bo = (t.clone().double().requires_grad_(True) for t in (x, w, b))
        yo = torch.nn.functional.linear(xo, wo, bo)
        if act:
            yo = torch.relu(yo)
        (yo * g.double()).sum().backward()
        xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        y = ops.linear(xg, wg, bg, act)      # MFMA kernels for the big-row shapes, library GEMM for the rest;
        # dW / db always through the split-K kernel when the shape allows
        (y * g.to(DEV)).sum().backward()
        assert rel_err(y, yo) < 2e-6
        assert rel_err(xg.grad, xo.grad) < 2e-5 and rel_err(wg.grad, wo.grad) < 2e-5 and rel_err(bg.grad, bo.grad) < 2e-5
    # base + x W^T + b through the autograd wrapper (LSTM gate pre-activations: G + h W_hh^T)
    base = torch.from_numpy(rs.randn(R, N).astype(np.float32))
    xg, wg, baseg = (t.to(DEV).requires_grad_(True) for t in (x, w, base))
    y = ops.linear(xg, wg, None, 0, base=baseg)
    (y * g.to(DEV)).sum().backward()
    assert rel_err(y, base.double() + x.double() @ w.double().t()) < 2e-6
    assert rel_err(baseg.grad, g) < 1e-7 and rel_err(xg.grad, g.double() @ w.double()) < 2e-5
    raw = ops.linear_raw(x.to(DEV), w.to(DEV), None, 0)
    acc = ops.linear_raw(x.to(DEV), w.to(DEV), b.to(DEV), 0, out=raw.clone(), accumulate=True)
    assert rel_err(acc, 2 * (x.double() @ w.double().t()) + b.double()) < 2e-6


@pytest.mark.parametrize("R,M,N", [(10560, 600, 200), (5280, 100, 200), (333, 300, 100), (7, 4, 8), (40000, 64, 64)])
def test_gemm_tn_weight_gradient_kernel(R, M, N):
    rs = np.random.RandomState(R + M)
    A = torch.from_numpy(rs.randn(R, M).astype(np.float32))
    B = torch.from_numpy(rs.randn(R, N).astype(np.float32))
    C, cs = ops.gemm_tn(A.to(DEV), B.to(DEV), want_colsum=True)
    want = A.double().t() @ B.double()
    assert rel_err(C, want) < 5e-6
    assert rel_err(cs, A.double().sum(0)) < 5e-6
    # strided views (column slices of wider buffers, shifted rows) are read in place
    wideA = torch.from_numpy(rs.randn(R + 3, M + 8).astype(np.float32)).to(DEV)
    wideB = torch.from_numpy(rs.randn(R + 3, 2 * N).astype(np.float32)).to(DEV)
    C2, _ = ops.gemm_tn(wideA[3:, 4:4 + M], wideB[:-3, N:])
    assert rel_err(C2, wideA[3:, 4:4 + M].double().cpu().t() @ wideB[:-3, N:].double().cpu()) < 5e-6


def test_gemm_tn_grouped_with_row_shifts():
    """Grouped launch: mixed shapes, B read at row r + shift (outside rows count as zero), column sums of A,
    outputs written into views of larger tensors; against a float64 torch composition."""
    rs = np.random.RandomState(21)
    specs = [(700, 300, 100, -16), (700, 300, 100, 16), (96, 300, 100, -96), (1760, 64, 200, 0), (333, 12, 36, 5),
             (40, 300, 100, -48)]   # the last one: |shift| > R, every product masked -> zeros
    probs, refs = [], []
    for (R, M, N, sh) in specs:
        wa = torch.from_numpy(rs.randn(R, M + 8).astype(np.float32)).to(DEV)
        wb = torch.from_numpy(rs.randn(R, N + 4).astype(np.float32)).to(DEV)
        A, B = wa[:, 4:4 + M], wb[:, :N]
        Cfull = torch.full((2, M, N + 4), 7.0, device=DEV)
        csfull = torch.full((3, M), 7.0, device=DEV)
        probs.append(dict(A=A, B=B, C=Cfull[1, :, :N], colsum=csfull[2], shift=sh))
        Bs = torch.zeros(R, N, dtype=torch.float64)
        lo, hi = max(0, -sh), min(R, R - sh)
        if hi > lo:
            Bs[lo:hi] = B.double().cpu()[lo + sh:hi + sh]
        refs.append((A.double().cpu().t() @ Bs, A.double().cpu().sum(0), Cfull, csfull))
    ops.gemm_tn_grouped(probs)
    for p, (Cw, csw, Cfull, csfull) in zip(probs, refs):
        scale = max(float(Cw.abs().max()), 1.0)
        assert float((p["C"].double().cpu() - Cw).abs().max()) / scale < 1e-5
        assert float((p["colsum"].double().cpu() - csw).abs().max()) < 1e-3
        assert float(Cfull[0].min()) == 7.0 and float(Cfull[1, :, -4:].min()) == 7.0 and float(csfull[:2].min()) == 7.0
    # no column sums requested
    C = torch.empty(300, 100, device=DEV)
    ops.gemm_tn_grouped([dict(A=probs[0]["A"], B=probs[0]["B"], C=C, shift=-16)])
    assert float((C.double().cpu() - refs[0][0]).abs().max()) / float(refs[0][0].abs().max()) < 1e-5


@pytest.mark.parametrize("R,K,N", [(300, 200, 600), (129, 100, 36), (1000, 36, 130), (257, 512, 200), (64, 8, 4),
                                   (2000, 44, 256)])
def test_linear_bf16_piece_kernel(R, K, N, kernel_variants):
    """The many-row projection kernel (three exact bf16 pieces per operand, csrc/linear_split.hip) forced on small
    and ragged shapes: K tails (K % 32 != 0), row / column tails, strided input and output, bias, ReLU, accumulate."""
    kernel_variants.setenv("MMDFN_LIN_CFG", "7")
    rs = np.random.RandomState(31)
    xw = torch.from_numpy(rs.randn(R, K + 12).astype(np.float32)).to(DEV)
    x = xw[:, 4:4 + K]                                   # row stride K + 12, 16-byte aligned start
    w = torch.from_numpy(rs.randn(N, K).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.randn(N).astype(np.float32)).to(DEV)
    want = x.double() @ w.double().t() + b.double()
    scale = float(want.abs().max())
    y = ops.linear_raw(x, w, b, 0)
    assert float((y.double() - want).abs().max()) / scale < 2e-6
    y = ops.linear_raw(x, w, b, 1)
    assert float((y.double() - want.clamp(min=0)).abs().max()) / scale < 2e-6
    wide = torch.full((R, N + 8), 3.0, device=DEV)
    out = wide[:, :N]
    ops.linear_raw(x, w, None, 0, out=out, accumulate=True)
    assert float((out.double() - (want - b.double() + 3.0)).abs().max()) / scale < 2e-6
    assert float(wide[:, N:].min()) == 3.0
    # same answer as the exact-f32 MFMA kernel to fp32 rounding
    kernel_variants.setenv("MMDFN_LIN_CFG", "8")
    y32 = ops.linear_raw(x, w, b, 0)
    assert float((y32 - ops.linear_raw(x, w, b, 0)).abs().max()) == 0.0
    kernel_variants.setenv("MMDFN_LIN_CFG", "7")
    assert float((ops.linear_raw(x, w, b, 0) - y32).abs().max()) / scale < 2e-6


@pytest.mark.parametrize("lengths,M,d", [([5], 3, 100), ([130, 40, 129], 3, 100), ([257, 31], 2, 200), ([300], 6, 36),
                                         ([128, 128], 1, 100)])
def test_tile_outer_bf16_piece_kernel(lengths, M, d, kernel_variants):
    """K6' (dA = dOut . H^T on the tile pattern) on the bf16-piece path, forced on small and ragged tiles: against a
    float64 product per tile, zero row padding, accumulate mode, and the f32-MFMA kernel."""
    rs = np.random.RandomState(41)
    lay = DialogueLayout.get(lengths, M, DEV)
    N = sum(lengths)
    wide = torch.from_numpy(rs.randn(M * N, 2 * d + 4).astype(np.float32)).to(DEV)
    X, Y = wide[:, :d], wide[:, d + 4:2 * d + 4]          # row-strided views
    kernel_variants.setenv("MMDFN_TILEDOT_SPLIT", "1")
    dt, dc = ops.tile_outer_raw(X, Y, lay)
    kernel_variants.setenv("MMDFN_TILEDOT_SPLIT", "0")
    dt0, dc0 = ops.tile_outer_raw(X, Y, lay)
    assert dc.numel() == 0 or abs_err(dc, dc0) == 0.0
    Xc, Yc = X.double().cpu(), Y.double().cpu()
    start = 0
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        for m in range(M):
            got = dt[base + m * L * ld: base + (m + 1) * L * ld].view(L, ld).double().cpu()
            want = Xc[m * N + start:m * N + start + L] @ Yc[m * N + start:m * N + start + L].t()
            assert float((got[:, :L] - want).abs().max()) / float(want.abs().max()) < 2e-6
            if ld > L:
                assert float(got[:, L:].abs().max()) == 0.0
        start += L
    assert float((dt - dt0).abs().max()) / float(dt0.abs().max()) < 2e-6
    # accumulate into existing gradients
    kernel_variants.setenv("MMDFN_TILEDOT_SPLIT", "1")
    acc_t, acc_c = dt0.clone(), dc0.clone()
    ops.tile_outer_raw(X, Y, lay, dtiles=acc_t, dcross=acc_c)
    assert float((acc_t - 2 * dt0).abs().max()) / float(dt0.abs().max()) < 4e-6


@pytest.mark.parametrize("lengths,M,nl,H", [([40, 17, 5], 3, 2, 100), ([33, 9], 3, 4, 100), ([20, 7], 2, 3, 36),
                                            ([130, 129, 128, 140], 6, 8, 100), ([257, 31], 2, 5, 100), ([12], 3, 16, 100)])
def test_tile_outer_over_all_layers_at_once(lengths, M, nl, H):
    """The GCN stack's adjacency gradient as ONE contraction of width nl H (the layers' operands are column blocks of two
    buffers) against nl accumulating launches of width H: the exact-f32 kernels cut the index into pieces of <= 200 columns,
    the bf16-piece kernel walks it; tiles against a float64 product, cross diagonals against the per-layer sum."""
    rs = np.random.RandomState(43)
    lay = DialogueLayout.get(lengths, M, DEV)
    N = sum(lengths)
    X = torch.from_numpy(rs.randn(M * N, nl * H).astype(np.float32)).to(DEV)
    Y = torch.from_numpy(rs.randn(M * N, nl * H).astype(np.float32)).to(DEV)
    dt, dc = ops.tile_outer_raw(X, Y, lay)
    dt0 = dc0 = None
    for i in range(nl):
        dt0, dc0 = ops.tile_outer_raw(X[:, i * H:(i + 1) * H], Y[:, i * H:(i + 1) * H], lay, dt0, dc0)
    assert float((dt - dt0).abs().max()) / float(dt0.abs().max()) < 4e-6
    if dc.numel():
        assert float((dc - dc0).abs().max()) / float(dc0.abs().max()) < 4e-6
    Xc, Yc = X.double().cpu(), Y.double().cpu()
    start = 0
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        for m in range(M):
            got = dt[base + m * L * ld: base + (m + 1) * L * ld].view(L, ld).double().cpu()
            want = Xc[m * N + start:m * N + start + L] @ Yc[m * N + start:m * N + start + L].t()
            assert float((got[:, :L] - want).abs().max()) / float(want.abs().max()) < 4e-6
            if ld > L:
                assert float(got[:, L:].abs().max()) == 0.0
        start += L


@pytest.mark.parametrize("R,H,nl", [(37, 100, 2), (5280, 100, 4), (24576, 100, 8), (600, 36, 3)])
def test_strided_state_forms_of_the_stack_kernels(R, H, nl):
    """mmdfn_lstm_gate_fwd_ld / mmdfn_gcnii_layer_bwd_ld: hidden states / dhi as column blocks of an (R, nl H) buffer -- bit-equal
    to the contiguous forms, neighbouring blocks untouched."""
    from mm_dfn_amd import _hip
    lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
    rs = np.random.RandomState(84)
    q, c = _rnd(rs, R, H), _rnd(rs, R, H)
    Wih, Whh, b_ih, b_hh = _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H), _rnd(rs, 4 * H)
    wide = _rnd(rs, R, nl * H)
    keep = wide.clone()
    h = wide[:, H:2 * H]                              # state of "layer 1" in place, result goes to block 2 (or 0)
    dst = 2 if nl > 2 else 0
    g0, h0, c0 = (torch.empty(R, n, device=DEV) for n in (4 * H, H, H))
    assert lib.mmdfn_lstm_gate_fwd(P(q), P(h.contiguous()), P(c), P(Wih), P(Whh), P(b_ih), P(b_hh), P(g0), P(h0), P(c0), R, H, st()) == 0
    g1, c1 = torch.empty_like(g0), torch.empty_like(c0)
    assert lib.mmdfn_lstm_gate_fwd_ld(P(q), P(h), P(c), P(Wih), P(Whh), P(b_ih), P(b_hh), P(g1), P(wide[:, dst * H:(dst + 1) * H]),
                                      P(c1), R, H, nl * H, st()) == 0
    assert torch.equal(wide[:, dst * H:(dst + 1) * H], h0) and torch.equal(g1, g0) and torch.equal(c1, c0)
    for b in range(nl):
        if b != dst:
            assert torch.equal(wide[:, b * H:(b + 1) * H], keep[:, b * H:(b + 1) * H])
    # layer backward: dhi into a column block
    dout, W = _rnd(rs, R, H), _rnd(rs, 2 * H, H, scale=0.2)
    gmask = (torch.from_numpy(rs.rand(R, H).astype(np.float32)) > 0.4).float().to(DEV) * 2.0
    dP0, dhi0, dh00 = (torch.empty(R, H, device=DEV) for _ in range(3))
    assert lib.mmdfn_gcnii_layer_bwd(P(dout), P(gmask), P(W), P(dP0), P(dhi0), P(dh00), 0.4, 0.2, R, H, H, 0, st()) == 0
    wide2 = keep.clone()
    dP1, dh01 = torch.empty_like(dP0), torch.empty_like(dh00)
    assert lib.mmdfn_gcnii_layer_bwd_ld(P(dout), P(gmask), P(W), P(dP1), P(wide2[:, H:2 * H]), P(dh01), 0.4, 0.2, R, H, H, 0,
                                        nl * H, st()) == 0
    assert torch.equal(wide2[:, H:2 * H], dhi0) and torch.equal(dP1, dP0) and torch.equal(dh01, dh00)
    assert torch.equal(wide2[:, :H], keep[:, :H]) and (nl < 3 or torch.equal(wide2[:, 2 * H:], keep[:, 2 * H:]))


def test_weight_gradient_batch_kernel():
    """mmdfn_gemm_tn_batch: every weight-gradient contraction of a step in one launch pair -- several segments per
    output (summed in the slab reduction), row shifts, strided operand views, two bias destinations, accumulate."""
    rs = np.random.RandomState(71)
    t = lambda *shape: torch.from_numpy(rs.randn(*shape).astype(np.float32)).to(DEV)
    H = 100
    # output 0: a GRU recurrent weight (300 x 100) from one shifted segment with column sums
    T, R = 9, 8
    dgh, y = t(T * R, 6 * H), t(T * R, 2 * H)
    # output 1: the layer-shared LSTM gate (400 x 100): three segments of different row counts, two bias targets
    dG = [t(n, 4 * H) for n in (333, 64, 1500)]
    q = [t(n, H) for n in (333, 64, 1500)]
    # output 2: a wide projection (200 x 512) accumulated into an existing gradient
    dy, x = t(777, 200), t(777, 512)
    w = [torch.empty(3 * H, H, device=DEV), torch.empty(4 * H, H, device=DEV), t(200, 512)]
    b = [torch.empty(3 * H, device=DEV), torch.empty(4 * H, device=DEV), torch.empty(4 * H, device=DEV), t(200)]
    old_w2, old_b3 = w[2].clone(), b[3].clone()
    o0 = dict(M=3 * H, N=H)
    o1 = dict(M=4 * H, N=H)
    o2 = dict(M=200, N=512)
    batch = [(o0, w[0], [b[0]], 0, [(dgh[:, 3 * H:], y[:, H:], R)]),
             (o1, w[1], [b[1], b[2]], 0, [(dG[i], q[i], 0) for i in range(3)]),
             (o2, w[2], [b[3]], 1, [(dy, x, 0)])]
    ops._launch_wgrad_batch(batch)
    A, B = dgh[:, 3 * H:].double().cpu(), y[:, H:].double().cpu()
    want0 = A[:-R].t() @ B[R:]                                # row r of A pairs with row r + R of B
    assert rel_err(w[0], want0) < 1e-5
    assert rel_err(b[0], A.sum(0)) < 1e-5
    want1 = sum(g.double().cpu().t() @ v.double().cpu() for g, v in zip(dG, q))
    assert rel_err(w[1], want1) < 1e-5
    wb = sum(g.double().cpu().sum(0) for g in dG)
    assert rel_err(b[1], wb) < 1e-5 and torch.equal(b[1], b[2])
    assert rel_err(w[2], old_w2.double().cpu() + dy.double().cpu().t() @ x.double().cpu()) < 1e-5
    assert rel_err(b[3], old_b3.double().cpu() + dy.double().cpu().sum(0)) < 1e-5
    # a negative shift (forward GRU direction: h_{t-1}) and a long reduction (many splits)
    Rl = 40000
    a2, b2 = t(Rl, 100), t(Rl, 200)
    out = torch.empty(100, 200, device=DEV)
    ops._launch_wgrad_batch([(dict(M=100, N=200), out, [], 0, [(a2, b2, -16)])])
    assert rel_err(out, a2[16:].double().cpu().t() @ b2[:-16].double().cpu()) < 1e-5
    # ragged sizes: partial tiles on both axes, row counts that are not multiples of the 32-row chunk, shifts, column
    # sums from the first column tile only
    for (Mo, No, Rr, sh) in [(36, 36, 70, 0), (100, 104, 333, 3), (52, 112, 1, 0), (300, 116, 257, -2), (64, 212, 999, 0),
                             (600, 224, 2050, 5), (44, 228, 96, 0)]:
        a3, b3 = t(Rr, Mo + 8)[:, 4:4 + Mo], t(Rr, No)
        o3, c3 = torch.empty(Mo, No, device=DEV), torch.empty(Mo, device=DEV)
        ops._launch_wgrad_batch([(dict(M=Mo, N=No), o3, [c3], 0, [(a3, b3, sh)])])
        Ad, Bd = a3.double().cpu(), b3.double().cpu()
        want = (Ad[:Rr - sh].t() @ Bd[sh:]) if sh >= 0 else (Ad[-sh:].t() @ Bd[:Rr + sh])
        assert rel_err(o3, want) < 1e-5, (Mo, No, Rr, sh)
        assert rel_err(c3, Ad.sum(0)) < 1e-5, (Mo, No, Rr, sh)


def test_weight_gradient_batch_tall_segments():
    """A batch with enough long segments (cfg5-sized graph-stack backward) runs them on the one-workgroup-per-output-slab form
    (gemm_tn_tall_kernel, LDS-DMA operand rows): 400 x 100 (7 row tiles per wave), 100 x 100 (2), 336 x 80 (the last wave holds
    no row tile); in the same launch, on the tiled bodies: short, shifted, odd-row-count, 200-column and 200 x 512 segments; several
    segments per output, strided views, column sums, accumulate -- against fp64."""
    rs = np.random.RandomState(77)
    t = lambda *shape: torch.from_numpy(rs.randn(*shape).astype(np.float32)).to(DEV)
    H = 100
    Rl = 8192 + 16 * 37
    dG = [t(Rl, 4 * H), t(24576, 4 * H), t(640, 4 * H)]               # tall, tall (used by six segments), short (tiled form)
    q = [t(Rl, 2 * H)[:, H:], t(24576, H), t(640, H)]                 # (a strided view as the B operand)
    reps = [0, 1, 1, 1, 1, 1, 1, 2]
    hi, dP = t(12288, H + 4)[:, 4:], t(12288, H)                     # 100 x 100
    a5, b5 = t(9600, 336), t(9600, 80)
    a6, b6 = t(16384, 100), t(16384, 200)
    a7, b7 = t(8192, 400), t(8192, 100)                              # shifted: tiled form
    a9, b9 = t(4096, 200), t(4096, 512)
    a10, b10 = t(4100, 100), t(4100, 100)                            # rows not a multiple of 16: tiled form
    w = [torch.empty(4 * H, H, device=DEV), t(H, H), torch.empty(336, 80, device=DEV), torch.empty(100, 200, device=DEV),
         torch.empty(400, 100, device=DEV), torch.empty(200, 512, device=DEV), torch.empty(100, 100, device=DEV)]
    b = [torch.empty(4 * H, device=DEV), torch.empty(4 * H, device=DEV), t(H), torch.empty(336, device=DEV),
         torch.empty(200, device=DEV)]
    old_w1, old_b2 = w[1].clone(), b[2].clone()
    batch = [(dict(M=4 * H, N=H), w[0], [b[0], b[1]], 0, [(dG[i], q[i], 0) for i in reps]),
             (dict(M=H, N=H), w[1], [b[2]], 1, [(hi, dP, 0)]),
             (dict(M=336, N=80), w[2], [b[3]], 0, [(a5, b5, 0)]),
             (dict(M=100, N=200), w[3], [], 0, [(a6, b6, 0)]),
             (dict(M=400, N=100), w[4], [], 0, [(a7, b7, 3)]),
             (dict(M=200, N=512), w[5], [b[4]], 0, [(a9, b9, 0)]),
             (dict(M=100, N=100), w[6], [], 0, [(a10, b10, 0)])]
    ops._launch_wgrad_batch(batch)
    d = lambda x: x.double().cpu()
    assert rel_err(w[0], sum(d(dG[i]).t() @ d(q[i]) for i in reps)) < 1e-5
    assert rel_err(b[0], sum(d(dG[i]).sum(0) for i in reps)) < 1e-5 and torch.equal(b[0], b[1])
    assert rel_err(w[1], d(old_w1) + d(hi).t() @ d(dP)) < 1e-5
    assert rel_err(b[2], d(old_b2) + d(hi).sum(0)) < 1e-5
    assert rel_err(w[2], d(a5).t() @ d(b5)) < 1e-5 and rel_err(b[3], d(a5).sum(0)) < 1e-5
    assert rel_err(w[3], d(a6).t() @ d(b6)) < 1e-5
    assert rel_err(w[4], d(a7)[:-3].t() @ d(b7)[3:]) < 1e-5
    assert rel_err(w[5], d(a9).t() @ d(b9)) < 1e-5 and rel_err(b[4], d(a9).sum(0)) < 1e-5
    assert rel_err(w[6], d(a10).t() @ d(b10)) < 1e-5
    # non-finite values in one operand row reach exactly the outputs that row feeds (the tall form reads a few floats past a row's
    # end into accumulator rows / columns that are never stored)
    a8, b8 = t(8192, 100), t(8192, 100)
    a8[4097, 99] = float("inf")
    b8[:, 99] = 0.0
    b8[5000, 99] = float("inf")
    o8 = torch.empty(100, 100, device=DEV)
    big = (dict(M=4 * H, N=H), w[0], [], 0, [(dG[1], q[1], 0)] * 7)      # (enough work in the batch for the tall form)
    ops._launch_wgrad_batch([big, (dict(M=100, N=100), o8, [], 0, [(a8, b8, 0)])])
    assert torch.isfinite(o8[:99, :99]).all() and not torch.isfinite(o8[99]).any() and not torch.isfinite(o8[:, 99]).any()


def _wgrad_reference(A, B, sh):
    Ad, Bd = A.double().cpu(), B.double().cpu()
    R = Ad.shape[0]
    if abs(sh) >= R:
        return torch.zeros(Ad.shape[1], Bd.shape[1], dtype=torch.float64)
    return (Ad[:R - sh].t() @ Bd[sh:]) if sh >= 0 else (Ad[-sh:].t() @ Bd[:R + sh])


# (rows, M, N, row shift, lda, ldb): what the bf16-piece form's tiles, chunks, splits and groups meet at their edges
WGRAD_SPLIT_CASES = [
    (1000, 100, 100, 0, 100, 100),      # one tile, 7 of 8 row tiles
    (333, 300, 200, 45, 600, 200),      # positive shift larger than a chunk: clamped rows in the last chunks of B
    (333, 300, 200, -7, 600, 200),      # negative shift: the first rows of B do not exist
    (320, 300, 200, 45, 600, 200),      # the same with every chunk full
    (97, 8, 4, 0, 8, 4),                # a sliver of one tile, one ragged chunk per split
    (31, 20, 228, 0, 24, 232),          # fewer rows than a chunk, three column blocks, strided operands
    (64, 128, 112, 5, 128, 112),        # exactly one tile, two chunks: one per group
    (4097, 132, 116, 1, 132, 116),      # one row and one column block more than a tile, one row more than the chunks
    (96, 100, 100, 64, 100, 100),       # shift of two chunks in a three-chunk segment
    (40, 400, 100, -40, 400, 100),      # shift = -rows: nothing pairs up (zeros), four row tiles
    (7040, 300, 100, 64, 600, 200),     # a party-GRU recurrent weight of cfg2
]


@pytest.mark.parametrize("R,M,N,sh,lda,ldb", WGRAD_SPLIT_CASES)
def test_weight_gradient_batch_bf16_piece_form(R, M, N, sh, lda, ldb, kernel_variants):
    """csrc/gemm_tn_split.hip (the batch's default form: operands cut into bf16 pieces, transposed through LDS planes, two
    wave groups in opposite phases) at the edges of its tiling, against fp64 and against the exact-f32 forms of gemm_tn.hip
    (MMDFN_TN_SPLIT=0 in the tuning build): weights and column sums at fp32 level, and a different kernel really ran."""
    rs = np.random.RandomState(1000 + R + M)
    t = lambda *shape: torch.from_numpy(rs.randn(*shape).astype(np.float32)).to(DEV)
    A, B = t(R, lda)[:, lda - M:], t(R, ldb)[:, :N]
    want, wcol = _wgrad_reference(A, B, sh), A.double().cpu().sum(0)
    got = {}
    for form in ("1", "0"):
        kernel_variants.setenv("MMDFN_TN_SPLIT", form)
        C, c1 = torch.full((M, N), float("nan"), device=DEV), torch.full((M,), float("nan"), device=DEV)
        ops._launch_wgrad_batch([(dict(M=M, N=N), C, [c1], 0, [(A, B, sh)])])
        got[form] = (C, c1)
        scale = float(want.abs().max()) + 1e-30
        assert float((C.double().cpu() - want).abs().max()) <= 2e-6 * scale + 1e-30, form
        assert rel_err(c1, wcol) < 1e-5, form
    if R * M * N > 100000 and abs(sh) < R:
        assert float((got["1"][0] - got["0"][0]).abs().max()) > 0.0          # (two different kernels)


def test_weight_gradient_batch_exact_f32_forms_stay_covered(kernel_variants):
    """The tiled and tall exact-f32 forms of gemm_tn.hip (what a batch with a misaligned operand falls back to) through
    the tuning switch: the two batch tests above, unchanged."""
    kernel_variants.setenv("MMDFN_TN_SPLIT", "0")
    test_weight_gradient_batch_kernel()
    test_weight_gradient_batch_tall_segments()


def test_weight_gradient_batch_misaligned_operand_falls_back():
    """mmdfn_gemm_tn_batch with an operand that starts 4 bytes off a 16-byte boundary (ops never passes one: it copies such
    views; a C-ABI caller may): the bf16-piece form needs aligned float4 rows, the batch falls back to the tiled form."""
    rs = np.random.RandomState(5)
    t = lambda *shape: torch.from_numpy(rs.randn(*shape).astype(np.float32)).to(DEV)
    base = t(700 * 104 + 1)
    A = base[1:].view(700, 104)[:, :100]
    B = t(700, 100)
    assert A.data_ptr() % 16 == 4
    C, c1 = torch.empty(100, 100, device=DEV), torch.empty(100, device=DEV)
    ops._launch_wgrad_batch([(dict(M=100, N=100), C, [c1], 0, [(A, B, 0)])])
    assert rel_err(C, A.double().cpu().t() @ B.double().cpu()) < 1e-5
    assert rel_err(c1, A.double().cpu().sum(0)) < 1e-5


@pytest.mark.parametrize("R,K,n1,n2", [(300, 200, 300, 300), (7040, 200, 300, 300), (129, 36, 4, 100), (2000, 100, 260, 52),
                                       (16640, 200, 300, 300)])      # (>= 16384 rows: the input gradient on the hand-written kernel)
def test_two_block_projection(R, K, n1, n2):
    """mmdfn_linear2: output columns from two weight / bias parameters (the directions of a bidirectional GRU layer),
    forward vs torch, gradients of x / both weights / both biases vs autograd on the concatenated form."""
    rs = np.random.RandomState(72)
    mk = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32)).to(DEV)
    x = mk(R, K).requires_grad_(True)
    w1, w2 = torch.nn.Parameter(mk(n1, K)), torch.nn.Parameter(mk(n2, K))
    b1, b2 = torch.nn.Parameter(mk(n1)), torch.nn.Parameter(mk(n2))
    W = mk(R, n1 + n2)
    y = ops.linear2(x, w1, w2, b1, b2)
    (y * W).sum().backward()
    xr = x.detach().double().cpu().requires_grad_(True)
    pr = [p.detach().double().cpu().requires_grad_(True) for p in (w1, w2, b1, b2)]
    yr = xr @ torch.cat([pr[0], pr[1]], 0).t() + torch.cat([pr[2], pr[3]], 0)
    (yr * W.double().cpu()).sum().backward()
    assert rel_err(y, yr) < 2e-6
    assert rel_err(x.grad, xr.grad) < 1e-5
    for p, r in zip((w1, w2, b1, b2), pr):
        assert p.grad is not None and rel_err(p.grad, r.grad) < 1e-5
    # with a stacked weight copy for the input gradient (what bigru2 passes): same gradients
    x2 = x.detach().clone().requires_grad_(True)
    g_before = [p.grad.clone() for p in (w1, w2, b1, b2)]
    (ops.linear2(x2, w1, w2, b1, b2, torch.cat([w1, w2], 0).detach()) * W).sum().backward()
    assert rel_err(x2.grad, xr.grad) < 1e-5
    for p, g0 in zip((w1, w2, b1, b2), g_before):
        assert rel_err(p.grad, 2 * g0.double().cpu()) < 1e-5          # accumulated into the existing .grad
    # stacked weight AND bias given: launches with few rows run as one library GEMM on the stacked operands, the others
    # on the two-block kernel -- same result either way
    y3 = ops.linear2(x.detach(), w1, w2, b1, b2, torch.cat([w1, w2], 0).detach(), torch.cat([b1, b2], 0).detach())
    assert rel_err(y3, yr) < 2e-6
    # no biases (the project-then-gather path adds the bias after the gather)
    y0 = ops.linear2(x.detach(), w1, w2, None, None)
    assert rel_err(y0, yr - torch.cat([pr[2], pr[3]], 0)) < 2e-6


# ------------------------------------------------------------------------------------------------------------------
# fused stages of the GCNII stack (csrc/gcn_stack.hip) against float64 restatements of model_GCN.py:453-472,176-189
# ------------------------------------------------------------------------------------------------------------------
def _rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.randn(*shape) * scale).astype(np.float32)).to(DEV)


def _keep(rs, *shape, p=0.5):
    return torch.from_numpy(((rs.uniform(size=shape) > p) / (1 - p)).astype(np.float32)).to(DEV)


@pytest.mark.parametrize("R,F,H,masked,residue", [(37, 200, 100, True, True), (16, 52, 36, False, False), (1000, 200, 100, True, True),
                                                   (5, 8, 4, True, False), (5280, 200, 100, True, True)])
def test_gcn_input_stage_kernels(R, F, H, masked, residue):
    from mm_dfn_amd import _hip
    lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
    rs = np.random.RandomState(81)
    x, W0, b0 = _rnd(rs, R, F), _rnd(rs, H, F, scale=0.1), _rnd(rs, H)
    mx, m0 = (_keep(rs, R, F), _keep(rs, R, H)) if masked else (None, None)
    ld = F + H if residue else F
    xd_buf = torch.full((R, ld), 7.0, device=DEV)
    h0, cur0 = torch.empty(R, H, device=DEV), torch.empty(R, H, device=DEV)
    assert lib.mmdfn_gcn_input_fwd(P(x), P(mx), P(W0), P(b0), P(m0), P(xd_buf), P(h0), P(cur0), R, F, H, ld, 1.0, st()) == 0
    d = lambda t: None if t is None else t.double().cpu()
    xd_w = d(x) * (d(mx) if masked else 1.0)
    h0_w = torch.relu(xd_w @ d(W0).t() + d(b0))
    cur_w = h0_w * (d(m0) if masked else 1.0)
    assert rel_err(xd_buf[:, :F], xd_w) < 1e-6 and (not residue or float(xd_buf[:, F:].min()) == 7.0)
    assert rel_err(h0, h0_w) < 2e-6 and rel_err(cur0, cur_w) < 2e-6
    # backward
    dcur0, dh0 = _rnd(rs, R, H), _rnd(rs, R, H)
    dout = _rnd(rs, R, ld)
    dpre, dx = torch.empty(R, H, device=DEV), torch.empty(R, F, device=DEV)
    assert lib.mmdfn_gcn_input_bwd(P(dcur0), P(m0), P(dh0), P(h0), P(W0), P(dout) if residue else None, P(mx), P(dpre), P(dx),
                                   R, F, H, ld, 1.0, st()) == 0
    dpre_w = (d(dcur0) * (d(m0) if masked else 1.0) + d(dh0)) * (h0_w > 0)
    dx_w = (dpre_w @ d(W0) + (d(dout)[:, :F] if residue else 0.0)) * (d(mx) if masked else 1.0)
    assert rel_err(dpre, dpre_w) < 1e-6 and rel_err(dx, dx_w) < 2e-6


@pytest.mark.parametrize("R,H,first", [(4100, 100, False), (333, 100, True), (2000, 96, False), (530, 68, False), (700, 84, True)])
def test_lstm_gate_backward_wide_form_is_bit_identical(R, H, first, kernel_variants):
    """Round 5: from 16 384 rows on the K8 backward runs ONE column block per product (dq | dh), every consumer wave contracting
    its second 16-column tile against weight fragments held in registers.  Forced here at small sizes (ragged last block, first
    layer, several H of its range 64 < H <= 100) against the two-column-block form: the same MFMA order per output column, so
    the results are bit-identical."""
    from mm_dfn_amd import _hip
    P, st = _hip.ptr, _hip.stream
    rs = np.random.RandomState(91)
    gates = torch.rand(R, 4 * H, device=DEV)
    c, cn, dha, dhb, dcn = (_rnd(rs, R, H) for _ in range(5))
    dres = _rnd(rs, R, H)
    Wih, Whh = _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H, H, scale=0.2)
    outs = {}
    for mode in ("0", "1"):
        kernel_variants.setenv("MMDFN_GATE_BWD_WIDE", mode)
        kernel_variants.setenv("MMDFN_GATE_WS", "1")
        dG, dq = torch.empty(R, 4 * H, device=DEV), torch.empty(R, H, device=DEV)
        dcp = None if first else torch.empty(R, H, device=DEV)
        dhp = None if first else torch.empty(R, H, device=DEV)
        assert _hip.lib().mmdfn_lstm_gate_bwd(P(gates), None if first else P(c), P(cn), P(dha), P(dhb), P(dcn), P(Wih), P(Whh),
                                               P(dres), P(dG), P(dcp), P(dq), P(dhp), R, H, 0 if first else 1, H, st()) == 0
        outs[mode] = [t.clone() for t in (dG, dq, dcp, dhp) if t is not None]
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
    assert float(outs["1"][1].abs().max()) > 0.0


@pytest.mark.parametrize("R,H,first", [(37, 100, False), (37, 100, True), (600, 36, False), (3, 4, True), (2000, 96, False),
                                       (5280, 100, False),
                                       # larger launches (ragged last block, first layer, narrow H, BASELINE cfg5 rows)
                                       (4097, 100, False), (4100, 100, True), (4111, 36, False), (24576, 100, False),
                                       # the bf16-piece forward (>= 16 384 rows): first layer, ragged last tile, narrow / odd unit blocks
                                       (24576, 100, True), (16500, 36, False), (17001, 96, True), (16385, 68, False)])
def test_lstm_gate_kernels(R, H, first):
    from mm_dfn_amd import _hip
    lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
    rs = np.random.RandomState(82)
    q = _rnd(rs, R, H)
    h, c = (None, None) if first else (_rnd(rs, R, H), _rnd(rs, R, H))
    Wih, Whh, b_ih, b_hh = _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H), _rnd(rs, 4 * H)
    bsum = b_ih + b_hh
    gates, h_out, c_out = (torch.empty(R, n, device=DEV) for n in (4 * H, H, H))
    assert lib.mmdfn_lstm_gate_fwd(P(q), P(h), P(c), P(Wih), P(Whh), P(b_ih), P(b_hh), P(gates), P(h_out), P(c_out), R, H,
                                   st()) == 0
    if R == 37:       # one bias pointer (the sum formed by the caller) gives the same result
        g1, h1, c1 = (torch.empty_like(t) for t in (gates, h_out, c_out))
        assert lib.mmdfn_lstm_gate_fwd(P(q), P(h), P(c), P(Wih), P(Whh), P(bsum), None, P(g1), P(h1), P(c1), R, H, st()) == 0
        assert rel_err(h1, h_out) < 1e-6 and rel_err(g1, gates) < 1e-6
    d = lambda t: None if t is None else t.double().cpu()
    qd = d(q).requires_grad_(True)
    hd = None if first else d(h).requires_grad_(True)
    cd = None if first else d(c).requires_grad_(True)
    G = qd @ d(Wih).t() + d(bsum) + (0 if first else hd @ d(Whh).t())
    i, f, g, o = (torch.sigmoid(G[:, :H]), torch.sigmoid(G[:, H:2 * H]), torch.tanh(G[:, 2 * H:3 * H]), torch.sigmoid(G[:, 3 * H:]))
    c_w = i * g + (0 if first else f * cd)
    h_w = o * torch.tanh(c_w)
    # fp32 accumulation of 200 products per pre-activation (|G| up to ~10 with two unit-variance biases) + the hardware
    # exp / rcp of the gate non-linearities: worst element of 2 M at the 1e-5 level, mean error ~1e-7
    assert rel_err(h_out, h_w) < 1e-5 and rel_err(c_out, c_w) < 1e-5
    assert rel_err(gates, torch.cat([i, f, g, o], 1)) < 1e-5
    if not first and lib.mmdfn_lstm_gate_takes_planes(R, H):
        # round 5: the many-row form with the cell's weights cut once into piece planes -- the same pieces, the same products
        planes = torch.empty(int(lib.mmdfn_lstm_gate_planes_workspace(H)), device=DEV)
        assert lib.mmdfn_lstm_gate_cut_weights(P(Wih), P(Whh), P(planes), H, st()) == 0
        g2, h2, c2 = (torch.empty_like(t) for t in (gates, h_out, c_out))
        assert lib.mmdfn_lstm_gate_fwd_pre(P(q), P(h), P(c), P(Wih), P(Whh), P(b_ih), P(b_hh), P(g2), P(h2), P(c2), R, H, H,
                                           P(planes), st()) == 0
        assert torch.equal(g2, gates) and torch.equal(h2, h_out) and torch.equal(c2, c_out)
    # backward: upstream gradients on h' (two addends), on c', and the residual addend of dq
    dh_a, dh_b, dc_n, dres_w = _rnd(rs, R, H), _rnd(rs, R, H), _rnd(rs, R, H), _rnd(rs, R, H + 12)
    dres = dres_w[:, 4:4 + H]                                       # strided residual gradient
    G.retain_grad()
    (h_w * (d(dh_a) + d(dh_b))).sum().backward(retain_graph=True)
    (c_w * d(dc_n)).sum().backward()
    dG, dq = torch.empty(R, 4 * H, device=DEV), torch.empty(R, H, device=DEV)
    dcp = None if first else torch.empty(R, H, device=DEV)
    dhp = None if first else torch.empty(R, H, device=DEV)
    assert lib.mmdfn_lstm_gate_bwd(P(gates), P(c), P(c_out), P(dh_a), P(dh_b), P(dc_n), P(Wih), P(Whh), P(dres), P(dG), P(dcp),
                                   P(dq), P(dhp), R, H, 0 if first else 1, H + 12, st()) == 0
    assert rel_err(dG, G.grad) < 5e-6
    assert rel_err(dq, qd.grad + d(dres)) < 5e-6
    if not first:
        assert rel_err(dhp, hd.grad) < 5e-6 and rel_err(dcp, cd.grad) < 5e-6
    # optional inputs absent
    assert lib.mmdfn_lstm_gate_bwd(P(gates), P(c), P(c_out), P(dh_a), None, None, P(Wih), P(Whh), None, P(dG), P(dcp),
                                   P(dq), P(dhp), R, H, 0 if first else 1, 0, st()) == 0
    assert torch.isfinite(dq).all()


def test_lstm_gate_forward_bf16_piece_form_against_the_exact_f32_form(kernel_variants):
    """The many-row K8 forward (csrc/lstm_gate_split.hip: three exact bf16 pieces per operand, six MFMA products, cell math from
    the accumulators) against the exact-f32 producer / consumer kernel on the same inputs, both against fp64: its error is at
    most 4x the exact-f32 kernel's (+ the transcendental units' 3e-7); also forced on a launch below its row threshold."""
    from mm_dfn_amd import _hip
    P, st = _hip.ptr, _hip.stream
    for R, H, first in ((24576, 100, False), (300, 100, False), (1000, 36, True)):
        rs = np.random.RandomState(84)
        q = _rnd(rs, R, H)
        h, c = (None, None) if first else (_rnd(rs, R, H), _rnd(rs, R, H))
        Wih, Whh, b_ih, b_hh = _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H, H, scale=0.2), _rnd(rs, 4 * H), _rnd(rs, 4 * H)
        res = {}
        for mode in ("0", "1"):
            kernel_variants.setenv("MMDFN_GATE_SPLIT", mode)
            gates, h_out, c_out = (torch.empty(R, n, device=DEV) for n in (4 * H, H, H))
            assert _hip.lib().mmdfn_lstm_gate_fwd(P(q), P(h), P(c), P(Wih), P(Whh), P(b_ih), P(b_hh), P(gates), P(h_out),
                                                  P(c_out), R, H, st()) == 0
            res[mode] = (gates, h_out, c_out)
        d = lambda t: None if t is None else t.double().cpu()
        G = d(q) @ d(Wih).t() + d(b_ih) + d(b_hh) + (0 if first else d(h) @ d(Whh).t())
        i, f, g, o = (torch.sigmoid(G[:, :H]), torch.sigmoid(G[:, H:2 * H]), torch.tanh(G[:, 2 * H:3 * H]), torch.sigmoid(G[:, 3 * H:]))
        c_w = i * g + (0 if first else f * d(c))
        want = (torch.cat([i, f, g, o], 1), o * torch.tanh(c_w), c_w)
        for k in range(3):
            e_f32 = float((res["0"][k].double().cpu() - want[k]).abs().max())
            e_bf = float((res["1"][k].double().cpu() - want[k]).abs().max())
            assert e_bf <= 4 * e_f32 + 3e-7, (R, H, first, k, e_bf, e_f32)
        assert float((res["0"][1] - res["1"][1]).abs().max()) > 0.0          # it really was another kernel


@pytest.mark.parametrize("ws", ["auto", "1", "0"])
@pytest.mark.parametrize("R,H,masked,has_q,ldo", [(37, 100, True, True, 300), (500, 100, False, False, 100), (9, 36, True, True, 36),
                                                  (2100, 96, True, False, 96), (5280, 100, True, True, 300),
                                                  # many rows: the producer / consumer forms (round 5) by default; ragged last block
                                                  (24576, 100, True, True, 300), (16401, 100, False, True, 100)])
def test_gcnii_layer_kernels(R, H, masked, has_q, ldo, ws, kernel_variants):
    """``ws``: the 4-wave kernels ("0") / the 8-wave producer / consumer kernels ("1") forced through the tuning build, or the
    dispatcher's own choice ("auto": producer / consumer from 64 rows on)."""
    from mm_dfn_amd import _hip
    if ws != "auto":
        kernel_variants.setenv("MMDFN_LAYER_WS", ws)
    lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
    rs = np.random.RandomState(83)
    hi, h0, W = _rnd(rs, R, H), _rnd(rs, R, H), _rnd(rs, 2 * H, H, scale=0.1)
    q = _rnd(rs, R, H) if has_q else None
    m = _keep(rs, R, H) if masked else None
    theta, alpha = 0.405, 0.2
    wide = torch.full((R, ldo), 3.0, device=DEV)
    out = wide[:, ldo - H:]
    gmask = torch.empty(R, H, device=DEV)
    assert lib.mmdfn_gcnii_layer_fwd(P(hi), P(h0), P(W), P(q), P(m), P(out), P(gmask), theta, alpha, R, H, ldo, 1.0, st()) == 0
    d = lambda t: None if t is None else t.double().cpu()
    hid, h0d = d(hi).requires_grad_(True), d(h0).requires_grad_(True)
    pre = theta * (torch.cat([hid, h0d], 1) @ d(W)) + (1 - theta) * ((1 - alpha) * hid + alpha * h0d)
    want = torch.relu(pre) * (d(m) if masked else 1.0) + (d(q) if has_q else 0.0)
    assert rel_err(out, want) < 2e-6
    assert ldo == H or float(wide[:, :ldo - H].min()) == 3.0
    assert torch.equal(gmask.cpu() != 0, ((pre > 0) & ((d(m) != 0) if masked else torch.ones_like(pre, dtype=torch.bool))))
    dwide = _rnd(rs, R, ldo)
    dout = dwide[:, ldo - H:]
    (want * d(dout)).sum().backward()
    dP, dhi = torch.empty(R, H, device=DEV), torch.empty(R, H, device=DEV)
    dh0 = _rnd(rs, R, H)
    old = dh0.clone()
    assert lib.mmdfn_gcnii_layer_bwd(P(dout), P(gmask), P(W), P(dP), P(dhi), P(dh0), theta, alpha, R, H, ldo, 1, st()) == 0
    gg = d(dout) * (pre > 0) * (d(m) if masked else 1.0)
    assert rel_err(dP, theta * gg) < 1e-6
    assert rel_err(dhi, hid.grad) < 5e-6
    assert rel_err(dh0, d(old) + h0d.grad) < 5e-6
    assert lib.mmdfn_gcnii_layer_bwd(P(dout), P(gmask), P(W), P(dP), P(dhi), P(dh0), theta, alpha, R, H, ldo, 0, st()) == 0
    assert rel_err(dh0, h0d.grad) < 5e-6


@pytest.mark.parametrize("nl,reason,residue,p", [(2, True, True, 0.0), (3, True, True, 0.5), (2, False, True, 0.5), (4, True, False, 0.0),
                                                 (1, True, True, 0.3)])
def test_fused_stack_equals_op_by_op_path(nl, reason, residue, p):
    """GCNII_lyc through the single-node fused stack vs the op-by-op HIP path (itself golden / oracle checked) on the
    same keep-masks: output, input gradient, adjacency gradient (through the feature gradient) and every parameter."""
    from mm_dfn_amd import gcn_stack
    rs = np.random.RandomState(84)
    lengths = [9, 4, 17]
    N = sum(lengths)
    feats0 = _rnd(rs, 3, N, 200)
    Rw = _rnd(rs, 3 * N, 300 if residue else 100)
    res = []
    R_, H_ = 3 * N, 100
    flags = ms = None
    if p > 0:
        flags = torch.from_numpy((rs.uniform(size=R_ * 200 + (1 + nl) * R_ * H_) > p).astype(np.float32)).to(DEV)
        ms = 1.0 / (1.0 - p)
    for fused in (True, False):
        net = GCNII_lyc(nfeat=200, nlayers=nl, nhidden=100, nclass=6, dropout=p, lamda=0.5, alpha=0.2, variant=True,
                        return_feature=True, use_residue=residue, reason_flag=reason)
        net.load_state_dict(synthetic.seeded_state_dict(net.state_dict(), 85))
        net = net.to(DEV).train()
        feats = feats0.clone().requires_grad_(True)
        adj = ops.build_adjacency(feats, lengths)
        if fused:
            y = gcn_stack.gcn_stack(adj.stacked_feats.reshape(R_, 200), adj, flags, ms or 1.0, net.lamda, net.alpha, reason,
                                    residue, net.fcs[0].weight, net.fcs[0].bias, net.rnn, [c.weight for c in net.convs])
        else:
            ones = torch.ones(R_ * 200 + (1 + nl) * R_ * H_, device=DEV)
            y = _op_by_op_with_masks(net, adj, ones if flags is None else flags * ms, R_, H_)
        (y * Rw).sum().backward()
        res.append((y.detach(), feats.grad.clone(), {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}))
    (y1, g1, p1), (y0, g0, p0) = res
    assert rel_err(y1, y0) < 5e-6
    assert rel_err(g1, g0) < 5e-5
    assert sorted(p1) == sorted(p0)
    for k in p0:
        assert rel_err(p1[k], p0[k]) < 5e-5, k


def _op_by_op_with_masks(net, adj, flat, R, H):
    """GCNII_lyc's op-by-op HIP path with externally supplied keep-masks (layout of gcn_stack.gcn_stack)."""
    import math
    F_ = 200
    nl = len(net.convs)
    mx = flat[:R * F_].view(R, F_)
    m0 = flat[R * F_:R * (F_ + H)].view(R, H)
    ml = [flat[R * (F_ + H) + i * R * H: R * (F_ + H) + (i + 1) * R * H].view(R, H) for i in range(nl)]
    x = adj.stacked_feats.reshape(R, F_) * mx
    h0 = ops.linear(x, net.fcs[0].weight, net.fcs[0].bias, act=1)
    cur = h0 * m0
    h = c = None
    if net.reason_flag:
        bsum = (net.rnn.bias_ih_l0 + net.rnn.bias_hh_l0).detach()
    for i, con in enumerate(net.convs):
        q = cur
        if net.reason_flag:
            G = ops.gate_linear(q, h, net.rnn.weight_ih_l0, net.rnn.weight_hh_l0, bsum, net.rnn.bias_ih_l0, net.rnn.bias_hh_l0)
            h, c = ops.lstm_pointwise(G, c)
            cur = h
        S2 = ops.propagate_concat(adj, cur, h0)
        Pm = ops.matmul_kn(S2, con.weight)
        cur = ops.gcnii_combine(Pm, S2, q if net.reason_flag else None, ml[i], math.log(net.lamda / (i + 1) + 1), net.alpha)
    return torch.cat([x, cur], -1) if net.use_residue else cur


@pytest.mark.parametrize("N,Wd,C,p,strided", [(37, 900, 6, 0.5, False), (1760, 900, 6, 0.0, True), (333, 1800, 7, 0.3, False),
                                                (5, 300, 2, 0.5, True), (16384, 900, 6, 0.5, False)])
def test_head_kernels(N, Wd, C, p, strided):
    """K9: dropout -> ReLU -> Linear -> log_softmax (model.py:1328-1337) fused, against float64 autograd on the same mask."""
    from mm_dfn_amd import _hip
    rs = np.random.RandomState(91)
    wide = _rnd(rs, N, Wd + 8)
    Fm = (wide[:, 4:4 + Wd] if strided else wide[:, :Wd].contiguous()).detach().requires_grad_(True)
    W, b = _rnd(rs, C, Wd, scale=0.05).requires_grad_(True), _rnd(rs, C).requires_grad_(True)
    mask = torch.from_numpy((rs.uniform(size=(N, Wd)) > p).astype(np.float32)).to(DEV) if p > 0 else None
    ms = 1.0 / (1.0 - p)
    logp = ops._Head.apply(Fm, mask, ms, W, b)
    G = _rnd(rs, N, C)
    (logp * G).sum().backward()
    Fd, Wd_, bd = (t.detach().double().cpu().requires_grad_(True) for t in (Fm, W, b))
    z = torch.relu(Fd * (mask.double().cpu() * ms if mask is not None else 1.0))
    want = torch.log_softmax(z @ Wd_.t() + bd, 1)
    (want * G.double().cpu()).sum().backward()
    assert rel_err(logp, want) < 2e-6
    assert rel_err(Fm.grad, Fd.grad) < 1e-5
    assert rel_err(W.grad, Wd_.grad) < 1e-5 and rel_err(b.grad, bd.grad) < 1e-5
    # bit-reproducible weight gradient (fixed reduction order)
    W2, b2, F2 = (t.detach().clone().requires_grad_(True) for t in (W, b, Fm))
    (ops._Head.apply(F2, mask, ms, W2, b2) * G).sum().backward()
    assert torch.equal(W2.grad, W.grad) and torch.equal(b2.grad, b.grad)


@pytest.mark.parametrize("M,N,Wm,C,p", [(3, 1760, 300, 6, 0.5), (6, 257, 100, 7, 0.0), (2, 5, 4, 2, 0.3)])
def test_head_reads_the_stacked_graph_output_in_place(M, N, Wm, C, p):
    """The (M, N, Wm) graph output standing for cat([F[0], .., F[M-1]], -1) (model_mm.py:113-117): the head kernels read
    the blocks where they are and write dF in the same layout -- same numbers as the head on the materialised
    concatenation (bit for bit: the same arithmetic in the same order)."""
    rs = np.random.RandomState(92)
    F3 = _rnd(rs, M, N, Wm).requires_grad_(True)
    W, b = _rnd(rs, C, M * Wm, scale=0.05).requires_grad_(True), _rnd(rs, C).requires_grad_(True)
    mask = torch.from_numpy((rs.uniform(size=(N, M * Wm)) > p).astype(np.float32)).to(DEV) if p > 0 else None
    ms = 1.0 / (1.0 - p)
    G = _rnd(rs, N, C)
    logp = ops._Head.apply(F3, mask, ms, W, b)
    (logp * G).sum().backward()
    F2 = F3.detach().permute(1, 0, 2).reshape(N, M * Wm).contiguous().requires_grad_(True)
    W2, b2 = (t.detach().clone().requires_grad_(True) for t in (W, b))
    want = ops._Head.apply(F2, mask, ms, W2, b2)
    (want * G).sum().backward()
    assert torch.equal(logp, want)
    assert torch.equal(F3.grad, F2.grad.view(N, M, Wm).permute(1, 0, 2))
    assert torch.equal(W.grad, W2.grad) and torch.equal(b.grad, b2.grad)
    # ... and the public op draws its own keep flags for the stacked form too (training) / is exact in eval
    ev = ops.head(F3.detach(), W.detach(), b.detach(), 0.5, False)
    z = torch.relu(F2.detach().double().cpu())
    assert rel_err(ev, torch.log_softmax(z @ W.detach().double().cpu().t() + b.detach().double().cpu(), 1)) < 2e-6
    tr = ops.head(F3.detach(), W.detach(), b.detach(), 0.5, True)
    assert tr.shape == (N, C) and bool(torch.isfinite(tr).all())


@pytest.mark.parametrize("shapes", [[(1760, 100, 200), (1760, 512, 200), (1760, 100, 200)], [(77, 600, 200)],
                                    [(1056, 36, 33), (3, 768, 5), (130, 200, 600)]])
def test_few_row_projection_group(shapes):
    """csrc/linear_small.hip: a group of small projections in one launch (32 x 32 tiles, the four waves split K, fixed-order
    LDS reduction) against the fp64 product; two-block weights, the K-major form, ReLU and accumulate."""
    rs = np.random.RandomState(17)
    t = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32)).to(DEV)
    probs, want = [], []
    for R, K, N in shapes:
        x, w, b = t(R, K), t(N, K), t(N)
        probs.append(dict(x=x, w=w, b=b))
        want.append(torch.relu(x.double() @ w.double().t() + b.double()))
    got = ops.linear_group_raw(probs, act=1)
    for g, wnt in zip(got, want):
        assert rel_err(g, wnt) < 2e-6
    # two weight blocks + accumulate into an existing output; K-major weight (y = x @ wk)
    R, K, N = max(shapes, key=lambda q: q[2])
    x, w1, w2, b1, b2, base, wk = t(R, K), t(N - 40, K), t(40, K), t(N - 40), t(40), t(R, N), t(K, N)
    out = base.clone()
    ops.linear_group_raw([dict(x=x, w=w1, w2=w2, b=b1, b2=b2, out=out, accumulate=True), dict(x=x, wk=wk)])
    ref = base.double() + x.double() @ torch.cat([w1, w2]).double().t() + torch.cat([b1, b2]).double()
    assert rel_err(out, ref) < 2e-6
    y2 = ops.linear_group_raw([dict(x=x, wk=wk)])[0]
    assert rel_err(y2, x.double() @ wk.double()) < 2e-6
    # bit-reproducible (fixed summation order)
    assert torch.equal(y2, ops.linear_group_raw([dict(x=x, wk=wk)])[0])
    # out-of-place addend (mmdfn_linear_group_addend): y = x @ wk + z with z only read -- an aligned contiguous addend, a strided
    # view of a wider buffer, and one whose rows start 4 bytes off a 16-byte boundary (scalar epilogue reads), next to a problem
    # without an addend in the same launch
    z1 = t(R, N)
    zwide = t(R, N + 8)
    z2 = zwide[:, 4:4 + N]
    z3 = t(R * N + 1)[1:].view(R, N)
    keep = [z1.clone(), zwide.clone(), z3.clone()]
    outs = ops.linear_group_raw([dict(x=x, wk=wk, addend=z1), dict(x=x, wk=wk, addend=z2), dict(x=x, wk=wk),
                                 dict(x=x, wk=wk, addend=z3)])
    prod = x.double() @ wk.double()
    assert rel_err(outs[0], prod + z1.double()) < 2e-6
    assert rel_err(outs[1], prod + z2.double()) < 2e-6
    assert torch.equal(outs[2], y2)
    assert rel_err(outs[3], prod + z3.double()) < 2e-6
    assert torch.equal(z1, keep[0]) and torch.equal(zwide, keep[1]) and torch.equal(z3, keep[2])      # (never written)
    assert all(o.data_ptr() not in (z1.data_ptr(), z2.data_ptr(), z3.data_ptr()) for o in outs)
