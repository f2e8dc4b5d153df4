"""Randomised stress of the bf16-piece kernels (propagate_split, tile_dot_split, linear_split): many ragged shapes,
each launched several times -- results must be bit-identical across repeats (no LDS / pipeline race) and agree with
the exact-f32 MFMA kernels / an fp64 product to fp32 rounding.  python tools/stress_split.py [n_cases] [seed]"""
import os

os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops  # noqa: E402
from mm_dfn_amd.layout import BlockTileAdjacency, DialogueLayout  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda"
worst = dict(prop=0.0, tile=0.0, lin=0.0)
for case in range(n_cases):
    B = int(rs.randint(1, 9))
    M = int(rs.randint(1, 7))
    Lmax = int(rs.choice([7, 40, 130, 300, 600]))
    lengths = [int(x) for x in rs.randint(1, Lmax + 1, size=B)]
    d = 4 * int(rs.randint(1, rs.choice([30, 130])))
    lay = DialogueLayout.get(lengths, M, DEV)
    N = lay.N
    tiles = torch.from_numpy(rs.uniform(-1, 1, size=lay.tile_elems).astype(np.float32)).to(DEV)
    # poison the row padding: it must never reach a product
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        if ld > L:
            tiles[base: base + M * L * ld].view(M * L, ld)[:, L:] = float("nan")
    cross = torch.from_numpy(rs.uniform(-1, 1, size=(lay.npairs, N)).astype(np.float32)).to(DEV)
    H = torch.from_numpy(rs.randn(M * N, d).astype(np.float32)).to(DEV)
    X = torch.from_numpy(rs.randn(M * N, d).astype(np.float32)).to(DEV)
    os.environ["MMDFN_PROP_CFG"] = "9"
    ref = ops.propagate_raw(tiles, cross, H, lay)
    os.environ["MMDFN_PROP_CFG"] = "8"
    outs = [ops.propagate_raw(tiles, cross, H, lay) for _ in range(3)]
    assert all(torch.equal(outs[0], o) for o in outs[1:]), ("propagate_split not deterministic", lengths, M, d)
    assert torch.isfinite(outs[0]).all(), ("propagate_split non-finite", lengths, M, d)
    e = float((outs[0] - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
    worst["prop"] = max(worst["prop"], e)
    assert e < 3e-6, ("propagate_split vs f32", e, lengths, M, d)
    os.environ["MMDFN_TILEDOT_SPLIT"] = "0"
    t0, _ = ops.tile_outer_raw(X, H, lay)
    os.environ["MMDFN_TILEDOT_SPLIT"] = "1"
    ts = [ops.tile_outer_raw(X, H, lay)[0] for _ in range(3)]
    assert all(torch.equal(ts[0], t) for t in ts[1:]), ("tile_dot_split not deterministic", lengths, M, d)
    e = float((ts[0] - t0).abs().max()) / max(float(t0.abs().max()), 1e-6)
    worst["tile"] = max(worst["tile"], e)
    assert e < 3e-6, ("tile_dot_split vs f32", e, lengths, M, d)
    R, K, Nn = int(rs.randint(1, 3000)), 4 * int(rs.randint(2, 160)), int(rs.randint(1, 700))
    x = torch.from_numpy(rs.randn(R, K).astype(np.float32)).to(DEV)
    w = torch.from_numpy(rs.randn(Nn, K).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.randn(Nn).astype(np.float32)).to(DEV)
    os.environ["MMDFN_LIN_CFG"] = "7"
    ys = [ops.linear_raw(x, w, b, 1) for _ in range(3)]
    assert all(torch.equal(ys[0], y) for y in ys[1:]), ("linear_split not deterministic", R, K, Nn)
    want = (x.double() @ w.double().t() + b.double()).clamp(min=0)
    e = float((ys[0].double() - want).abs().max()) / max(float(want.abs().max()), 1e-6)
    worst["lin"] = max(worst["lin"], e)
    assert e < 3e-6, ("linear_split vs fp64", e, R, K, Nn)
    if case % 10 == 9:
        print("case %d ok, worst rel err so far %s" % (case + 1, worst), flush=True)
print("stress ok:", n_cases, "cases", worst)
