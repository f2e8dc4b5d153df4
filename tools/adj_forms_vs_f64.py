"""Adjacency build (K5): the strip form (adjacency_small.hip) and the many-launch form (adjacency.hip) against an fp64 evaluation of
the oracle -- adjacency entries (absolute) and d(features) (relative to the largest entry).  Tuning build (MMDFN_ADJ_SMALL)."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import mmdfn_oracle as O
from mm_dfn_amd import ops
for lengths, M, D in [([110, 64, 33], 3, 200), ([33, 1, 32], 3, 200), ([128, 3], 2, 64)]:
    rs = np.random.RandomState(31)
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(M, N, D).astype(np.float32))
    R = torch.from_numpy(rs.randn(M * N, M * N).astype(np.float32))
    f64 = feats.double().requires_grad_(True)
    want = O.create_big_adj([f64[m] for m in range(M)], lengths, 0.7)
    (want * R.double()).sum().backward()
    for form in ("0", "1"):
        os.environ["MMDFN_ADJ_SMALL"] = form
        fg = feats.cuda().requires_grad_(True)
        adj = ops.build_adjacency(fg, lengths, 0.7)
        ea = float((adj.to_dense().detach().double().cpu() - want.detach()).abs().max())
        (adj.to_dense() * R.cuda()).sum().backward()
        eg = float((fg.grad.double().cpu() - f64.grad).abs().max() / f64.grad.abs().max())
        print(lengths, M, D, "form", form, "adj abs err vs f64 %.2e   grad rel err vs f64 %.2e" % (ea, eg))
