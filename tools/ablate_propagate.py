"""Ablation of the propagate kernel phases (MMDFN_PROP_ABL bits: 1 no cross-terms, 2 no MFMA, 4 no H staging loads,
8 no tile-strip loads).  Prints HIP-event times; workloads large enough that host launch overhead is hidden."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops
from bench_kernels import WORKLOADS, timeit
name = sys.argv[1]; cfg = sys.argv[2]
w = WORKLOADS[name]
lengths = w["lengths"]; M, d = w["M"], w["d"]; N = sum(lengths)
adj = ops.build_adjacency(torch.randn(M, N, 200, device="cuda"), lengths)
H = torch.randn(M * N, d, device="cuda")
os.environ["MMDFN_PROP_CFG"] = cfg
for abl in [0, 1, 2, 3, 4, 8, 12, 6, 14, 15]:
    os.environ["MMDFN_PROP_ABL"] = str(abl)
    t = timeit(lambda: ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout), 30)
    print("cfg %s abl %2d : %.1f us" % (cfg, abl, t * 1e6), flush=True)
