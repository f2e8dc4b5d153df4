"""Sweep the propagate v2 tilings (MMDFN_PROP_CFG override) on one workload; run under rocprofv3
--kernel-trace --stats to get true per-kernel durations (kernel names carry the template args)."""
import os

os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops, synthetic  # noqa: E402
from bench_kernels import WORKLOADS  # noqa: E402

name = sys.argv[1]
cfgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(8))
w = WORKLOADS[name]
rs = np.random.RandomState(1)
lengths = w["lengths"] or synthetic.make_lengths(rs, w["B"], w["L"], True, min_len=3)
M, d = w["M"], w["d"]
N = sum(lengths)
adj = ops.build_adjacency(torch.randn(M, N, 200, device="cuda"), lengths)
H = torch.randn(M * N, d, device="cuda")
ref = None
for c in [-1] + cfgs:
    if c >= 0:
        os.environ["MMDFN_PROP_CFG"] = str(c)
    for _ in range(20):
        out = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    torch.cuda.synchronize()
    if ref is None:
        ref = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, transpose=True)  # v1 kernel, symmetric A
    print(c, "max|v2 - v1| =", float((out - ref).abs().max()), flush=True)
