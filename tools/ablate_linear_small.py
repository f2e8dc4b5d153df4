"""Compile-time timing ablations of the few-row projection kernel (tuning build, MMDFN_LSM_ABL: 1 no MFMA, 2 no loads, 4 no
reduction / epilogue); R = 1760, K = 512, N = 200."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops
x, w, b = torch.randn(1760, 512, device="cuda"), torch.randn(200, 512, device="cuda"), torch.randn(200, device="cuda")
def gtime(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3
for abl in (0, 1, 2, 4, 6, 5, 7, 0):
    os.environ["MMDFN_LSM_ABL"] = str(abl)
    print("abl %d: %.1f us" % (abl, gtime(lambda: ops.linear_group_raw([dict(x=x, w=w, b=b)]))), flush=True)
print("library: %.1f us" % gtime(lambda: torch.nn.functional.linear(x, w, b)))
