"""Compile-time timing ablations of the few-row projection kernel (tuning build, MMDFN_LSM_ABL: 1 no MFMA, 2 no operand DMA,
4 no reduction / epilogue); the tile form follows the launcher's rule (MMDFN_LSM_TM=32|64 overrides)."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops
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
for (R, K, N, km, abls) in [(1760, 512, 200, 0, (0, 1, 2, 4, 3, 7)), (7040, 600, 200, 1, (0, 1, 2, 4, 3, 7)),
                            (7040, 200, 600, 0, (0, 1, 2, 4, 3, 7))]:
    x = torch.randn(R, K, device="cuda")
    w = torch.randn(K, N, device="cuda") if km else torch.randn(N, K, device="cuda")
    prob = dict(x=x, wk=w) if km else dict(x=x, w=w, b=torch.randn(N, device="cuda"))
    print("R=%d K=%d N=%d kmajor=%d" % (R, K, N, km))
    for abl in abls:
        os.environ["MMDFN_LSM_ABL"] = str(abl)
        print("  abl %2d: %.1f us" % (abl, gtime(lambda: ops.linear_group_raw([prob]))), flush=True)
    os.environ.pop("MMDFN_LSM_ABL")
