"""MFMA linear kernel vs torch F.linear (hipBLASLt) on the hot-path shapes; graph-captured timing."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
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
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

shapes = [("party gi", 10560, 200, 600), ("text gi", 1760, 200, 600), ("party dX", 10560, 600, 200), ("linear_v", 1760, 512, 200),
          ("linear_l", 1760, 100, 200), ("fcs0", 5280, 200, 100), ("lstm gate", 5280, 100, 400), ("S2.W", 5280, 200, 100),
          ("cfg4 party gi", 21120, 200, 600), ("cfg3 party gi", 28512, 200, 600), ("cfg5 fcs0", 98304, 200, 100),
          ("cfg5 gate", 98304, 100, 400), ("cfg5 proj", 98304, 512, 200), ("party gi 2mod", 7040, 200, 600), ("party gi 2mod cfg4", 14080, 200, 600), ("mid", 8192, 200, 256), ("mid2", 16384, 100, 128)]
if os.environ.get("HOT"):
    # the GEMMs that stay on the library in the cfg2 step (profiles/r02_linear_vs_hipblaslt.txt): modality projections
    # (forward) and the GRU input-gradient GEMMs dG . [W_ih; W_ih_reverse] (for the hand-written kernel the weight is
    # given pre-transposed, i.e. its best case: no transpose launch counted)
    shapes = [("linear_a / linear_l fwd", 1760, 100, 200), ("linear text fwd", 1760, 512, 200), ("ctx GRU dX", 1760, 600, 200),
              ("party GRU dX", 7040, 600, 200), ("party GRU gi (ours in production)", 7040, 200, 600),
              ("ctx GRU gi (ours in production)", 1760, 200, 600), ("cfg4 ctx GRU gi", 3520, 200, 600),
              ("cfg4 linear_a fwd", 3520, 100, 200), ("cfg4 linear text fwd", 3520, 512, 200), ("cfg4 ctx GRU dX", 3520, 600, 200),
              ("cfg4 party GRU dX", 14080, 600, 200), ("cfg3 text fwd", 1056, 600, 200), ("cfg3 party dX", 19008, 600, 200)]
if os.environ.get("ONLY"):
    shapes = [s_ for s_ in shapes if os.environ["ONLY"] in s_[0]]
cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1]
for name, R, K, N in shapes:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    ref = torch.nn.functional.linear(x, w, b)
    t0 = gtime(lambda: torch.nn.functional.linear(x, w, b))
    line = "%-14s R=%6d K=%4d N=%4d  torch %7.1f us (%5.1f TF)" % (name, R, K, N, t0, 2.0 * R * K * N / t0 / 1e6)
    for c in cfgs:
        if c >= 0: os.environ["MMDFN_LIN_CFG"] = str(c)
        y = ops.linear_raw(x, w, b)
        err = float((y - ref).abs().max() / ref.abs().max())
        t1 = gtime(lambda: ops.linear_raw(x, w, b))
        line += " | cfg%2d %7.1f us (%5.1f TF) err %.1e" % (c, t1, 2.0 * R * K * N / t1 / 1e6, err)
    print(line, flush=True)
