"""Split-K weight-gradient kernel (gemm_tn) vs torch (A.t() @ B + A.sum(0)) on the hot-path shapes."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops, _hip

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

shapes = [("party dW_ih", 10560, 600, 200), ("party dW_hh", 10464, 300, 100), ("text dW_ih", 1760, 600, 200),
          ("text dW_hh", 1744, 300, 100), ("fcs0", 5280, 100, 200), ("lstm G", 5280, 400, 100), ("S2.W", 5280, 200, 100),
          ("linear_v", 1760, 200, 512), ("linear_l", 1760, 200, 100)]
splits_list = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
for name, R, M, N in shapes:
    A = torch.randn(R, M, device="cuda"); B = torch.randn(R, N, device="cuda")
    t0 = gtime(lambda: (A.t() @ B, A.sum(0)))
    line = "%-12s R=%6d M=%4d N=%4d torch %6.1f us" % (name, R, M, N, t0)
    for sp in splits_list:
        if sp: os.environ["MMDFN_TN_SPLITS"] = str(sp)
        else: os.environ.pop("MMDFN_TN_SPLITS", None)
        t1 = gtime(lambda: ops.gemm_tn(A, B, want_colsum=True))
        line += " | s=%2d %6.1f us" % (sp if sp else _hip.lib().mmdfn_gemm_tn_splits(R, M, N), t1)
    print(line, flush=True)
