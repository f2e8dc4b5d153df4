"""Many-row projection y = x W^T + b: the bf16-piece many-row kernel (ops.linear_raw -> csrc/linear_split.hip / linear.hip)
against the LDS-staged few-row kernel (ops.linear_group_raw -> csrc/linear_small.hip) at the row counts where
ops.linear_preferred switches between them.    python tools/bench_proj_forms.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops  # noqa: E402


def timeit(fn, n=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for R, K, N in ((7040, 200, 600), (5280, 200, 100), (5280, 100, 400), (3520, 200, 600), (14080, 200, 600), (10560, 200, 100),
                (19008, 200, 600), (4096, 256, 600), (1760, 200, 600)):
    xs = [torch.randn(R, K, device="cuda") for _ in range(4)]
    w = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda")
    outs = [torch.empty(R, N, device="cuda") for _ in range(4)]
    it = [0]

    def many():
        i = it[0] % 4; it[0] += 1
        ops.linear_raw(xs[i], w, b, 0, out=None)

    def few():
        i = it[0] % 4; it[0] += 1
        ops.linear_group_raw([dict(x=xs[i], w=w, b=b)], 0)
    tm, tf = timeit(many), timeit(few)
    fl = 2.0 * R * K * N
    print("R=%6d K=%4d N=%4d  many-row %.1f us (%.0f TF)   few-row %.1f us (%.0f TF)   preferred=%s" % (
        R, K, N, tm, fl / tm / 1e6, tf, fl / tf / 1e6, ops.linear_preferred(R, K, N)), flush=True)
