"""Plane-form projection (csrc/linear_planes.hip) against the forms it replaces, at the GRU input contractions of the BASELINE
configs: forward (R x 200 -> 600, two-block weight) and input gradient (R x 600 -> 200).  Captured launches, HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops  # noqa: E402

dev = "cuda"


def timeit(fn, n=40, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


for R in (1760, 3520, 7040, 8800, 14080, 19008):
    x = torch.randn(R, 200, device=dev)
    dy = torch.randn(R, 600, device=dev)
    w1 = torch.randn(300, 200, device=dev) * 0.1
    w2 = torch.randn(300, 200, device=dev) * 0.1
    wcat = torch.cat([w1, w2])
    b1, b2 = torch.randn(300, device=dev), torch.randn(300, device=dev)
    y = torch.empty(R, 600, device=dev)
    dx = torch.empty(R, 200, device=dev)
    ops.weight_planes(w1, w2)
    ops.weight_planes(w1, w2, True)
    t_pl = timeit(lambda: ops.linear_planes_raw(x, w1, w2, b1, b2, out=y))
    t_old = timeit(lambda: ops.linear2(x, w1, w2, b1, b2) if False else ops.dense_nk(x, wcat, torch.cat([b1, b2]), out=y))
    t_grp = timeit(lambda: ops.linear_group_raw([dict(x=x, w=w1, w2=w2, b=b1, b2=b2, out=y)]))
    t_dpl = timeit(lambda: ops.linear_planes_raw(dy, w1, w2, transposed=True, out=dx))
    t_dold = timeit(lambda: ops.dense_kn(dy, wcat))
    t_cut = timeit(lambda: ops.refresh_planes())
    fl = 2.0 * R * 200 * 600
    print("R=%6d  fwd: planes %6.1f us (%5.1f TF)  dense_nk %6.1f  few-row %6.1f | dX: planes %6.1f us  dense_kn %6.1f | refresh (%d entries) %5.1f us"
          % (R, t_pl, fl / t_pl / 1e6, t_old, t_grp, t_dpl, t_dold, len(ops._PLANES), t_cut), flush=True)
