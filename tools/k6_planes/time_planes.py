"""K6 timing, rotating buffer sets (846 MB > the 256 MB MALL): fp32-input bf16-piece kernel vs the plane kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_dfn_amd import ops
import planes_ops as P_

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
l5 = [512] * B
d = 100
sets = []
for i in range(3):
    g = torch.Generator(device=dev).manual_seed(500 + i)
    adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
    H = torch.randn(6 * sum(l5), d, device=dev, generator=g)
    sets.append((adj, H, P_.cut_planes(H), torch.empty_like(H)))


def timeit(fn, iters=21, warm=15, reps=5):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i % 3)
    for _ in range(warm):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


def f_split(i):
    adj, H, P, out = sets[i]
    ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=out)


def f_planes(i):
    adj, H, P, out = sets[i]
    P_.propagate_planes_raw(adj.tiles, adj.cross, H, P, adj.layout, out=out)


def f_cut(i):
    adj, H, P, out = sets[i]
    P_.cut_planes(H, out=P)


adj, H, P, out = sets[0]
a = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
b = P_.propagate_planes_raw(adj.tiles, adj.cross, H, P, adj.layout)
print("max |split - planes| = %.3g" % float((a - b).abs().max()))
bytes_ = adj.layout.propagate_bytes(d)
for name, fn in (("split", f_split), ("planes", f_planes), ("cut", f_cut), ("split", f_split), ("planes", f_planes)):
    us = timeit(fn)
    print("%-7s %.1f us  frac %.3f" % (name, us, bytes_ / (us * 1e-6) / 8e12))
