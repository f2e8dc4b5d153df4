"""Compile-time ablations of the K6 plane kernel (tuning library, MMDFN_PLANES_ABL), rotating buffers, cfg5 B=32."""
import os, sys
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


def timeit(fn, iters=21, warm=12, reps=4):
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


def f_planes(i):
    adj, H, P, out = sets[i]
    P_.propagate_planes_raw(adj.tiles, adj.cross, H, P, adj.layout, out=out)


names = {0: "full", 1: "no cross terms", 2: "no MFMA", 4: "no cutting", 12: "no A loads, no cutting", 16: "no DMA",
         48: "no DMA, no frag reads", 60: "MFMA + barriers + epilogue only", 64: "no epilogue", 66: "no MFMA, no epilogue",
         76: "no A side, no epilogue", 112: "no B side, no epilogue", 124: "MFMA + barriers only", 126: "loop skeleton"}
for abl in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else names.keys())]:
    os.environ["MMDFN_PLANES_ABL"] = str(abl)
    print("%3d %-34s %.1f us" % (abl, names.get(abl, "?"), timeit(f_planes)), flush=True)
