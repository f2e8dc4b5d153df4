"""K6 (production dispatch: fp32-input bf16-piece kernel) time per launch vs the number of dialogues: how much of the
cfg5 B=32 launch is round quantisation (768 workgroups on 512 slots = 1.5 rounds)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_dfn_amd import ops
dev = torch.device("cuda")
for B in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "16,21,32,42,43,64".split(","))]:
    l5 = [512] * B
    nset = max(3, int(900e6 / (B * 8.81e6)) + 1)
    sets = []
    for i in range(nset):
        g = torch.Generator(device=dev).manual_seed(500 + i)
        adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
        H = torch.randn(6 * sum(l5), 100, device=dev, generator=g)
        sets.append((adj, H, torch.empty_like(H)))
    def fn(i):
        adj, H, out = sets[i % nset]
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=out)
    for i in range(nset):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    iters = 7 * nset
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / (iters * 4) * 1e3
    items = B * 6 * 4
    by = sets[0][0].layout.propagate_bytes(100)
    print("B=%3d items=%5d rounds=%.2f  %.1f us  %.3f us/item  frac %.3f" % (B, items, items / 512, us, us / items, by / (us * 1e-6) / 8e12), flush=True)
    del sets, g
    torch.cuda.empty_cache()
