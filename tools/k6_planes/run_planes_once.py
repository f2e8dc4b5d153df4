"""A handful of K6 launches for rocprofv3 passes: argv[1] = planes | split, argv[2] = launches (rotating 3 buffer sets)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_dfn_amd import ops
import planes_ops as P_
dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "planes"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
l5 = [512] * 32
sets = []
for i in range(3):
    g = torch.Generator(device=dev).manual_seed(500 + i)
    adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
    H = torch.randn(6 * sum(l5), 100, device=dev, generator=g)
    sets.append((adj, H, P_.cut_planes(H), torch.empty_like(H)))
torch.cuda.synchronize()
for it in range(n):
    adj, H, P, out = sets[it % 3]
    if which == "planes":
        P_.propagate_planes_raw(adj.tiles, adj.cross, H, P, adj.layout, out=out)
    else:
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=out)
torch.cuda.synchronize()
