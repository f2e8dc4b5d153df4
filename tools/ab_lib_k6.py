"""Same-process A/B of two BUILDS of the K6 kernel at cfg5 (protocol of bench.py's roofline_cfg5 leg): the production library against
the tuning library, which may have been built from a modified source (build the production library, edit, then
`python -c "from mm_dfn_amd import build; build.build(tuning=True)"`).  Round 5: `#pragma unroll 5` of the cross-modal loop (all five
modalities' loads in flight) 90.4 us against 90.0 us for `unroll 2`: no difference; non-temporal loads of the adjacency strip 104 us
against 90.5 (the 32-byte pieces of a 128-byte line are fetched by four consecutive chunks: around the L2 each of them goes to HBM)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mm_dfn_amd import ops, _hip
dev = "cuda"; l5 = [512] * 32; d = 100
sets = []
for i in range(3):
    g = torch.Generator(device=dev).manual_seed(500 + i)
    adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
    H = torch.randn(6 * sum(l5), d, device=dev, generator=g)
    sets.append((adj, H, torch.empty_like(H)))
alg = sets[0][0].layout.propagate_bytes(d)
def run(tuning):
    _hip.set_tuning(tuning)
    for adj, H, o in sets: ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=o)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for it in range(21):
            adj, H, o = sets[it % 3]; ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=o)
    for _ in range(15): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 105 * 1e3
for rep in range(3):
    for t in (False, True):
        us = run(t); print("lib=%s %.1f us %.3f" % ("tuning(B)" if t else "prod(A)", us, alg / (us * 1e-6) / 8e12), flush=True)
