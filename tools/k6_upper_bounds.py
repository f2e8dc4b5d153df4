"""Upper bounds of the two work-removing K6 levers of VERDICT r04 item 2, measured BEFORE building them (tuning build,
results of the ablated launches are wrong, only their time counts), protocol of bench.py's roofline_cfg5 leg (3 rotating
(adjacency, H, out) sets = 846 MB > the MALL, launches captured in a hipGraph, HIP events):
  (a) tail tile for columns 96..:   MMDFN_SPLIT_ABLC=4 drops ALL 12 MFMAs of the fourth 32-column tile per chunk (a 16x16x32
      tail tile still has to issue 12 half-length ones: it can buy at most half of this)
  (b) cross-modal rows once:        MMDFN_PROP_ABL=2 reads ONE cross-modal row per output row instead of M - 1 = 5 (what a
      precomputed addend would leave); MMDFN_PROP_ABL=1 reads none
"""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops  # noqa: E402

dev = "cuda"
l5 = [512] * 32
d = 100
sets = []
for i in range(3):
    g = torch.Generator(device=dev).manual_seed(500 + i)
    adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
    H = torch.randn(6 * sum(l5), d, device=dev, generator=g)
    sets.append((adj, H, torch.empty_like(H)))
lay = sets[0][0].layout
alg = lay.propagate_bytes(d)


def run(ablc, abl, tail=0):
    os.environ["MMDFN_SPLIT_ABLC"] = str(ablc)
    os.environ["MMDFN_PROP_ABL"] = str(abl)
    os.environ["MMDFN_SPLIT_TAIL"] = str(tail)
    for adj, H, o in sets:
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=o)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for it in range(21):
            adj, H, o = sets[it % 3]
            ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=o)
    for _ in range(15):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 105 * 1e3


rows = [("full kernel, four 32-column tiles", 0, 0, 0), ("BUILT: three tiles + the 16-column tail tile", 0, 0, 1),
        ("(a) no fourth-tile MFMAs", 4, 0, 0), ("(b) one cross-modal row instead of five", 0, 2, 0),
        ("(a) + (b)", 4, 2, 0), ("tail tile + (b)", 0, 2, 1), ("no cross-modal rows", 0, 1, 0), ("tail tile + no cross-modal rows", 0, 1, 1)]
for rep in range(2):
    for name, ablc, abl, tail in rows:
        us = run(ablc, abl, tail)
        print("%-44s %6.1f us   %.3f of 8 TB/s" % (name, us, alg / (us * 1e-6) / 8e12), flush=True)
