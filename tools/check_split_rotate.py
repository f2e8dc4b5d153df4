"""cfg5 propagate timing with a ROTATING set of output buffers (each launch writes a fresh 39 MB buffer, as a real
layer stack does) vs one reused buffer: separates Infinity-Cache residency from kernel quality."""
import os

os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lengths = [512] * B
M, d = 6, 100
N = sum(lengths)
adj = ops.build_adjacency(torch.randn(M, N, 200, device="cuda"), lengths)
H = torch.randn(M * N, d, device="cuda")
nb = 24
outs = [torch.empty(M * N, d, device="cuda") for _ in range(nb)]
Hs = [torch.randn(M * N, d, device="cuda") for _ in range(nb)]
for mode in ("same out, same H", "rotating out", "rotating out and H"):
    for _ in range(5):
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=outs[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(96):
        o = outs[0] if mode.startswith("same") else outs[i % nb]
        h = Hs[i % nb] if mode.endswith("and H") else H
        ops.propagate_raw(adj.tiles, adj.cross, h, adj.layout, out=o)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 96 * 1e3
    print("B=%d %-22s %.1f us  %.1f%% of 8 TB/s" % (B, mode, us, 100 * adj.layout.propagate_bytes(d) / (us * 1e-6) / 8e12), flush=True)
