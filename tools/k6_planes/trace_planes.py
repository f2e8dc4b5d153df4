"""Per-workgroup phase timeline of the K6 plane kernel (s_memtime stamps, tuning library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mm_dfn_amd import ops
import planes_ops as P_

dev = torch.device("cuda")
B = 32
l5 = [512] * B
d = 100
sets = []
for i in range(3):
    g = torch.Generator(device=dev).manual_seed(500 + i)
    adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
    H = torch.randn(6 * sum(l5), d, device=dev, generator=g)
    sets.append((adj, H, P_.cut_planes(H), torch.empty_like(H)))
nwg = 32 * 6 * 4
trace = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
os.environ["MMDFN_TRACE_PTR"] = str(trace.data_ptr())
os.environ["MMDFN_PLANES_ABL"] = "128"
for it in range(30):
    adj, H, P, out = sets[it % 3]
    P_.propagate_planes_raw(adj.tiles, adj.cross, H, P, adj.layout, out=out)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(nwg, 8)
st = t[:, :7].astype(np.float64)
t0 = st[:, 0].min()
st -= t0
hw = t[:, 7]
xcc = hw >> 32
hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf
se = (hwid >> 13) & 0x7   # gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
sh = (hwid >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 50 + cu
print("memtime ticks are 100 MHz (10 ns)?  total span %.0f ticks" % st[:, 6].max())
names = ["entry", "setup done", "prologue done", "loop done", "epi pass0", "epi pass1", "end"]
d_ = np.diff(st, axis=1)
print("phase durations (ticks): mean / p10 / p90")
for k in range(6):
    print("  %-14s -> %-14s %8.0f %8.0f %8.0f" % (names[k], names[k + 1], d_[:, k].mean(), np.percentile(d_[:, k], 10), np.percentile(d_[:, k], 90)))
print("start times: first-round (%d WGs start < 10%% of span)" % (st[:, 0] < 0.1 * st[:, 6].max()).sum())
order = np.argsort(st[:, 0])
print("distinct CUs:", len(set(key.tolist())))
for k in sorted(set(key.tolist()))[:3]:
    idx = [i for i in order if key[i] == k]
    print("CU", k, [(int(st[i, 0]), int(st[i, 2]), int(st[i, 3]), int(st[i, 6])) for i in idx])
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "trace_planes.npy"), t)
