"""Timing ablations of K6 at the bench workload (cfg2: 16 dialogues x 110 utterances, M = 3, d = 100; tuning build,
MMDFN_PROP_ABL bits: 1 no cross-modal terms, 2 no MFMA, 4 no H loads, 8 no tile loads), rotating buffer sets."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops, synthetic
from mm_dfn_amd.layout import DialogueLayout
B, L, M, d = 16, 110, 3, 100
lay = DialogueLayout([L] * B, M, "cuda")
NS = 48
sets = []
for _ in range(NS):
    tiles = torch.randn(int(lay.tile_base[-1]), device="cuda")
    cross = torch.randn(lay.npairs, lay.N, device="cuda")
    H = torch.randn(M * lay.N, d, device="cuda")
    sets.append((tiles, cross, H))
k = [0]
def run():
    t, c, h = sets[k[0] % NS]; k[0] += 1
    ops.propagate_raw(t, c, h, lay)
def gtime(fn, iters=48):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3
for abl in (0, 1, 2, 4, 8, 12, 14, 15):
    os.environ["MMDFN_PROP_ABL"] = str(abl)
    print("  abl %2d: %.2f us" % (abl, gtime(run)), flush=True)
