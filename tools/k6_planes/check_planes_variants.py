"""Correctness + timing of the K6 plane-kernel variants (tuning library: MMDFN_PLANES_VER = 1 | 2 | 3, MMDFN_PLANES_NT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np
import torch
from mm_dfn_amd import ops
import planes_ops as P_

dev = torch.device("cuda")
variants = [("1", "0", "0"), ("2", "0", "0"), ("2", "1", "0"), ("2", "0", "1"), ("2", "1", "1")]
if len(sys.argv) > 1:
    variants = [tuple(v.split(":")) for v in sys.argv[1].split(",")]


def setv(v):
    os.environ["MMDFN_PLANES_VER"], os.environ["MMDFN_PLANES_NT"], os.environ["MMDFN_PLANES_KM"] = v


# ---- correctness on ragged / odd shapes against the fp64 dense product
from util import random_block_adjacency
for lengths, M, d in [([5], 3, 100), ([32, 33, 31, 64], 3, 100), ([129, 127, 128, 200], 2, 128), ([260, 40], 6, 64),
                      ([513], 3, 100), ([9, 31], 1, 36), ([512, 300, 257, 511], 6, 100)]:
    adj, dense, _, _ = random_block_adjacency(13, lengths, M, dev)
    lay = adj.layout
    tiles = adj.tiles.clone()
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        if ld > L:
            tiles[base: base + M * L * ld].view(M * L, ld)[:, L:] = float("nan")
    H = torch.from_numpy(np.random.RandomState(7).randn(M * sum(lengths), d).astype(np.float32)).to(dev)
    want = dense.double() @ H.double().cpu()
    P = P_.cut_planes(H)
    for v in variants:
        setv(v)
        out = P_.propagate_planes_raw(tiles, adj.cross, H, P, lay)
        wv = want if v[2] == "0" else dense.double().t() @ H.double().cpu()     # k-major reads = the transposed tiles
        err = float((out.double().cpu() - wv).abs().max() / wv.abs().max())
        print("%-28s M=%d d=%3d  ver=%s nt=%s km=%s  rel err %.2e %s" % (lengths, M, d, v[0], v[1], v[2], err, "" if err < 1e-5 else "  <-- FAIL"))

# ---- timing, cfg5 B=32, rotating buffer sets
B = 32
l5 = [512] * B
sets = []
for i in range(3):
    g = torch.Generator(device=dev).manual_seed(500 + i)
    adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
    H = torch.randn(6 * sum(l5), 100, device=dev, generator=g)
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


def f_split(i):
    adj, H, P, out = sets[i]
    ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=out)


print("split (fp32 H)            %.1f us" % timeit(f_split))
for rep in range(2):
    for v in variants:
        setv(v)
        for abl in ("0", "1", "64"):
            os.environ["MMDFN_PLANES_ABL"] = abl
            if v[0] == "1" and abl != "0":
                continue
            print("ver=%s nt=%s km=%s abl=%-3s         %.1f us" % (v[0], v[1], v[2], abl, timeit(f_planes)), flush=True)
os.environ["MMDFN_PLANES_ABL"] = "0"
