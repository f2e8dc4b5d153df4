"""Accuracy + timing of the producer / consumer bf16-piece propagate (kernel "10": tools/k6_pc/bin/libk6pc.so, outside the
product library since round 5) against the bf16-piece kernel of rounds 1-3 (MMDFN_PROP_CFG=8), the exact-f32 kernels (9) and
an fp64 dense product.

    bash tools/k6_pc/build.sh && python tools/k6_pc/check_pc.py    # correctness cases + cfg5 timing (rotating buffer sets, captured graph)
"""
import os

os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
for sub in ("tests", "oracle"):   # tests/util.py builds the dense fp64 reference (test infrastructure only)
    sys.path.insert(0, os.path.join(ROOT, sub))
from mm_dfn_amd import ops  # noqa: E402
import pc_ops  # noqa: E402
from util import random_block_adjacency  # noqa: E402

DEV = "cuda"
CASES = [
    ([5], 3, 100), ([7, 3, 1], 3, 100), ([32, 33, 31, 64], 3, 100), ([110, 64, 65, 27], 3, 100),
    ([129, 127, 128, 200], 2, 100), ([260, 40], 6, 64), ([513], 3, 100), ([140, 77], 3, 112),
    ([512] * 3, 6, 100), ([300, 512, 17, 129, 1, 255, 256, 257, 400], 4, 100), ([161] * 40, 5, 36),
]


def run(cfg, tiles, cross, H, lay, out=None):
    if int(cfg) == 10:
        return pc_ops.propagate_pc(tiles, cross, H, lay, out=out)
    os.environ["MMDFN_PROP_CFG"] = str(cfg)
    return ops.propagate_raw(tiles, cross, H, lay, out=out)


bad = 0
for lengths, M, d in CASES:
    adj, dense, _, _ = random_block_adjacency(13, lengths, M, DEV)
    lay = adj.layout
    tiles = adj.tiles.clone()
    for i, L in enumerate(lengths):
        ld = int(lay.ld_host[i]); base = int(lay.tile_base_host[i])
        if ld > L:
            tiles[base: base + M * L * ld].view(M * L, ld)[:, L:] = float("nan")
    rs = np.random.RandomState(7)
    H = torch.from_numpy(rs.randn(M * sum(lengths), d).astype(np.float32)).to(DEV)
    want = (dense.double() @ H.double().cpu())
    o_pc = run(10, tiles, adj.cross, H, lay)
    o_f32 = run(9, tiles, adj.cross, H, lay)
    torch.cuda.synchronize()
    e_pc = float((o_pc.double().cpu() - want).abs().max())
    e_f32 = float((o_f32.double().cpu() - want).abs().max())
    ok = e_pc <= 4 * e_f32 + 1e-7
    bad += not ok
    print("%-44s M=%d d=%3d  |pc-f64| %.3g  |f32-f64| %.3g  %s" % (lengths if len(lengths) < 10 else "%d x %d" % (len(lengths), lengths[0]),
                                                                 M, d, e_pc, e_f32, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)

# ---- timing at BASELINE cfg5 (B = 32 and 8), rotating buffer sets, captured graph
for B in (32, 8):
    M, d, L = 6, 100, 512
    lengths = [L] * B
    N = B * L
    sets = []
    nset = 3 if B == 32 else 6
    for s in range(nset):
        adj = ops.build_adjacency(torch.randn(M, N, 200, device=DEV), lengths)
        sets.append((adj.tiles, adj.cross, torch.randn(M * N, d, device=DEV), torch.empty(M * N, d, device=DEV), adj.layout))
    by = sets[0][4].propagate_bytes(d)
    runs = [(8, 0), (10, 0), (8, 0), (10, 0)] + [(10, int(a)) for a in os.environ.get("PC_ABLS", "").split(",") if a]
    for cfg, abl in runs:
        os.environ["MMDFN_PC_ABL"] = str(abl)
        for t, c, h, o, lay in sets:
            run(cfg, t, c, h, lay, out=o)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for rep in range(7):
                    for t, c, h, o, lay in sets:
                        run(cfg, t, c, h, lay, out=o)
            for _ in range(12):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                g.replay()
            e1.record(st)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * 7 * nset)
        if abl & 64:
            t, c, h, o, lay = sets[0]
            run(10, t, c, h, lay, out=o)
            torch.cuda.synchronize()
            nwg = min(256, B * M * 4)
            st_ = o.view(-1)[: nwg * 16].view(nwg, 16).double().mean(0).cpu().numpy()
            n = max(st_[2], 1)
            print("   stamps (mean per workgroup, cycles per chunk period): consumer chunk %.0f barrier %.0f | B-producer issue %.0f "
                  "wait %.0f cut %.0f epi %.0f barrier %.0f | A-producer issue %.0f wait %.0f cut %.0f epi %.0f barrier %.0f | "
                  "kernel %.0f cycles in %.2f us -> %.2f GHz (chunks %d)"
                  % (st_[0] / n, st_[1] / n, *(st_[5:15] / n), st_[3], st_[4] / 100.0, st_[3] / (st_[4] * 10.0), n))
        print("cfg5 B=%d kernel %d abl %2d: %.1f us/launch  %.3f of 8 TB/s" % (B, cfg, abl, us, by / (us * 1e-6) / 8e12), flush=True)
