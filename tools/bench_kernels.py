"""Kernel micro-benchmarks (HIP-event timing on the launch stream) for the graph kernels.

    python tools/bench_kernels.py [--what propagate|adj|all] [--iters 50]

Prints one JSON line per (kernel, workload) with algorithmic bytes/flops and the
achieved GB/s and TFLOP/s.  Workloads: cfg2 (B=16, L=110), cfg4 shard (B=32, L=110), cfg3 (MELD-like
ragged), cfg5 (L=512, M=6, B=8/32, d=100 and d=512).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops, synthetic  # noqa: E402
from mm_dfn_amd.layout import DialogueLayout  # noqa: E402

DEV = "cuda"
WORKLOADS = {
    "cfg2": dict(lengths=[110] * 16, M=3, d=100),
    "cfg2_ragged": dict(lengths=None, B=16, L=110, M=3, d=100),
    "cfg4_shard": dict(lengths=[110] * 32, M=3, d=100),
    "cfg3_meld": dict(lengths=None, B=32, L=33, M=3, d=100),
    "cfg5_b8_d100": dict(lengths=[512] * 8, M=6, d=100),
    "cfg5_b32_d100": dict(lengths=[512] * 32, M=6, d=100),
    "cfg5_b8_d512": dict(lengths=[512] * 8, M=6, d=512),
    "mid_L256_b16": dict(lengths=[256] * 16, M=3, d=100),
    "mid_L128_b64": dict(lengths=[128] * 64, M=3, d=100),
    "mid_L200_b32": dict(lengths=[200] * 32, M=3, d=100),
    "mid_L384_b8": dict(lengths=[384] * 8, M=3, d=100),
}


def timeit(fn, iters):
    for _ in range(5):
        fn()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    for name, w in WORKLOADS.items():
        if a.only and a.only not in name:
            continue
        rs = np.random.RandomState(1)
        lengths = w["lengths"] or synthetic.make_lengths(rs, w["B"], w["L"], True, min_len=3)
        M, d = w["M"], w["d"]
        N = sum(lengths)
        feats = torch.randn(M, N, 200, device=DEV)
        adj = ops.build_adjacency(feats, lengths)
        lay = adj.layout
        H = torch.randn(M * N, d, device=DEV)
        dO = torch.randn(M * N, d, device=DEV)
        if a.what in ("all", "propagate"):
            t = timeit(lambda: ops.propagate_raw(adj.tiles, adj.cross, H, lay), a.iters)
            by, fl = lay.propagate_bytes(d), lay.propagate_flops(d)
            print(json.dumps({"kernel": "propagate_fwd", "workload": name, "us": t * 1e6, "alg_bytes": by,
                              "GBps": by / t / 1e9, "hbm_frac": by / t / 8e12, "TFLOPs": fl / t / 1e12}), flush=True)
            t = timeit(lambda: ops.tile_outer_raw(dO, H, lay), a.iters)
            by2 = 4 * lay.nnz + 8 * M * N * d
            print(json.dumps({"kernel": "tile_outer(dA)", "workload": name, "us": t * 1e6, "alg_bytes": by2,
                              "GBps": by2 / t / 1e9, "TFLOPs": fl / t / 1e12}), flush=True)
        if a.what in ("all", "adj"):
            t = timeit(lambda: ops.build_adjacency(feats, lengths), a.iters)
            by3 = 4 * M * N * 200 + 4 * lay.nnz
            print(json.dumps({"kernel": "adj_build(4 kernels)", "workload": name, "us": t * 1e6, "alg_bytes": by3,
                              "GBps": by3 / t / 1e9}), flush=True)


if __name__ == "__main__":
    main()
