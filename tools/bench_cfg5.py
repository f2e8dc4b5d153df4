"""BASELINE.json config 5: synthetic long-dialogue stress (L=512, 6 modality streams x 512-d inputs,
8 GCN layers) -- graph dynamic-fusion stack only (the reference's encoders are trimodal, model_mm.py:97-106;
M=6 is beyond what the reference can run).  Runs  projection 512->200 per stream ->
adjacency build (K5) -> GCNII_lyc with 8 layers (K6/K7/K8) -> sum loss -> backward  and prints one JSON line
with utterances/s and the K6 roofline numbers for this workload.

    python tools/bench_cfg5.py [--B 8] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import GCNII_lyc, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--L", type=int, default=512)
    ap.add_argument("--M", type=int, default=6)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    lengths = [a.L] * a.B
    N = a.L * a.B
    proj = torch.nn.ModuleList([torch.nn.Linear(512, 200) for _ in range(a.M)]).to(dev)
    net = GCNII_lyc(nfeat=200, nlayers=a.layers, nhidden=100, nclass=6, dropout=0.5, lamda=0.5, alpha=0.2, variant=True,
                    return_feature=True, use_residue=True, reason_flag=True).to(dev).train()
    xs = [torch.randn(N, 512, device=dev) for _ in range(a.M)]
    params = list(proj.parameters()) + list(net.parameters())

    def step():
        for p in params:
            p.grad = None
        feats = torch.stack([proj[m](xs[m]) for m in range(a.M)], 0)          # (M, N, 200)
        adj = ops.build_adjacency(feats, lengths)
        out = net(feats.reshape(a.M * N, 200), lengths, None, adj)
        out.sum().backward()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps

    feats = torch.randn(a.M, N, 200, device=dev)
    adj = ops.build_adjacency(feats, lengths)
    H = torch.randn(a.M * N, 100, device=dev)
    for _ in range(3):
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    iters = 20
    with torch.cuda.graph(g):
        for _ in range(iters):
            ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    by = adj.layout.propagate_bytes(100)
    fl = adj.layout.propagate_flops(100)
    print(json.dumps({"workload": "cfg5: B=%d dialogues, L=%d, M=%d streams x 512-d, %d GCN layers (graph stack only)"
                                  % (a.B, a.L, a.M, a.layers),
                      "utterances_per_s": N / dt, "ms_per_step": dt * 1e3,
                      "propagate_fwd": {"avg_launch_us": us, "algorithmic_bytes": by, "GBps": by / us / 1e3,
                                        "hbm_frac": by / us / 1e3 / 8000.0, "TFLOPs": fl / us / 1e6,
                                        "f32_mfma_frac": fl / us / 1e6 / 157.3}}), flush=True)


if __name__ == "__main__":
    main()
