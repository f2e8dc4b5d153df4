"""Fused single-node GCN stack vs the op-by-op path (bf16-piece GEMMs at scale) as a function of the row count:
where should gcn_stack.ROW_LIMIT sit?    python tools/time_stack_paths.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import GCNII_lyc, gcn_stack, ops  # noqa: E402
from mm_dfn_amd.graphs import CapturedStep  # noqa: E402

dev = "cuda"
for B, L, M, nl in [tuple(int(v) for v in c.split("x")) for c in os.environ.get("CASES", "16x110x3x2,32x110x3x2,48x110x3x2,4x512x6x8,8x512x6x8,16x512x6x8,32x512x6x8").split(",")]:
    lengths = [L] * B
    N = L * B
    feats = torch.randn(M, N, 200, device=dev, requires_grad=True)
    for limit in (1 << 30, 0):
        gcn_stack.ROW_LIMIT = limit
        net = GCNII_lyc(nfeat=200, nlayers=nl, nhidden=100, nclass=6, dropout=0.5, lamda=0.5, alpha=0.2, variant=True,
                        return_feature=True, use_residue=True, reason_flag=True).to(dev).train()

        def step():
            adj = ops.build_adjacency(feats, lengths)
            out = net(adj.stacked_feats.reshape(M * N, 200), lengths, None, adj)
            out.sum().backward()
            return out.sum()

        cap = CapturedStep(net, step, warmup=1)
        for _ in range(3):
            cap.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            cap.replay()
        torch.cuda.synchronize()
        print("rows %6d (B=%d L=%d M=%d, %d layers)  %-10s %8.3f ms" % (M * N, B, L, M, nl, "fused" if limit else "op-by-op",
                                                                        (time.perf_counter() - t0) / 10 * 1e3), flush=True)
        feats.grad = None
