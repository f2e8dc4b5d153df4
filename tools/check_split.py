"""Accuracy + timing of the bf16-piece propagate (MMDFN_PROP_CFG=8) against the exact-f32 kernels and an
fp64 dense product.  Run under tools/prof_stats.sh for kernel durations."""
import os

os.environ["MMDFN_TUNING_LIB"] = "1"   # the MMDFN_* switches below exist only in the -DMMDFN_TUNING build
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops, synthetic  # noqa: E402
from bench_kernels import WORKLOADS  # noqa: E402

names = sys.argv[1].split(",")
for name in names:
    w = WORKLOADS[name]
    rs = np.random.RandomState(1)
    lengths = w["lengths"] or synthetic.make_lengths(rs, w["B"], w["L"], True, min_len=3)
    M, d = w["M"], w["d"]
    N = sum(lengths)
    adj = ops.build_adjacency(torch.randn(M, N, 200, device="cuda"), lengths)
    H = torch.randn(M * N, d, device="cuda")
    os.environ.pop("MMDFN_PROP_CFG", None)
    for _ in range(10):
        ref = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=ref)
    e1.record()
    torch.cuda.synchronize()
    print("%s f32-mfma kernel: %.1f us/launch" % (name, e0.elapsed_time(e1) * 10))
    exact = None
    if M * N <= 12000:
        exact = (adj.to_dense().double() @ H.double())
    for occ in os.environ.get("OCCS", "2").split(","):
        os.environ["MMDFN_PROP_CFG"] = "8"
        os.environ["MMDFN_SPLIT_OCC"] = occ
        for _ in range(10):
            out = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
        torch.cuda.synchronize()
        msg = "%s occ%s max|split - f32mfma| = %.3g" % (name, occ, float((out - ref).abs().max()))
        if exact is not None:
            msg += "  |split-f64| = %.3g  |f32mfma-f64| = %.3g" % (float((out.double() - exact).abs().max()),
                                                                  float((ref.double() - exact).abs().max()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        msg += "  %.1f us/launch  %.1f%% of 8 TB/s" % (us, 100 * adj.layout.propagate_bytes(d) / (us * 1e-6) / 8e12)
        print(msg, flush=True)
