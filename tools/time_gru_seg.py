"""Segmented vs plain recurrence launches on the same rows (what the bookkeeping costs): python tools/time_gru_seg.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from mm_dfn_amd import _hip
H = 100
dev = "cuda"
lib = _hip.lib()


def timeit(fn, iters=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(T, B, P, nm, ctx_rows):
    rows = nm * B * P
    rs = np.random.RandomState(0)
    spk = rs.randint(0, P, size=(T, B))
    rank = -np.ones((T, B, P), np.int32)
    for b in range(B):
        for p in range(P):
            idx = np.nonzero(spk[:, b] == p)[0]
            rank[idx, b, p] = np.arange(len(idx))
    rank = torch.from_numpy(rank).to(dev)
    groups = [(ctx_rows, None), (rows, rank)]
    gi = [torch.randn(T, r, 600, device=dev) for r, _ in groups]
    whh = [torch.randn(300, 100, device=dev) * 0.1 for _ in range(4)]
    bhh = [torch.randn(300, device=dev) * 0.1 for _ in range(4)]
    y = [torch.zeros(T, r, 200, device=dev) for r, _ in groups]
    g = [torch.zeros(T, r, 2, 4, 100, device=dev) for r, _ in groups]
    dy = [torch.randn(T, r, 200, device=dev) for r, _ in groups]
    dgi = [torch.empty(T, r, 600, device=dev) for r, _ in groups]
    dgh = [torch.empty(T, r, 600, device=dev) for r, _ in groups]
    n = 2
    R = _hip.int_array([r for r, _ in groups])
    Ts = _hip.int_array([T, T])
    pa = _hip.ptr_array

    def legacy_f():
        lib.mmdfn_gru_seq_fwd(n, pa(gi), pa(whh), pa(bhh), pa(y), pa(g), R, Ts, H, _hip.stream())

    def legacy_b():
        lib.mmdfn_gru_seq_bwd(n, pa(dy), pa(y), pa(g), pa(whh), pa(dgi), pa(dgh), R, Ts, H, _hip.stream())

    def seg(use_rank, tdir):
        rk = pa([None, rank if use_rank else None])
        Pa, BPa, td = _hip.int_array([1, P]), _hip.int_array([1, B * P]), _hip.int_array([-1, tdir])

        def f():
            rc = lib.mmdfn_gru_seq_fwd_seg(n, pa(gi), pa(whh), pa(bhh), pa(y), pa(g), R, Ts, H, rk, Pa, BPa, td, pa([None, None]), _hip.stream())
            assert rc == 0

        def b():
            rc = lib.mmdfn_gru_seq_bwd_seg(n, pa(dy), pa(y), pa(g), pa(whh), pa(dgi), pa(dgh), R, Ts, H, rk, Pa, BPa, td, pa([None, None]), pa([None, None]), _hip.stream())
            assert rc == 0
        return f, b

    print("T=%d ctx=%d party=%d (B=%d P=%d nm=%d): plain chains %d" % (T, ctx_rows, rows, B, P, nm, 2 * (rows + ctx_rows)))
    print("  plain            fwd %6.1f  bwd %6.1f" % (timeit(legacy_f), timeit(legacy_b)))
    for name, ur, td in (("seg, no rank    ", False, -1), ("seg, rank, full ", True, -1), ("seg, fwd merged ", True, 0)):
        f, b = seg(ur, td)
        print("  %s fwd %6.1f  bwd %6.1f" % (name, timeit(f), timeit(b)))


case(110, 32, 2, 2, 32)      # cfg4 shapes
case(110, 16, 2, 2, 16)      # cfg2 shapes
case(33, 32, 9, 2, 32)       # cfg3 shapes
