"""The MFMA form of the GRU recurrence (csrc/gru_mfma.hip) against the one-sequence-per-workgroup kernels, launch time per
(T, rows), plus the forward kernel's timing ablations (tuning build: MMDFN_GRU_MF_ABL 1 no operand loads in the loop, 2 no result
stores, 4 no MFMAs):  python tools/time_gru_mfma.py"""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mm_dfn_amd import _hip
H = 100
dev = "cuda"


def run(T, rows, iters=30):
    gi = torch.randn(T, rows, 600, device=dev)
    whh = [torch.randn(300, 100, device=dev) * 0.1 for _ in range(2)]
    bhh = [torch.randn(300, device=dev) * 0.1 for _ in range(2)]
    y = torch.empty(T, rows, 200, device=dev); g = torch.empty(T, rows, 2, 4, 100, device=dev)
    dy = torch.randn(T, rows, 200, device=dev); dgi = torch.empty(T, rows, 600, device=dev); dgh = torch.empty_like(dgi)
    lib = _hip.lib()

    def f():
        lib.mmdfn_gru_seq_fwd(1, _hip.ptr_array([gi]), _hip.ptr_array(whh), _hip.ptr_array(bhh), _hip.ptr_array([y]), _hip.ptr_array([g]), _hip.int_array([rows]), _hip.int_array([T]), H, _hip.stream())

    def b():
        lib.mmdfn_gru_seq_bwd(1, _hip.ptr_array([dy]), _hip.ptr_array([y]), _hip.ptr_array([g]), _hip.ptr_array(whh), _hip.ptr_array([dgi]), _hip.ptr_array([dgh]), _hip.int_array([rows]), _hip.int_array([T]), H, _hip.stream())
    out = []
    for fn in (f, b):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    return out


for T, rows in ((33, 960), (33, 16)):
    for mode, abl in (("scalar", 0), ("mfma", 0), ("mfma", 1), ("mfma", 2), ("mfma", 3), ("mfma", 4), ("mfma", 7), ("mfma", 8), ("mfma", 0), ("mfma", 8)):
        os.environ["MMDFN_GRU_MFMA_MIN"] = "100000000" if mode == "scalar" else "0"
        os.environ["MMDFN_GRU_MF_ABL"] = str(abl)
        f, b = run(T, rows)
        print("T=%d rows=%d %-6s abl %d: fwd %.1f us (%.2f/step)  bwd %.1f us (%.2f/step)" % (T, rows, mode, abl, f, f / T, b, b / T), flush=True)
