"""Timing ablations of the forward GRU recurrence (tuning build, MMDFN_GRU_ABL bits: 1 no matvec, 2 no transcendental gate
math, 4 no block traffic (stash / flush / prefetch of the 8-step blocks), 8 no deferred result writes, 16 no per-step
barrier; wrong results).  cfg2 shapes: 16 + 64 sequences, T = 110."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import _hip

H = 100
shapes = [(110, 16), (110, 64)]
gi = [torch.randn(T, R, 6 * H, device="cuda") for T, R in shapes]
whh = [torch.randn(3 * H, H, device="cuda") * 0.1 for _ in range(4)]
bhh = [torch.randn(3 * H, device="cuda") * 0.1 for _ in range(4)]
ys = [torch.empty(T, R, 2 * H, device="cuda") for T, R in shapes]
gates = [torch.empty(T, R, 2, 4, H, device="cuda") for T, R in shapes]
lib = _hip.lib()


def run():
    rc = lib.mmdfn_gru_seq_fwd(2, _hip.ptr_array(gi), _hip.ptr_array(whh), _hip.ptr_array(bhh), _hip.ptr_array(ys),
                               _hip.ptr_array(gates), _hip.int_array([16, 64]), _hip.int_array([110, 110]), H, _hip.stream())
    assert rc == 0


variants = [("io", "1", "0", 0), ("io+scalar-fma", "1", "1", 0), ("4-wave", "0", "0", 0), ("io", "1", "0", 0), ("io+scalar-fma", "1", "1", 0)]
if len(sys.argv) > 1 and sys.argv[1] == "fold":
    # round 6: upper bound of the constant-folding lead (ABL 64: no bias adds, no log2(e) multiplies on the step's chain)
    variants = [("io", "1", "0", 0), ("io, folded constants (bound)", "1", "0", 64)] * 4
if len(sys.argv) > 1 and sys.argv[1] == "abl":
    variants = [("io abl %d" % a, "1", "0", a) for a in (0, 1, 2, 4, 8, 16, 3, 11, 23, 31, 0)]
for name, io, sf, abl in variants:
    os.environ["MMDFN_GRU_IO"], os.environ["MMDFN_GRU_SCALAR_FMA"] = io, sf
    if abl:
        os.environ["MMDFN_GRU_ABL"] = str(abl)
    else:
        os.environ.pop("MMDFN_GRU_ABL", None)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); e1.synchronize()
    print("%-30s: %6.1f us per launch, %.3f us per step" % (name, e0.elapsed_time(e1) * 1e3 / 50, e0.elapsed_time(e1) * 1e3 / 50 / 110), flush=True)
