"""Timing ablations of the forward GRU recurrence (tuning build, MMDFN_GRU_ABL bits: 1 no FMAs, 2 no transcendentals,
4 no barrier, 8 no staged-operand reads / result writes).  cfg2 shapes: 16 + 64 sequences, T = 110."""
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


for abl in (0, 1, 2, 3, 4, 7, 8, 11, 15, 0):
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
    print("abl %2d: %6.1f us per launch, %.3f us per step" % (abl, e0.elapsed_time(e1) * 1e3 / 50, e0.elapsed_time(e1) * 1e3 / 50 / 110), flush=True)
