"""Compile-time timing ablations of the producer / consumer K8 backward kernel (tuning build, MMDFN_GATE_ABL bits: 1 no MFMA, 2 no operand
loads, 4 no gate math, 8 no result stores) at R rows, H = 100.      python tools/ablate_gate_bwd.py [R]"""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import _hip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
H = 100
lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
t = lambda *s: torch.randn(*s, device="cuda")
gates, c, c_out, dh_a, dh_b, dc_n = torch.sigmoid(t(R, 4 * H)), t(R, H), t(R, H), t(R, H), t(R, H), t(R, H)
Wih, Whh, dres = t(4 * H, H), t(4 * H, H), t(R, H)
dG, dq, dcp, dhp = t(R, 4 * H), t(R, H), t(R, H), t(R, H)
def run():
    assert lib.mmdfn_lstm_gate_bwd(P(gates), P(c), P(c_out), P(dh_a), P(dh_b), P(dc_n), P(Wih), P(Whh), P(dres), P(dG), P(dcp),
                                   P(dq), P(dhp), R, H, 1, H, st()) == 0
def gtime(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3
os.environ["MMDFN_GATE_WS"] = "0"
print("R = %d: four-wave kernel %.1f us" % (R, gtime(run)))
for name, env in (("producer / consumer kernel", dict(MMDFN_GATE_WS="1")),):
    os.environ.update(env)
    print(" ", name)
    for abl in (0, 1, 2, 4, 8, 3, 6, 12, 7, 14, 15):
        os.environ["MMDFN_GATE_ABL"] = str(abl)
        print("    abl %2d: %.1f us" % (abl, gtime(run)), flush=True)
    os.environ.pop("MMDFN_GATE_ABL")
