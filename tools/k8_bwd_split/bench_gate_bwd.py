"""K8 backward, exact-f32 producer / consumer kernel vs the bf16-piece form (tuning build): python tools/bench_gate_bwd.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mm_dfn_amd import _hip
_hip.set_tuning(True)
P, st = _hip.ptr, _hip.stream
H = 100
ABLS = [int(x) for x in os.environ.get("ABLS", "0").split(",")]
for R in (98304,) if len(ABLS) > 1 else (24576, 49152, 98304):
    sets = []
    for _ in range(3):
        g = torch.rand(R, 4 * H, device="cuda"); c, cn = torch.randn(R, H, device="cuda"), torch.randn(R, H, device="cuda")
        dha, dhb, dcn, dres = (torch.randn(R, H, device="cuda") for _ in range(4))
        out = [torch.empty(R, 4 * H, device="cuda")] + [torch.empty(R, H, device="cuda") for _ in range(3)]
        sets.append((g, c, cn, dha, dhb, dcn, dres, out))
    Wih, Whh = torch.randn(4 * H, H, device="cuda") * 0.2, torch.randn(4 * H, H, device="cuda") * 0.2
    for mode, abl in [("0", 0)] + [("1", a) for a in ABLS]:
        os.environ["MMDFN_GATE_BWD_SPLIT"] = mode
        os.environ["MMDFN_GATE_BWD_ABL"] = str(abl)
        def run():
            for g, c, cn, dha, dhb, dcn, dres, o in sets:
                assert _hip.lib().mmdfn_lstm_gate_bwd(P(g), P(c), P(cn), P(dha), P(dhb), P(dcn), P(Wih), P(Whh), P(dres), P(o[0]), P(o[1]),
                                                      P(o[2]), P(o[3]), R, H, 1, H, st()) == 0
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print("R=%6d abl %2d  %s  %.1f us  %.1f TFLOP/s useful  %.2f TB/s of algorithmic traffic" % (R, abl, "bf16-piece" if mode == "1" else "exact-f32 ", us, 2 * R * 4 * H * 2 * H / us / 1e6, R * 16 * H * 4 / us / 1e6))
