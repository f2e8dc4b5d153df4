"""K8 (LSTM cell of the GCN stack) forward: exact-f32 producer / consumer kernel (MMDFN_GATE_SPLIT=0) against the bf16-piece
form (MMDFN_GATE_SPLIT=1, csrc/lstm_gate_split.hip) at cfg5 row counts; rotating operand sets, captured graph.
    python tools/bench_gate.py [rows ...]"""
import os
import sys

os.environ["MMDFN_TUNING_LIB"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import _hip  # noqa: E402

H = 100
P, st = _hip.ptr, _hip.stream
for R in [int(x) for x in sys.argv[1:]] or [24576, 98304, 12288]:
    nset = 4
    sets = [dict(q=torch.randn(R, H, device="cuda"), h=torch.randn(R, H, device="cuda"), c=torch.randn(R, H, device="cuda"),
                 gates=torch.empty(R, 4 * H, device="cuda"), ho=torch.empty(R, H, device="cuda"), co=torch.empty(R, H, device="cuda"))
            for _ in range(nset)]
    Wih, Whh = torch.randn(4 * H, H, device="cuda") * 0.2, torch.randn(4 * H, H, device="cuda") * 0.2
    b1, b2 = torch.randn(4 * H, device="cuda"), torch.randn(4 * H, device="cuda")
    for mode in ("0", "1", "0", "1"):
        os.environ["MMDFN_GATE_SPLIT"] = mode

        def run():
            for s in sets:
                assert _hip.lib().mmdfn_lstm_gate_fwd(P(s["q"]), P(s["h"]), P(s["c"]), P(Wih), P(Whh), P(b1), P(b2), P(s["gates"]),
                                                      P(s["ho"]), P(s["co"]), R, H, st()) == 0
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                for _ in range(5):
                    run()
            for _ in range(5):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                g.replay()
            e1.record(stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (25 * nset)
        print("rows %6d  %-22s %7.1f us  %6.1f TFLOP/s (fp32-equivalent)" % (R, "bf16-piece" if mode == "1" else "exact-f32 (ws)", us,
                                                                           2.0 * R * 2 * H * 4 * H / us / 1e6), flush=True)
