"""K8 backward (mmdfn_lstm_gate_bwd): the producer / consumer kernel with four column blocks against its WIDE form (one column
block per product, the second tile of every consumer wave on weight fragments held in registers), rotating operand sets,
captured graph:   python tools/bench_gate_bwd.py [rows ...]      (tuning build: MMDFN_GATE_BWD_WIDE=0|1)"""
import os
import sys

os.environ["MMDFN_TUNING_LIB"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import _hip  # noqa: E402

H = 100
P, st = _hip.ptr, _hip.stream
for R in [int(x) for x in sys.argv[1:]] or [98304, 24576, 5280]:
    nset = 3
    r = lambda *s: torch.randn(*s, device="cuda")
    sets = [dict(gates=torch.rand(R, 4 * H, device="cuda"), c=r(R, H), cn=r(R, H), dha=r(R, H), dhb=r(R, H), dcn=r(R, H), dres=r(R, H),
                 dG=torch.empty(R, 4 * H, device="cuda"), dcp=torch.empty(R, H, device="cuda"), dq=torch.empty(R, H, device="cuda"),
                 dhp=torch.empty(R, H, device="cuda")) for _ in range(nset)]
    Wih, Whh = r(4 * H, H) * 0.2, r(4 * H, H) * 0.2
    outs = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["MMDFN_GATE_BWD_WIDE"] = mode

        def run():
            for s in sets:
                assert _hip.lib().mmdfn_lstm_gate_bwd(P(s["gates"]), P(s["c"]), P(s["cn"]), P(s["dha"]), P(s["dhb"]), P(s["dcn"]), P(Wih),
                                                      P(Whh), P(s["dres"]), P(s["dG"]), P(s["dcp"]), P(s["dq"]), P(s["dhp"]), R, H, 1, H,
                                                      st()) == 0
        run()
        torch.cuda.synchronize()
        outs[mode] = [sets[0][k].clone() for k in ("dG", "dcp", "dq", "dhp")]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(3):
                run()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        e1.synchronize()
        print("rows %6d wide=%s  %.1f us per launch" % (R, mode, e0.elapsed_time(e1) / (5 * 3 * nset) * 1e3), flush=True)
    d = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(outs["0"], outs["1"]))
    print("rows %6d max rel difference between the two forms %.2e" % (R, d), flush=True)
