"""Per-kernel timing of the fused GCN-stack stages and the head at dialogue-graph size (HIP events around 50 launches
captured in a hipGraph).    python tools/bench_stack.py [R]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import _hip  # noqa: E402

R, H, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 5280), 100, 200
dev = "cuda"
t = lambda *s: torch.randn(*s, device=dev)
lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
x, mx, m0, hi, h0, q, m, c, h = t(R, F), t(R, F), t(R, H), t(R, H), t(R, H), t(R, H), t(R, H), t(R, H), t(R, H)
W0, b0, W, Wih, Whh, bsum = t(H, F), t(H), t(2 * H, H), t(4 * H, H) * .1, t(4 * H, H) * .1, t(4 * H)
xd, o1, o2, gates, gmask = t(R, F + H), t(R, H), t(R, H), t(R, 4 * H).sigmoid(), t(R, H)
dG, o3, o4, o5 = t(R, 4 * H), t(R, H), t(R, H), t(R, F)
CASES = {
    "gcn_input_fwd": lambda: lib.mmdfn_gcn_input_fwd(P(x), P(mx), P(W0), P(b0), P(m0), P(xd), P(o1), P(o2), R, F, H, F + H, 2.0, st()),
    "gcn_input_bwd": lambda: lib.mmdfn_gcn_input_bwd(P(q), P(m0), P(h), P(h0), P(W0), P(xd), P(mx), P(o1), P(o5), R, F, H, F + H, 2.0, st()),
    "lstm_gate_fwd": lambda: lib.mmdfn_lstm_gate_fwd(P(q), P(h), P(c), P(Wih), P(Whh), P(bsum), None, P(dG), P(o1), P(o2), R, H, st()),
    "lstm_gate_fwd(first)": lambda: lib.mmdfn_lstm_gate_fwd(P(q), None, None, P(Wih), P(Whh), P(bsum), None, P(dG), P(o1), P(o2), R, H, st()),
    "lstm_gate_bwd": lambda: lib.mmdfn_lstm_gate_bwd(P(gates), P(c), P(h), P(q), P(hi), P(m), P(Wih), P(Whh), P(h0), P(dG), P(o1), P(o2), P(o3), R, H, 1, H, st()),
    "gcnii_layer_fwd": lambda: lib.mmdfn_gcnii_layer_fwd(P(hi), P(h0), P(W), P(q), P(m), P(o1), P(gmask), 0.4, 0.2, R, H, H, 2.0, st()),
    "gcnii_layer_bwd": lambda: lib.mmdfn_gcnii_layer_bwd(P(q), P(gmask), P(W), P(o1), P(o2), P(o3), 0.4, 0.2, R, H, H, 1, st()),
}
for name, fn in CASES.items():
    for _ in range(3):
        assert fn() == 0, name
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    e1.synchronize()
    print("%-22s R=%d  %7.2f us" % (name, R, e0.elapsed_time(e1) / 200 * 1e3), flush=True)
