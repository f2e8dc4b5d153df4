"""K7 (gcnii_layer fwd / bwd): 4-wave kernels vs the producer / consumer forms (tuning build): python tools/bench_layer.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mm_dfn_amd import _hip
_hip.set_tuning(True)
P, st = _hip.ptr, _hip.stream
H = 100
for R in (5280, 24576, 98304):
    sets = []
    for _ in range(3):
        hi, h0, q = (torch.randn(R, H, device="cuda") for _ in range(3))
        m = (torch.rand(R, H, device="cuda") > 0.5).float()
        out, gm, dP, dhi, dh0 = (torch.empty(R, H, device="cuda") for _ in range(5))
        dout = torch.randn(R, H, device="cuda")
        sets.append((hi, h0, q, m, out, gm, dP, dhi, dh0, dout))
    W = torch.randn(2 * H, H, device="cuda") * 0.1
    for mode in ("0", "1"):
        os.environ["MMDFN_LAYER_WS"] = mode
        def fwd():
            for hi, h0, q, m, out, gm, dP, dhi, dh0, dout in sets:
                assert _hip.lib().mmdfn_gcnii_layer_fwd(P(hi), P(h0), P(W), P(q), P(m), P(out), P(gm), 0.405, 0.2, R, H, H, 2.0, st()) == 0
        def bwd():
            for hi, h0, q, m, out, gm, dP, dhi, dh0, dout in sets:
                assert _hip.lib().mmdfn_gcnii_layer_bwd(P(dout), P(gm), P(W), P(dP), P(dhi), P(dh0), 0.405, 0.2, R, H, H, 1, st()) == 0
        res = []
        for fn in (fwd, bwd):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 30 * 1e3)
        print("R=%6d  %s  fwd %.1f us (%.2f TB/s of 6 streams)  bwd %.1f us" % (R, "producer/consumer" if mode == "1" else "4-wave          ", res[0], 6 * R * H * 4 / res[0] / 1e6, res[1]))
