"""K7 forward / backward and K8 backward: the plane kernels (csrc/gcn_planes.hip) against the exact-f32 kernels of csrc/gcn_stack.hip
at the stack's row counts (cfg2 5 280, cfg4 10 560, cfg5 B=8 24 576, cfg5 B=32 98 304); captured launches, HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_dfn_amd import _hip, ops  # noqa: E402

dev = "cuda"
lib, P, st = _hip.lib(), _hip.ptr, _hip.stream
H = 100


def timeit(fn, n=10, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


for R in (5280, 10560, 24576, 98304):
    r = lambda *s: torch.randn(*s, device=dev)
    hi, h0, q, m = r(R, H), r(R, H), r(R, H), (torch.rand(R, H, device=dev) > 0.5).float()
    W = torch.nn.Parameter(r(2 * H, H) * 0.1)
    out, gmask = torch.empty(R, H, device=dev), torch.empty(R, H, device=dev)
    pf, pb = ops.weight_planes(W, mode=1), ops.weight_planes(W, mode=0)
    t_f0 = timeit(lambda: lib.mmdfn_gcnii_layer_fwd(P(hi), P(h0), P(W), P(q), P(m), P(out), P(gmask), 0.4, 0.2, R, H, H, 2.0, st()))
    t_f1 = timeit(lambda: lib.mmdfn_gcnii_layer_fwd_planes(P(hi), P(h0), P(pf.buf), P(q), P(m), P(out), P(gmask), 0.4, 0.2, R, H, H, 2.0, st()))
    dout, dP, dhi, dh0 = r(R, H), torch.empty(R, H, device=dev), torch.empty(R, H, device=dev), torch.zeros(R, H, device=dev)
    t_b0 = timeit(lambda: lib.mmdfn_gcnii_layer_bwd(P(dout), P(gmask), P(W), P(dP), P(dhi), P(dh0), 0.4, 0.2, R, H, H, 1, st()))
    t_b1 = timeit(lambda: lib.mmdfn_gcnii_layer_bwd_planes(P(dout), P(gmask), P(pb.buf), P(dP), P(dhi), P(dh0), 0.4, 0.2, R, H, H, 1, H, st()))
    gates = torch.sigmoid(r(R, 4 * H))
    c_prev, c_new, dh_a, dh_b, dc_next, dres = r(R, H), r(R, H), r(R, H), r(R, H), r(R, H), r(R, H)
    w_ih, w_hh = torch.nn.Parameter(r(4 * H, H) * 0.1), torch.nn.Parameter(r(4 * H, H) * 0.1)
    dG, dcp, dq, dhp = torch.empty(R, 4 * H, device=dev), torch.empty(R, H, device=dev), torch.empty(R, H, device=dev), torch.empty(R, H, device=dev)
    pg = ops.weight_planes(w_ih, w_hh, mode=3)
    t_g0 = timeit(lambda: lib.mmdfn_lstm_gate_bwd(P(gates), P(c_prev), P(c_new), P(dh_a), P(dh_b), P(dc_next), P(w_ih), P(w_hh), P(dres),
                                                  P(dG), P(dcp), P(dq), P(dhp), R, H, 1, H, st()))
    t_g1 = timeit(lambda: lib.mmdfn_lstm_gate_bwd_planes(P(gates), P(c_prev), P(c_new), P(dh_a), P(dh_b), P(dc_next), P(pg.buf), P(dres),
                                                         P(dG), P(dcp), P(dq), P(dhp), R, H, 1, H, st()))
    print("R=%6d  K7 fwd: exact %6.1f planes %6.1f us | K7 bwd: exact %6.1f planes %6.1f us | K8 bwd: exact %6.1f planes %6.1f us"
          % (R, t_f0, t_f1, t_b0, t_b1, t_g0, t_g1), flush=True)
