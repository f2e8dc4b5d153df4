"""Parity (fp64) and time of the weight-streaming projection experiment against the product kernels.
    bash tools/proj_stream/build.sh && python tools/proj_stream/check_proj_stream.py"""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mm_dfn_amd import ops, _hip  # noqa: E402

L = ctypes.CDLL(os.path.join(HERE, "bin", "libprojstream.so"))
L.ps_planes_bytes.restype = ctypes.c_int64
L.ps_planes_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
L.ps_cut_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
L.ps_project.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
P, st = _hip.ptr, _hip.stream


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for R, K, N in ((7040, 200, 600), (1760, 200, 600), (3520, 200, 600), (14080, 200, 600), (19008, 200, 600), (5280, 200, 100),
                (7040, 100, 600), (1760, 512, 200), (1000, 200, 604), (33, 12, 40)):
    xs = [torch.randn(R, K, device="cuda") for _ in range(4)]
    W = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda")
    planes = torch.empty(int(L.ps_planes_bytes(N, K)), dtype=torch.uint8, device="cuda")
    ys = [torch.empty(R, N, device="cuda") for _ in range(4)]
    assert L.ps_cut_weight(P(W), P(planes), N, K, K, 0, st()) == 0
    rc = L.ps_project(P(xs[0]), P(planes), P(b), P(ys[0]), R, K, N, K, N, 0, st())
    if rc != 0:
        print("R=%d K=%d N=%d: not covered (rc %d)" % (R, K, N, rc)); continue
    torch.cuda.synchronize()
    want = xs[0].double() @ W.double().t() + b.double()
    ref = ops.linear_group_raw([dict(x=xs[0], w=W, b=b)], 0)[0]
    e_new = float((ys[0].double() - want).abs().max() / want.abs().max())
    e_ref = float((ref.double() - want).abs().max() / want.abs().max())
    it = [0]

    def new():
        i = it[0] % 4; it[0] += 1
        L.ps_project(P(xs[i]), P(planes), P(b), P(ys[i]), R, K, N, K, N, 0, st())

    def cut():
        L.ps_cut_weight(P(W), P(planes), N, K, K, 0, st())

    def many():
        i = it[0] % 4; it[0] += 1
        ops.linear_raw(xs[i], W, b, 0)

    def few():
        i = it[0] % 4; it[0] += 1
        ops.linear_group_raw([dict(x=xs[i], w=W, b=b)], 0)
    tn, tc, tm, tf = timeit(new), timeit(cut), timeit(many), timeit(few)
    print("R=%6d K=%4d N=%4d  streamed %.1f us (+ cut %.1f us once per step)  many-row %.1f  few-row %.1f   rel err %.1e (few-row %.1e)"
          % (R, K, N, tn, tc, tm, tf, e_new, e_ref), flush=True)
