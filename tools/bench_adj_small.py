"""K5 (adjacency build) forward / backward: the one-workgroup-per-(dialogue, modality) form of adjacency_small.hip against the
many-launch form of adjacency.hip, same inputs, captured launches with rotating buffer sets.  Tuning build (the switch
MMDFN_ADJ_SMALL=0 only exists there).

    python tools/bench_adj_small.py            # cfg2 / cfg4 / cfg3-like shapes
"""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import _hip, ops  # noqa: E402,F401
from mm_dfn_amd.layout import DialogueLayout  # noqa: E402
from mm_dfn_amd.ops_pad import _lay_args  # noqa: E402

dev = "cuda"
SHAPES = [("cfg2: 16 x 110, M=3, D=200", [110] * 16, 3, 200), ("cfg4: 32 x 110", [110] * 32, 3, 200),
          ("cfg2 ragged", [110, 97, 64, 33, 80, 71, 45, 27, 102, 58, 39, 90, 66, 51, 30, 70], 3, 200),
          ("cfg3-like: 32 x 33", [33] * 32, 3, 200), ("64 x 110", [110] * 64, 3, 200), ("128 x 110", [110] * 128, 3, 200)]
NSET = 6


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(NSET):
            fn(k)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (reps * NSET) * 1e3


def buffers(lay, M, N, D, gen):
    f32 = dict(dtype=torch.float32, device=dev)
    b = dict(feats=torch.randn(M, N, D, device=dev, generator=gen), unit=torch.empty(M, N, D, **f32), norm=torch.empty(M, N, **f32),
             cosg=torch.empty(lay.tile_elems, **f32), cdot=torch.empty(lay.npairs, N, **f32), rdeg=torch.empty(M, N, **f32),
             tiles=torch.empty(lay.tile_elems, **f32), cross=torch.empty(lay.npairs, N, **f32),
             dtiles=torch.randn(lay.tile_elems, device=dev, generator=gen), dcross=torch.randn(lay.npairs, N, device=dev, generator=gen),
             wsym=torch.empty(lay.tile_elems, **f32), etile=torch.empty(lay.tile_elems, **f32), ecross=torch.empty(lay.npairs, N, **f32),
             ddeg=torch.empty(M, N, **f32), dunit=torch.empty(M, N, D, **f32), dfeats=torch.empty(M, N, D, **f32),
             addend=torch.randn(M, N, D, device=dev, generator=gen))
    return b


def run_fwd(b, lay, M, N, D):
    P = _hip.ptr
    rc = _hip.lib().mmdfn_adj_build(P(b["feats"]), P(b["unit"]), P(b["norm"]), P(b["cosg"]), P(b["cdot"]), P(b["rdeg"]), P(b["tiles"]),
                                    P(b["cross"]), *_lay_args(lay), lay.B, M, N, D, lay.max_len, 1.0, _hip.stream())
    _hip.check(rc, "mmdfn_adj_build")


def run_bwd(b, lay, M, N, D):
    P = _hip.ptr
    rc = _hip.lib().mmdfn_adj_build_bwd(P(b["dtiles"]), P(b["dcross"]), P(b["unit"]), P(b["norm"]), P(b["cosg"]), P(b["cdot"]),
                                        P(b["rdeg"]), P(b["tiles"]), P(b["cross"]), P(b["wsym"]), P(b["etile"]), P(b["ecross"]),
                                        P(b["ddeg"]), P(b["dunit"]), P(b["dfeats"]), P(b["addend"]), *_lay_args(lay), lay.B, M, N, D,
                                        lay.max_len, 1.0, _hip.stream())
    _hip.check(rc, "mmdfn_adj_build_bwd")


for name, lengths, M, D in SHAPES:
    N = sum(lengths)
    lay = DialogueLayout.get(lengths, M, torch.device(dev))
    res = {}
    for form in ("1", "0"):
        os.environ["MMDFN_ADJ_SMALL"] = form
        gen = torch.Generator(device=dev).manual_seed(5)
        bufs = [buffers(lay, M, N, D, gen) for _ in range(NSET)]
        t_f = timed(lambda k=0: run_fwd(bufs[k], lay, M, N, D))
        t_b = timed(lambda k=0: run_bwd(bufs[k], lay, M, N, D))
        torch.cuda.synchronize()
        res[form] = (t_f, t_b, bufs[0])
    a, o = res["1"][2], res["0"][2]
    dt = float((a["tiles"] - o["tiles"]).abs().max())
    dc = float((a["cross"] - o["cross"]).abs().max())
    dg = float((a["dfeats"] - o["dfeats"]).abs().max() / o["dfeats"].abs().max())
    print("%-28s forward %6.1f -> %6.1f us   backward %6.1f -> %6.1f us   max|dT| %.1e  max|dcross| %.1e  rel dfeats %.1e" % (
        name, res["0"][0], res["1"][0], res["0"][1], res["1"][1], dt, dc, dg), flush=True)
