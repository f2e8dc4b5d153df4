"""Few-row projection kernel (csrc/linear_small.hip, grouped launches) vs the library GEMMs it replaces, on the shapes
of the cfg2 / cfg3 / cfg4 steps; graph-captured timing, correctness against torch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops


def gtime(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3


dev = "cuda"
R_ = lambda *s: torch.randn(*s, device=dev)
print("== single problems: y = x W^T + b (NK) and dX = dY Wcat (KN)")
for name, R, K, N, km in [("linear_a / linear_l fwd", 1760, 100, 200, 0), ("linear_v fwd", 1760, 512, 200, 0),
                          ("ctx GRU gi", 1760, 200, 600, 0), ("ctx GRU dX", 1760, 600, 200, 1), ("party GRU dX", 7040, 600, 200, 1),
                          ("party GRU gi", 7040, 200, 600, 0), ("cfg4 linear_v fwd", 3520, 512, 200, 0), ("cfg4 ctx gi", 3520, 200, 600, 0),
                          ("cfg4 ctx dX", 3520, 600, 200, 1), ("cfg4 party dX", 14080, 600, 200, 1), ("cfg3 text fwd", 1056, 600, 200, 0),
                          ("cfg3 party dX", 19008, 600, 200, 1)]:
    x = R_(R, K)
    if km:
        wk = R_(K, N)
        ref = x @ wk
        t0 = gtime(lambda: x @ wk)
        y = ops.linear_group_raw([dict(x=x, wk=wk)])[0]
        t1 = gtime(lambda: ops.linear_group_raw([dict(x=x, wk=wk)]))
    else:
        w, b = R_(N, K), R_(N)
        ref = torch.nn.functional.linear(x, w, b)
        t0 = gtime(lambda: torch.nn.functional.linear(x, w, b))
        y = ops.linear_group_raw([dict(x=x, w=w, b=b)])[0]
        t1 = gtime(lambda: ops.linear_group_raw([dict(x=x, w=w, b=b)]))
    err = float((y - ref).abs().max() / ref.abs().max())
    print("%-26s R=%6d K=%4d N=%4d  library %6.1f us | few-row kernel %6.1f us  err %.1e" % (name, R, K, N, t0, t1, err), flush=True)

print("== grouped launches vs the separate library launches they replace")
xa, xv, xl = R_(1760, 100), R_(1760, 512), R_(1760, 100)
wa, wv, wl = R_(200, 100), R_(200, 512), R_(200, 100)
ba, bv, bl = R_(200), R_(200), R_(200)
F = torch.nn.functional.linear
t0 = gtime(lambda: (F(xa, wa, ba), F(xv, wv, bv), F(xl, wl, bl)))
t1 = gtime(lambda: ops.linear_group_raw([dict(x=xa, w=wa, b=ba), dict(x=xv, w=wv, b=bv), dict(x=xl, w=wl, b=bl)]))
print("three modality projections (cfg2):      library 3 launches %6.1f us | one grouped launch %6.1f us" % (t0, t1))
dyc, dyp, wc1, wc2 = R_(1760, 600), R_(7040, 600), R_(600, 200), R_(600, 200)
t0 = gtime(lambda: (dyc @ wc1, dyp @ wc2))
t1 = gtime(lambda: ops.linear_group_raw([dict(x=dyc, wk=wc1), dict(x=dyp, wk=wc2)]))
print("ctx + party GRU input gradients (cfg2): library 2 launches %6.1f us | one grouped launch %6.1f us" % (t0, t1))
xc, xp = R_(1760, 200), R_(7040, 200)
w1, w2, w3, w4 = R_(300, 200), R_(300, 200), R_(300, 200), R_(300, 200)
b1, b2, b3, b4 = R_(300), R_(300), R_(300), R_(300)
t0 = gtime(lambda: (F(xc, torch.cat([w1, w2]), torch.cat([b1, b2])), F(xp, torch.cat([w3, w4]), torch.cat([b3, b4]))))
t1 = gtime(lambda: ops.linear_group_raw([dict(x=xc, w=w1, w2=w2, b=b1, b2=b2), dict(x=xp, w=w3, w2=w4, b=b3, b2=b4)]))
y = ops.linear_group_raw([dict(x=xc, w=w1, w2=w2, b=b1, b2=b2)])[0]
ref = F(xc, torch.cat([w1, w2]), torch.cat([b1, b2]))
print("ctx + party GRU gi, two weight blocks:  library (+cat) %6.1f us | one grouped launch %6.1f us  err %.1e" % (
    t0, t1, float((y - ref).abs().max() / ref.abs().max())))
