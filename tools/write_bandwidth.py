"""Device write / read / copy rates on large buffers (rotating, > 256 MB MALL): the practical ceilings the HBM-bound kernels are
measured against.  torch elementwise kernels (fill, sum, copy), the library gemv and this package's column-sum kernel (a plain
16-byte-load streaming read) through a captured graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops
def gtime(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for mb in (50, 200, 800):
    n = mb * 1024 * 1024 // 4
    bufs = [torch.empty(n, device="cuda") for _ in range(3)]
    outs = [torch.empty(n, device="cuda") for _ in range(3)]
    k = [0]
    def fill():
        bufs[k[0] % 3].fill_(1.0); k[0] += 1
    def read():
        torch.sum(bufs[k[0] % 3]); k[0] += 1
    def copy():
        outs[k[0] % 3].copy_(bufs[k[0] % 3]); k[0] += 1
    mats = [b.view(-1, 512) for b in bufs]
    vec = torch.randn(512, device="cuda")
    def gemv():
        torch.mv(mats[k[0] % 3], vec); k[0] += 1
    def csum():
        ops.colsum(mats[k[0] % 3]); k[0] += 1
    tf, tr, tc, tg, ts = gtime(fill), gtime(read), gtime(copy), gtime(gemv), gtime(csum)
    print("%4d MB: fill %.2f TB/s (%.1f us) | read (sum) %.2f TB/s (%.1f us) | copy %.2f TB/s of traffic (%.1f us) | gemv read %.2f TB/s | "
          "column-sum read %.2f TB/s" % (mb, n * 4 / tf / 1e12, tf * 1e6, n * 4 / tr / 1e12, tr * 1e6, 2 * n * 4 / tc / 1e12, tc * 1e6,
                                         n * 4 / tg / 1e12, n * 4 / ts / 1e12))
