"""K6 at cfg5 (B=32, L=512, M=6, d=100), round 6: which workgroups share a CU, and what the cross-modal rows cost as L2 -> L1
traffic rather than as instructions.  Tuning build; protocol of bench.py's roofline_cfg5 leg for the timings.

  MMDFN_PROP_ABL bits (propagate_split.hip): 1 no cross-modal rows, 2 one row instead of five, 8 the rows alias 8 rows per workgroup
  (L1 hits: the instruction cost without the traffic), 256 / 512 workgroup decode variants (neighbours on a CU = two row blocks of
  one modality / two modalities of one row block), MMDFN_TRACE_PTR = per-workgroup s_memtime stamps + HW_ID.
"""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops  # noqa: E402

dev = "cuda"
B = int(os.environ.get("K6_B", "32"))
l5 = [512] * B
d = 100
sets = []
for i in range(3):
    g = torch.Generator(device=dev).manual_seed(500 + i)
    adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
    H = torch.randn(6 * sum(l5), d, device=dev, generator=g)
    sets.append((adj, H, torch.empty_like(H)))
lay = sets[0][0].layout
alg = lay.propagate_bytes(d)
os.environ["MMDFN_SPLIT_TAIL"] = "1"
os.environ["MMDFN_SPLIT_ABLC"] = "0"


def run(abl):
    os.environ["MMDFN_PROP_ABL"] = str(abl)
    os.environ.pop("MMDFN_TRACE_PTR", None)
    for adj, H, o in sets:
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=o)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for it in range(21):
            adj, H, o = sets[it % 3]
            ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=o)
    for _ in range(15):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 105 * 1e3


def check(abl):
    """decode variants must not change the result"""
    os.environ["MMDFN_PROP_ABL"] = "0"
    adj, H, o = sets[0]
    ref = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout).clone()
    os.environ["MMDFN_PROP_ABL"] = str(abl)
    got = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    torch.cuda.synchronize()
    return bool(torch.equal(ref, got))


def trace(abl):
    nwg = ((B + 7) // 8) * 8 * 6 * 4
    tr = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    os.environ["MMDFN_PROP_ABL"] = str(abl)
    os.environ["MMDFN_TRACE_PTR"] = str(tr.data_ptr())
    for it in range(12):
        adj, H, o = sets[it % 3]
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=o)
    torch.cuda.synchronize()
    os.environ.pop("MMDFN_TRACE_PTR", None)
    t = tr.cpu().numpy().reshape(nwg, 8)
    st = t[:, :5].astype(np.float64)
    st -= st[:, 0].min()
    hw = t[:, 7]
    xcc, hwid = hw >> 32, hw & 0xffffffff
    key = xcc * 1000 + ((hwid >> 13) & 7) * 100 + ((hwid >> 12) & 1) * 50 + ((hwid >> 8) & 0xf)
    names = ["entry", "set-up", "loop done", "pass 0", "pass 1"]
    dur = np.diff(st, axis=1)
    print("  abl=%d: span %.0f ticks; %d CUs; phase means (ticks): %s" % (
        abl, st[:, 4].max(), len(set(key.tolist())),
        ", ".join("%s->%s %.0f" % (names[k], names[k + 1], dur[:, k].mean()) for k in range(4))))
    # co-residency: for every CU the workgroups in start order, as XCD-local indices yq = bid >> 3
    first = st[:, 0] < 0.25 * st[:, 4].max()
    print("  first-round workgroups: %d, later: %d" % (first.sum(), (~first).sum()))
    per_cu = {}
    for b in np.argsort(st[:, 0]):
        per_cu.setdefault(int(key[b]), []).append(int(b))
    npair = nshare = 0
    deltas = {}
    for k, bl in per_cu.items():
        fr = [b for b in bl if first[b]]
        if len(fr) == 2 and (fr[0] & 7) == (fr[1] & 7):
            npair += 1
            dy = abs((fr[0] >> 3) - (fr[1] >> 3))
            deltas[dy] = deltas.get(dy, 0) + 1
    print("  CUs whose two first-round workgroups are on one XCD: %d; |yq difference| histogram: %s" % (
        npair, sorted(deltas.items(), key=lambda kv: -kv[1])[:8]))
    ex = sorted(per_cu.items())[:4]
    for k, bl in ex:
        print("   CU %d: %s" % (k, [(b & 7, b >> 3, int(st[b, 0]), int(st[b, 2]), int(st[b, 4])) for b in bl]))
    # epilogue phase overlap: how many workgroups are inside their epilogue at the same time (bursts)
    return t


print("decode variants bit-equal:", check(256), check(512))
rows = [("shipped", 0), ("neighbours = row blocks of one modality (256)", 256), ("neighbours = modalities of one row block (512)", 512),
        ("cross rows alias 8 rows / workgroup (L1 hits; timing only)", 8), ("no cross-modal rows (timing only)", 1),
        ("one cross-modal row (timing only)", 2), ("512 + alias", 520)]
for rep in range(2):
    for name, abl in rows:
        us = run(abl)
        print("%-62s %6.1f us   %.3f of 8 TB/s" % (name, us, alg / (us * 1e-6) / 8e12), flush=True)
for abl in (0, 256, 512):
    t = trace(abl)
    np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "k6_trace_%d.npy" % abl), t)
