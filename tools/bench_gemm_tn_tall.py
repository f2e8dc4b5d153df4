"""Weight-gradient batch of a BASELINE cfg5 graph-stack backward (8 layers: LSTM-gate segments 400 x 100 from dG^T q and dG^T h,
GCN-layer segments 100 x 100) timed as ONE mmdfn_gemm_tn_batch call: tall form against the 64 x 112 tiles
(MMDFN_TN_NO_TALL=1, tuning build), rotating operand sets against one hot set.
    python tools/bench_gemm_tn_tall.py [B] [ENV=V[,ENV=V...] ...]     one timed run per argument after B
TALL_ONLY=1 in the environment: only the 31 graph-stack segments.  MMDFN_TN_TALL_ABL=1|2|4|6|7|8|12: compile-time ablations of
the tall form (1 no operand DMA, 2 no wait / barrier, 4 no fragment reads, 8 no MFMAs; timing only).  MMDFN_TN_TALL_ROWS7 /
MMDFN_TN_TALL_ROWS2: rows per workgroup of the 49- / 14-tile kind."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R, H, LAYERS = B * 6 * 512, 100, 8
t = lambda *s: torch.randn(*s, device="cuda")

def make_batch(hot):
    n = 1 if hot else LAYERS
    dG, q, h = [t(R, 4 * H) for _ in range(n)], [t(R, H) for _ in range(n)], [t(R, H) for _ in range(n)]
    hi, dP, h0 = [t(R, H) for _ in range(n)], [t(R, H) for _ in range(n)], t(R, H)
    w_ih, w_hh, b_ih, b_hh = t(4 * H, H), t(4 * H, H), t(4 * H), t(4 * H)
    conv = [t(2 * H, H) for _ in range(LAYERS)]
    batch = [(dict(M=4 * H, N=H), w_ih, [b_ih, b_hh], 0, [(dG[i % n], q[i % n], 0) for i in range(LAYERS)]),
             (dict(M=4 * H, N=H), w_hh, [], 0, [(dG[i % n], h[i % n], 0) for i in range(1, LAYERS)])]
    for i in range(LAYERS):
        batch.append((dict(M=H, N=H), conv[i][:H], [], 0, [(hi[i % n], dP[i % n], 0)]))
        batch.append((dict(M=H, N=H), conv[i][H:], [], 0, [(h0, dP[i % n], 0)]))
    # the stack's input layer (100 x 200) and the six stream projections (200 x 512, B x 512 rows each)
    if os.environ.get("TALL_ONLY"):      # only the 31 graph-stack segments
        return batch
    dpre, xd = t(R, H), t(R, 2 * H)
    batch.append((dict(M=H, N=2 * H), t(H, 2 * H), [t(H)], 0, [(dpre, xd, 0)]))
    for m in range(6):
        batch.append((dict(M=2 * H, N=512), t(2 * H, 512), [t(2 * H)], 0, [(t(B * 512, 2 * H), t(B * 512, 512), 0)]))
    return batch

flops = 2.0 * R * H * (4 * H * (2 * LAYERS - 1) + H * 2 * LAYERS)
if not os.environ.get("TALL_ONLY"):
    flops += 2.0 * R * H * 2 * H + 6 * 2.0 * B * 512 * 2 * H * 512
for hot in (False, True):
    batch = make_batch(hot)
    for env in sys.argv[2:] or ["MMDFN_TN_NO_TALL=1", "MMDFN_TN_NO_TALL=0"]:
        for kv in env.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
        for _ in range(2): ops._launch_wgrad_batch(batch)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(4): ops._launch_wgrad_batch(batch)
        for _ in range(2): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 4 * 1e3
        print("B=%d %s %-40s %8.1f us  %6.1f TFLOP/s (useful)" % (B, "hot " if hot else "cold", env, us, flops / us / 1e6), flush=True)
