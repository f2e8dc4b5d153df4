"""List the weight-gradient contractions one training step queues (ops.queue_wgrad -> mmdfn_gemm_tn_batch): rows, output
shape, row shift and matrix work per segment.    python tools/dump_wgrad_segments.py [cfg2|cfg3|cfg4|cfg2_refdims|cfg5|cfg5_b32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, ops, synthetic, train  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if name in synthetic.STREAM_CONFIGS:
    cfg = dict(synthetic.STREAM_CONFIGS[name])
    model = synthetic.build_stream_model(dropout=0.5, **cfg)
    b = synthetic.make_stream_batch(2021, device="cuda", **cfg)
else:
    cfg = dict(synthetic.CONFIGS[name])
    model = synthetic.build_model(dropout=0.5, **cfg)
    b = synthetic.make_batch(2021, device="cuda", **cfg)
model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
model = model.cuda().train()
label = train.flatten_labels(b["label"], b["lengths"])
loss_f = FocalLoss(gamma=0.5)

seen = []
from mm_dfn_amd import ops_wgrad
orig = ops_wgrad._prepare_wgrad_batch


def spy(batch):
    for (o, Ct, cs, a, segs) in batch:
        for (At, Bt, s_) in segs:
            seen.append((At.shape[0], o["M"], o["N"], s_, At.stride(0), Bt.stride(0), len(cs)))
    return orig(batch)


ops_wgrad._prepare_wgrad_batch = spy
if name in synthetic.STREAM_CONFIGS:
    out = model(b["streams"], b["qmask"], b["umask"], b["lengths"])[0]
else:
    out = model(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
train.backward(loss_f(out, label))
torch.cuda.synchronize()
tot = 0.0
print("%7s %5s %5s %5s %6s %6s %4s %9s" % ("rows", "M", "N", "shift", "lda", "ldb", "bias", "MFLOP"))
for (R, M, N, s_, lda, ldb, nb) in seen:
    f = 2.0 * R * M * N / 1e6
    tot += f
    print("%7d %5d %5d %5d %6d %6d %4d %9.1f" % (R, M, N, s_, lda, ldb, nb, f))
print("%d segments, %.2f GFLOP" % (len(seen), tot / 1e3))
