"""What the weight-gradient riders take: per GRU backward launch of one step, the idle CUs, the budget, and every queued gradient
(output rows M, columns N, segments, contraction rows, taken?).  python tools/rider_log.py [cfg2|cfg3|cfg4] [--ragged]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_dfn_amd import ops_wgrad, synthetic, train
from mm_dfn_amd.loss import FocalLoss

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = dict(synthetic.CONFIGS[name])
model = synthetic.build_model(dropout=0.5, **cfg).cuda().train()
batch = synthetic.make_batch(2021, ragged="--ragged" in sys.argv, device="cuda", **cfg)
label = train.flatten_labels(batch["label"], batch["lengths"])
loss_f = FocalLoss(gamma=0.5)
for it in range(2):
    ops_wgrad.RIDER_LOG = []
    model.zero_grad(set_to_none=True)
    logp = model(batch["textf"], batch["qmask"], batch["umask"], batch["lengths"], batch["acouf"], batch["visuf"])[0]
    train.backward(loss_f(logp, label))
torch.cuda.synchronize()
for e in ops_wgrad.RIDER_LOG:
    print("launch: idle CUs %d, T %d, budget %.2f GFLOP, taken %.2f GFLOP in %d segments" % (e["idle"], e["T"], e["budget_gflop"],
                                                                                          e["taken_gflop"], e["taken_segments"]))
    for M, N, ns, R, t in e["queued"]:
        print("    %s  %4d x %4d  segments %2d  rows %6d  %.2f GFLOP" % ("RIDES" if t else "stays", M, N, ns, R, 2e-9 * R * M * N))
