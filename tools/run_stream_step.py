"""BASELINE config 5 through the module stack (MultiStreamGraphModel), N captured fwd + loss + bwd steps: the workload
behind the cfg5 rocprof trace.    python tools/run_stream_step.py [cfg5|cfg5_b32] [steps] [fused-stack row limit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, gcn_stack, synthetic, train  # noqa: E402
from mm_dfn_amd.graphs import CapturedStep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if len(sys.argv) > 3:
    gcn_stack.ROW_LIMIT = int(sys.argv[3])
cfg = dict(synthetic.STREAM_CONFIGS[name])
model = synthetic.build_stream_model(dropout=0.5, **cfg)
model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
model = model.cuda().train()
b = synthetic.make_stream_batch(2021, device="cuda", **cfg)
label = train.flatten_labels(b["label"], b["lengths"])
loss_f = FocalLoss(gamma=0.5)


def fwd_bwd():
    loss = loss_f(model(b["streams"], b["qmask"], b["umask"], b["lengths"])[0], label)
    train.backward(loss)
    return loss


cap = CapturedStep(model, fwd_bwd, warmup=1)
cap.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    cap.replay()
e1.record()
e1.synchronize()
print("ran %d steps of %s, %.3f ms/step, loss %.4f" % (steps, name, e0.elapsed_time(e1) / steps, float(cap.loss)))
