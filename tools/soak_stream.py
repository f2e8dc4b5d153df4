"""cfg5 soak: N optimizer steps (fwd + loss + bwd through the weight-gradient batch's tall form + Adam) on one synthetic
long-dialogue batch: finite, decreasing loss.    python tools/soak_stream.py [steps] [cfg5|cfg5_b32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
name = sys.argv[2] if len(sys.argv) > 2 else "cfg5"
cfg = dict(synthetic.STREAM_CONFIGS[name])
model = synthetic.build_stream_model(dropout=0.3, **cfg)
model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
model = model.cuda().train()
b = synthetic.make_stream_batch(2021, device="cuda", **cfg)
label = train.flatten_labels(b["label"], b["lengths"])
loss_f = FocalLoss(gamma=0.5)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
hist = []
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    loss = loss_f(model(b["streams"], b["qmask"], b["umask"], b["lengths"])[0], label)
    train.backward(loss)
    opt.step()
    hist.append(float(loss))
print("%s: %d steps, loss %.4f -> %.4f (min %.4f), peak memory %.0f MiB" % (name, steps, hist[0], hist[-1], min(hist),
                                                                        torch.cuda.max_memory_allocated() / 2 ** 20))
assert all(x == x for x in hist), "NaN loss"
assert hist[-1] < hist[0], "loss did not decrease"
