"""Where do the library / ATen launches of one eager cfg2 step come from?  Runs a few eager fwd + loss + bwd steps under
torch.profiler with Python stacks and prints, per device kernel that is NOT one of ours, the call sites inside
mm_dfn_amd that launched it.      python tools/trace_aten.py [cfg2] > gpurun_out/aten_sites.txt"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic, train, dialogue_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if len(sys.argv) > 2:                                   # party-row threshold of the project-then-gather form
    dialogue_model.PROJECT_THEN_GATHER_ROWS = int(sys.argv[2])
cfg = dict(synthetic.CONFIGS[name])
model = synthetic.build_model(dropout=0.5, **cfg)
model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
model = model.cuda().train()
batch = synthetic.make_batch(2021, ragged=False, device="cuda", **cfg)
lengths = batch["lengths"]
label = train.flatten_labels(batch["label"], lengths)
loss_f = FocalLoss(gamma=0.5)


def step():
    for p in model.parameters():
        p.grad = None
    logp = model(batch["textf"], batch["qmask"], batch["umask"], lengths, batch["acouf"], batch["visuf"])[0]
    loss = loss_f(logp, label)
    train.backward(loss)            # (the step as bench.py runs it: weight gradients batched)


for _ in range(3):
    step()
torch.cuda.synchronize()
NSTEP = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(NSTEP):
        step()
    torch.cuda.synchronize()

ours = ("anonymous namespace",)
sites = collections.defaultdict(lambda: [0, 0.0])
events = prof.events()
for ev in events:
    if not ev.kernels:
        continue
    if ev.cpu_children:            # only the innermost op that launched the kernel
        if any(c.kernels for c in ev.cpu_children):
            continue
    for k in ev.kernels:
        if any(o in k.name for o in ours) and "at::native" not in k.name:
            continue
        frames = [f for f in (ev.stack or []) if "mm_dfn_amd" in f or "tools/" in f or "autograd" in f]
        where = " <- ".join(f.strip().split("/")[-1] for f in frames[:3]) or "(no python frame: autograd engine)"
        key = (k.name[:70], ev.name, str(ev.input_shapes)[:80], where)
        sites[key][0] += 1
        sites[key][1] += k.duration
for (kname, op, shapes, where), (n, us) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
    print("%5.1f x  %7.1f us/step  %-28s %s\n        %s\n        %s" % (n / NSTEP, us / NSTEP, op, kname, shapes, where))
