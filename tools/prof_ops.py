import sys, os, torch
sys.path.insert(0, os.getcwd())
from mm_dfn_amd import FocalLoss, synthetic, train
cfg = dict(synthetic.CONFIGS["cfg2"])
m = synthetic.build_model(dropout=0.5, **cfg)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 1)); m = m.cuda().train()
b = synthetic.make_batch(2, device="cuda", **cfg)
label = train.flatten_labels(b["label"], b["lengths"]); lf = FocalLoss(gamma=0.5)
def step():
    m.zero_grad(set_to_none=True)
    logp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
    lf(logp, label).backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::add", "aten::add_", "aten::zeros", "aten::zero_", "aten::fill_", "aten::cat", "aten::zeros_like", "aten::sum", "aten::copy_", "aten::clone", "aten::mul", "aten::contiguous")]
for e in sorted(rows, key=lambda e: (e.key, -e.count)):
    print("%-18s x%-3d %s" % (e.key, e.count, str(e.input_shapes)[:110]))
