"""A/B: first party-GRU layer as gather-then-project (default at cfg2) vs project-then-gather, whole captured cfg2 step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic, train, dialogue_model
from mm_dfn_amd.graphs import CapturedStep

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = dict(synthetic.CONFIGS[cfgname])
batch = synthetic.make_batch(2021, ragged=False, device="cuda", **cfg)
lengths = batch["lengths"]
label = train.flatten_labels(batch["label"], lengths)
loss_f = FocalLoss(gamma=0.5)


def timed(thr):
    dialogue_model.PROJECT_THEN_GATHER_ROWS = thr
    model = synthetic.build_model(dropout=0.5, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.cuda().train()

    def fwd_bwd():
        loss = loss_f(model(batch["textf"], batch["qmask"], batch["umask"], lengths, batch["acouf"], batch["visuf"])[0], label)
        train.backward(loss)
        return loss

    cap = CapturedStep(model, fwd_bwd, warmup=3)
    for _ in range(10):
        cap.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        cap.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 100


for thr in (12000, 1, 12000, 1):
    print("%s  project-then-gather %s: %.4f ms/step" % (cfgname, "ON " if thr == 1 else "off", timed(thr)), flush=True)
