"""Race hunt: the same training step (same inputs, same dropout draws) replayed many times must give BIT-IDENTICAL gradients --
every kernel of this path sums in a fixed order, so any difference is an LDS / barrier hazard.  Covers the round-5 kernels:
cfg3 (MFMA form of the GRU recurrence), cfg4 (valid-length launches), cfg2, cfg5 through the module stack (K6 tail tile, wide K8
backward, K7 producer / consumer, one adjacency-gradient launch, wide weight-gradient tiles).
    python tools/soak_determinism.py [replays]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic, train  # noqa: E402
from mm_dfn_amd.graphs import CapturedStep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
loss_f = FocalLoss(gamma=0.5)


def check(name, model, fwd_bwd, replays):
    step = CapturedStep(model, fwd_bwd, warmup=1)
    st = torch.cuda.get_rng_state()
    ref = None
    bad = 0
    for i in range(replays):
        torch.cuda.set_rng_state(st)
        from mm_dfn_amd import ops
        ops.reseed_flags(1234) if hasattr(ops, "reseed_flags") else None
        step.replay()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        if ref is None:
            ref = g.clone()
        elif not torch.equal(g, ref):
            bad += 1
    print("%-10s %d replays, %d differ from the first%s" % (name, replays, bad, "" if bad == 0 else "   <-- NOT deterministic"), flush=True)
    return bad


total = 0
for cfgname, ragged in (("cfg2", False), ("cfg3", True), ("cfg4", True)):
    cfg = dict(synthetic.CONFIGS[cfgname])
    model = synthetic.build_model(dropout=0.0, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev).train()
    b = synthetic.make_batch(2021, ragged=ragged, device=dev, **cfg)
    label = train.flatten_labels(b["label"], b["lengths"])

    def fwd_bwd(model=model, b=b, label=label):
        loss = loss_f(model(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0], label)
        train.backward(loss)
        return loss
    total += check(cfgname, model, fwd_bwd, n)
    del model
for name, reps in (("cfg5", max(20, n // 4)), ("cfg5_b32", max(10, n // 10))):
    cfg = dict(synthetic.STREAM_CONFIGS[name])
    model = synthetic.build_stream_model(dropout=0.0, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev).train()
    batch = synthetic.make_stream_batch(2021, device=dev, **cfg)
    label = train.flatten_labels(batch["label"], batch["lengths"])

    def fwd_bwd(model=model, batch=batch, label=label):
        loss = loss_f(model(batch["streams"], batch["qmask"], batch["umask"], batch["lengths"])[0], label)
        train.backward(loss)
        return loss
    total += check(name, model, fwd_bwd, reps)
    del model
    torch.cuda.empty_cache()
print("OK" if total == 0 else "FAILED: %d replays differed" % total)
sys.exit(1 if total else 0)
