"""Where the streamed leg's time goes: the SAME 32 ragged cfg2 batches
  A  replayed from exact-signature captured steps, inputs resident, nothing else (the device time of the real dialogues)
  B  replayed from BUCKETED entries (padding dialogue + bucket rounding + index retarget + label gather), inputs resident
  C  streamed through train_or_eval_graph_model with the bucketed cache (pinned-host prefetch, FlatAdam step, metrics)
so that  B - A = what the bucket costs on the device (padding to the bucket, the extra dialogue, retarget copies)  and
C - B = host / H2D / optimizer / metrics.  One MI355X; prints ms per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic, train  # noqa: E402
from mm_dfn_amd import data as D  # noqa: E402
from mm_dfn_amd.optim import FlatAdam  # noqa: E402

dev = "cuda"
cfg = dict(synthetic.CONFIGS["cfg2"])
NB = 32


def make(seed0, n, pin):
    out = []
    for i in range(n):
        b = synthetic.make_batch(seed0 + i, ragged=True, **cfg)
        t = [b[k] for k in ("textf", "visuf", "acouf", "qmask", "umask", "label")]
        out.append([x.pin_memory() for x in t] + [["u%d" % i]] if pin else [x.to(dev) for x in t])
    return out


def model_and_loss():
    m = synthetic.build_model(dropout=0.5, **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 2021))
    return m.to(dev).train(), FocalLoss(gamma=0.5)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / NB * 1e3


batches = make(9000, NB, pin=False)
utt = sum(int(b[4].sum()) for b in batches) / NB
m, loss_f = model_and_loss()
exact = train.StepGraphCache(m, loss_f, max_entries=NB + 4)
lens = [train.lengths_from_umask(b[4]) for b in batches]
a = timed(lambda: [exact.step(tuple(b), l, True) for b, l in zip(batches, lens)])
del exact
m, loss_f = model_and_loss()
buck = train.StepGraphCache(m, loss_f, max_entries=96, bucket_rows=32)
for w in range(4):
    buck.precapture(make(7000 + 100 * w, NB, pin=False), train_flag=True)
b_ = timed(lambda: [buck.step(tuple(b), l, True) for b, l in zip(batches, lens)])
h0, m0 = buck.hits, buck.misses
opt = FlatAdam(m, lr=3e-4, weight_decay=1e-4)
pre = D.DevicePrefetcher(make(9000, NB, pin=True), device=dev)
c = timed(lambda: train.train_or_eval_graph_model(m, loss_f, pre, 0, True, opt, False, graph_cache=buck))
print("avg utterances per batch %.0f" % utt)
print("A exact-signature replays, resident inputs          %.3f ms/step" % a)
print("B bucketed replays, resident inputs                 %.3f ms/step   (B - A = %.3f: bucket padding + extra dialogue + retarget)" % (b_, b_ - a))
print("C streamed pass (H2D + FlatAdam + metrics), bucketed %.3f ms/step   (C - B = %.3f: host / H2D / optimizer / metrics)" % (c, c - b_))
print("bucketed entries %d, captures during the timed runs %d" % (len(buck.entries), buck.misses - m0))
