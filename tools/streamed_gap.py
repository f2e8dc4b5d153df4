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


HOST = {}


def timed(fn, reps=3, tag=None):
    fn()
    torch.cuda.synchronize()
    best = host = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        th = time.perf_counter() - t0               # the host is done issuing (the pass loop's own .cpu() at its end included)
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        if t1 < best:
            best, host = t1, th
    if tag:
        HOST[tag] = host / NB * 1e3
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
b_ = timed(lambda: [buck.step(tuple(b), l, True) for b, l in zip(batches, lens)], tag='B')
h0, m0 = buck.hits, buck.misses
opt = FlatAdam(m, lr=3e-4, weight_decay=1e-4)
class _Batch(list):
    lengths = None


res = []                                                               # the same batches, already on the device, lengths on the host
for i, (b, l) in enumerate(zip(batches, lens)):
    r = _Batch(list(b) + [["r%d" % i]])
    r.lengths = list(l)
    res.append(r)
c1 = timed(lambda: train.train_or_eval_graph_model(m, loss_f, res, 0, True, opt, False, graph_cache=buck), tag='C1')

pre = D.DevicePrefetcher(make(9000, NB, pin=True), device=dev)
c = timed(lambda: train.train_or_eval_graph_model(m, loss_f, pre, 0, True, opt, False, graph_cache=buck), tag='C')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
train.train_or_eval_graph_model(m, loss_f, pre, 0, True, opt, False, graph_cache=buck)
pr.disable(); torch.cuda.synchronize()

print("avg utterances per batch %.0f" % utt)
print("A exact-signature replays, resident inputs          %.3f ms/step" % a)
print("B bucketed replays, resident inputs                 %.3f ms/step   (B - A = %.3f: bucket padding + extra dialogue + retarget)" % (b_, b_ - a))
print("C1 pass loop, resident inputs (FlatAdam + metrics)  %.3f ms/step   (C1 - B = %.3f: optimizer step, plane refresh, metrics, loop)" % (c1, c1 - b_))
print("C streamed pass (H2D + FlatAdam + metrics), bucketed %.3f ms/step   (C - C1 = %.3f: pinned-host prefetch / staging)" % (c, c - c1))
print("host issue time per step: B %.3f ms, C1 %.3f ms, C %.3f ms" % (HOST["B"], HOST["C1"], HOST["C"]))
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
print("bucketed entries %d, captures during the timed runs %d" % (len(buck.entries), buck.misses - m0))
