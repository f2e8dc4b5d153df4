"""Timing aid: what the steps of a bucketed StepGraphCache cost -- the capture, the first replay-path step of an entry, exact-bucket
hits, hits served by a larger bucket (train.StepGraphCache larger_bucket_fallback)."""
import time, torch, cProfile, pstats, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mm_dfn_amd import synthetic, train as T
from mm_dfn_amd.loss import FocalLoss
CFG = dict(P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
m = synthetic.build_model(dropout=0.5, **CFG); m = m.cuda().train()
loss_f = FocalLoss(gamma=0.5)
cache = T.StepGraphCache(m, loss_f, bucket_rows=32)
def batch(seed, lengths):
    b = synthetic.make_batch(seed, lengths=lengths, device="cuda", B=len(lengths), L=max(lengths), **CFG)
    return (b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]), lengths
big = [110]*10 + [60]*6          # N = 1460 -> bucket 1472
exact = [110]*10 + [58]*6        # N = 1448 -> 1472 exact
small = [110]*10 + [50]*6        # N = 1400 -> 1408: fallback to 1472 (pad 72)
def run(lengths, seed, n=5):
    ts = []
    for i in range(n):
        inp, l = batch(seed + i, lengths)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cache.step(inp, l, True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return ts
print("capture", run(big, 1, 1))
print("2nd hit on the captured batch shape", run(big, 2, 2))
print("exact  ", run(exact, 10))
print("fallbk ", run(small, 20), cache.hits, cache.misses, cache.fallbacks)
print("exact  ", run(exact, 30))
pr = cProfile.Profile(); pr.enable(); run(small, 40, 3); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
