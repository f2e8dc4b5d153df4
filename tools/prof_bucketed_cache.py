import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from mm_dfn_amd import FocalLoss, synthetic, train
from mm_dfn_amd import data as D
from mm_dfn_amd.optim import FlatAdam
from mm_dfn_amd import layout
dev = "cuda"
cfg = dict(synthetic.CONFIGS["cfg2"])
def make(seed0, n):
    out = []
    for i in range(n):
        b = synthetic.make_batch(seed0 + i, ragged=True, **cfg)
        out.append([b["textf"].pin_memory(), b["visuf"].pin_memory(), b["acouf"].pin_memory(), b["qmask"].pin_memory(), b["umask"].pin_memory(), b["label"].pin_memory(), ["u%d" % i]])
    return out
model = synthetic.build_model(dropout=0.5, **cfg); model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021)); model = model.to(dev)
loss_f = FocalLoss(gamma=0.5)
opt = FlatAdam(model, lr=3e-4, weight_decay=1e-4)
cache = train.StepGraphCache(model, loss_f, max_entries=96, bucket_rows=32)
for w in range(3):
    train.train_or_eval_graph_model(model, loss_f, D.DevicePrefetcher(make(7000 + 100 * w, 32), device=dev), 0, True, opt, False, graph_cache=cache)
# instrument
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
wrap(layout.IndexScope, "retarget", "retarget")
wrap(cache, "step", "step_total")
wrap(opt, "step", "opt_step")
orig_replay = None
from mm_dfn_amd.graphs import CapturedStep
wrap(CapturedStep, "replay", "replay_call")
un = make(9500, 32)
torch.cuda.synchronize(); t0 = time.perf_counter()
train.train_or_eval_graph_model(model, loss_f, D.DevicePrefetcher(un, device=dev), 0, True, opt, False, graph_cache=cache)
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("pass host %.1f ms, incl sync %.1f ms (%.3f ms/step)" % (t_host * 1e3, t_all * 1e3, t_all / 32 * 1e3))
for k, v in T.items(): print("  %-12s %.3f ms/step" % (k, v / 32 * 1e3))
print("hits", cache.hits, "misses", cache.misses)
import cProfile, pstats
un2 = make(9700, 32)
pr = cProfile.Profile(); pr.enable()
train.train_or_eval_graph_model(model, loss_f, D.DevicePrefetcher(un2, device=dev), 0, True, opt, False, graph_cache=cache)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
