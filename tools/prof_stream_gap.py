"""Where the streamed pass loop's time goes (cfg2 ragged, 32 different batches, captured-step cache, FlatAdam): wall time per
step against the device time per step (torch profiler: kernels + memcpys on every stream) and the host's own enqueue time
per step (the loop timed without a final synchronise).  python tools/prof_stream_gap.py [n_batches]"""
import os
import sys
import time

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic, train  # noqa: E402
from mm_dfn_amd import data as D  # noqa: E402
from mm_dfn_amd.optim import FlatAdam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
cfg = dict(synthetic.CONFIGS["cfg2"])
batches = []
for i in range(n):
    b = synthetic.make_batch(3000 + i, ragged=True, **cfg)
    batches.append([b["textf"].pin_memory(), b["visuf"].pin_memory(), b["acouf"].pin_memory(), b["qmask"].pin_memory(),
                    b["umask"].pin_memory(), b["label"].pin_memory(), ["b%d" % i]])
loss_f = FocalLoss(gamma=0.5)
model = synthetic.build_model(dropout=0.5, **cfg)
model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
model = model.to(dev)
opt = FlatAdam(model, lr=3e-4, weight_decay=1e-4)
cache = train.StepGraphCache(model, loss_f, max_entries=n + 4)
RECYCLE = os.environ.get("RECYCLE", "1") == "1"      # A/B: device staging ring instead of a fresh allocation per batch
DIRECT = os.environ.get("DIRECT", "1") == "1"        # A/B: host -> static input buffers of the captured step directly


class _NoBind(D.DevicePrefetcher):
    def bind_graph_cache(self, graph_cache, train_flag):
        pass


run = lambda: train.train_or_eval_graph_model(
    model, loss_f, (D.DevicePrefetcher if DIRECT else _NoBind)(batches, device=dev, recycle=RECYCLE), 0, True, opt, False,
    graph_cache=cache)
for _ in range(3):
    run()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    run()
    best = min(best, time.perf_counter() - t0)
print("RECYCLE=%s DIRECT=%s: best of 5 passes %.3f ms per step" % (RECYCLE, DIRECT, best / n * 1e3))
if os.environ.get("QUICK"):
    sys.exit(0)
t0 = time.perf_counter()
run()
t_host = time.perf_counter() - t0            # includes the metrics' .cpu() at the end of the pass (a sync)
torch.cuda.synchronize()
t0 = time.perf_counter()
run()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    run()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
busy = sum(e.device_time for e in ev) / 1e3
spans = sorted((e.time_range.start, e.time_range.end) for e in ev)
union, cur_s, cur_e = 0.0, None, None
for s_, e_ in spans:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union += (cur_e - cur_s) if cur_e is not None else 0.0
print("per step: wall %.3f ms | device busy (sum over streams) %.3f ms | device occupied (union) %.3f ms | pass incl. metrics %.3f ms"
      % (wall / n * 1e3, busy / n, union / 1e3 / n, t_host / n * 1e3))
agg = {}
for e in ev:
    k = e.name[:70]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += e.device_time
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-70s x%5.1f/step %8.1f us/step" % (k, c / n, t / n))
cpu = {}
for e in prof.key_averages():
    if e.device_type == torch.autograd.DeviceType.CPU and e.self_cpu_time_total > 0:
        cpu[e.key[:60]] = (e.count, e.self_cpu_time_total)
print("host (self time per step):")
for k, (c, t) in sorted(cpu.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-60s x%5.1f/step %8.1f us/step" % (k, c / n, t / n))
