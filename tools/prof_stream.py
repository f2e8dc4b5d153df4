"""Host-side profile (cProfile) of the drop-in pass loop over streamed ragged batches, eager vs captured-step cache.
    python tools/prof_stream.py [n_batches]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic, train  # noqa: E402
from mm_dfn_amd import data as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
cfg = dict(synthetic.CONFIGS["cfg2"])
batches = []
for i in range(n):
    b = synthetic.make_batch(3000 + i, ragged=True, **cfg)
    batches.append([b["textf"].pin_memory(), b["visuf"].pin_memory(), b["acouf"].pin_memory(), b["qmask"].pin_memory(),
                    b["umask"].pin_memory(), b["label"].pin_memory(), ["b%d" % i]])
loss_f = FocalLoss(gamma=0.5)
for mode in ("eager", "captured"):
    model = synthetic.build_model(dropout=0.5, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=1e-4)
    cache = train.StepGraphCache(model, loss_f, max_entries=n + 4) if mode == "captured" else None
    run = lambda: train.train_or_eval_graph_model(model, loss_f, D.DevicePrefetcher(batches, device=dev), 0, True, opt, False,
                                                  graph_cache=cache)
    run()
    run()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    run()
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("==== %s: %.1f ms per batch" % (mode, dt / n * 1e3))
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:44]))
