"""Soak (SOAK_FLAT_ADAM=1: fused optimizer; SOAK_GRAPH=1: captured-step cache): many training steps on ragged synthetic data (every batch a new shape): finite losses, decreasing
training loss, bounded device memory (no per-shape leak: DialogueLayout cache is capped, graph pools are not used)."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import FocalLoss, synthetic
from mm_dfn_amd import data as D, train as T
from mm_dfn_amd.optim import FlatAdam

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
p = D.write_synthetic_pickle(os.path.join(tempfile.mkdtemp(), "f.pkl"), n_train=120, n_test=31, max_len=110, min_len=20, seed=3)
m = synthetic.build_model(P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512, dropout=0.3)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 4)); m = m.cuda()
opt = (FlatAdam(m, lr=3e-4, weight_decay=1e-5) if os.environ.get("SOAK_FLAT_ADAM") else
       torch.optim.Adam(m.parameters(), lr=3e-4, weight_decay=1e-5))
tr, va, te = D.get_IEMOCAP_loaders(p, batch_size=16, valid_rate=0.1, bucketed=True)
t0 = time.time(); peak0 = None
hist = []
def log(msg):
    hist.append(msg)
out = T.fit(m, FocalLoss(gamma=0.5), opt, D.DevicePrefetcher(tr), D.DevicePrefetcher(va), D.DevicePrefetcher(te),
            n_epochs=epochs, patience=10 ** 6, valid_rate=0.1, log=log,
            graph_cache=True if os.environ.get("SOAK_GRAPH") else None)
torch.cuda.synchronize()
h = out["history"]
mem = torch.cuda.max_memory_allocated() / 2 ** 20
print("epochs %d in %.1f s; train loss %.4f -> %.4f; test F1 last %.2f; peak device memory %.0f MiB" % (
    out["epochs_run"], time.time() - t0, h["train_loss"][0], h["train_loss"][-1], h["test_fscore"][-1], mem))
assert all(x == x for x in h["train_loss"] + h["valid_loss"] + h["test_loss"]), "NaN loss"
assert h["train_loss"][-1] < h["train_loss"][0]
assert mem < 4096
print("soak ok")
