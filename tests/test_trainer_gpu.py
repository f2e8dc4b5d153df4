"""GPU: pickle -> loaders -> (prefetched) batches -> train_or_eval_graph_model / fit on the HIP model."""
import numpy as np
import pytest
import torch

from mm_dfn_amd import FocalLoss, synthetic
from mm_dfn_amd import data as D
from mm_dfn_amd import train as T

pytestmark = pytest.mark.gpu
CFG = dict(P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)


def _model(seed=5, dropout=0.0):
    m = synthetic.build_model(dropout=dropout, **CFG)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), seed))
    return m.cuda()


def test_prefetcher_yields_the_loader_batches_on_device(tmp_path):
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=13, n_test=5, max_len=30, seed=2)
    _, _, te = D.get_IEMOCAP_loaders(p, batch_size=2, valid_rate=0.0)
    plain = list(te)
    pre = list(D.DevicePrefetcher(te, depth=2))
    assert len(plain) == len(pre) == 3
    for a, b in zip(plain, pre):
        for x, y in zip(a[:6], b[:6]):
            assert y.is_cuda and torch.equal(x, y.cpu())
        assert a[6] == b[6]


def test_eval_pass_same_through_prefetcher_and_bucketing(tmp_path):
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=4, n_test=21, max_len=40, seed=4)
    m = _model()
    loss_f = FocalLoss(gamma=0.5)
    names = ['hap', 'sad', 'neu', 'ang', 'exc', 'fru']
    _, _, te = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0)
    _, _, te_b = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0, bucketed=True)
    r0 = T.train_or_eval_graph_model(m, loss_f, te, cuda_flag=True, target_names=names)
    r1 = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(te), cuda_flag=False, target_names=names)
    r2 = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(te_b), cuda_flag=False, target_names=names)
    assert r0[2] == r1[2] and r0[3] == r1[3] and r0[6] == r1[6] and np.array_equal(r0[5], r1[5])
    # bucketing regroups the dialogues (dialogues are independent, so per-utterance predictions are unchanged):
    # same multiset of (label, pred) pairs, same accuracy / F1
    assert r2[3] == r0[3] and r2[6] == r0[6]
    assert sorted(zip(r0[4].tolist(), r0[5].tolist())) == sorted(zip(r2[4].tolist(), r2[5].tolist()))


def test_fit_trains_and_stops(tmp_path):
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=20, n_test=6, max_len=24, seed=6)
    m = _model(dropout=0.1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
    tr, va, te = D.get_IEMOCAP_loaders(p, batch_size=8, valid_rate=0.2, bucketed=True)
    wrap = D.DevicePrefetcher
    out = T.fit(m, FocalLoss(gamma=0.5), opt, wrap(tr), wrap(va), wrap(te), n_epochs=6, patience=2, valid_rate=0.2,
                log=None)
    h = out["history"]
    assert 1 <= out["epochs_run"] <= 6 and all(np.isfinite(h["train_loss"]))
    assert h["train_loss"][-1] < h["train_loss"][0]      # it learns the (memorisable) synthetic labels
    assert 0 <= out["by_f1"]["epoch"] < out["epochs_run"]


def test_load_reference_weights_roundtrip(tmp_path):
    m = _model(seed=9)
    path = str(tmp_path / "sd.pt")
    torch.save({k: v.cpu() for k, v in m.state_dict().items()}, path)
    m2 = synthetic.build_model(**CFG).cuda()
    T.load_reference_weights(m2, path)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
