"""Host logic next to the hot path: the pickle -> Dataset -> collate contract against the
reference's own dataloader classes (when /root/reference is present), the bucketing sampler, and the dual-patience
epoch loop against a literal restatement of run_train_erc.py:609-639."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from mm_dfn_amd import data as D
from mm_dfn_amd import train as T

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import ref_shim  # noqa: E402


@pytest.mark.parametrize("name", ["IEMOCAP", "MELD"])
def test_dataset_items_and_collate_match_reference(tmp_path, name):
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    ref_shim.install()
    import dataloader as R  # the reference's module
    p = str(tmp_path / "f.pkl")
    D.write_synthetic_pickle(p, dataset=name, n_train=7, n_test=3, n_speakers=2 if name == "IEMOCAP" else 9,
                             n_classes=6 if name == "IEMOCAP" else 7, D_t=20, D_a=12, D_v=16, seed=3)
    for train in (True, False):
        ours = getattr(D, name + "Dataset")(p, train)
        ref = getattr(R, name + "Dataset")(p, train)
        assert len(ours) == len(ref)
        items_o = [ours[i] for i in range(len(ours))]
        items_r = [ref[i] for i in range(len(ref))]
        for a, b in zip(items_o, items_r):
            for x, y in zip(a[:6], b[:6]):
                assert x.dtype == y.dtype and torch.equal(x, y)
            assert a[6] == b[6]
        bo, br = ours.collate_fn(items_o), ref.collate_fn(items_r)
        assert len(bo) == len(br) == 7
        for x, y in zip(bo[:6], br[:6]):
            assert x.shape == y.shape and torch.equal(x, y)
        assert bo[6] == br[6]
    if name == "MELD":
        assert ours.return_labels() == ref.return_labels()


def test_batch_layout_without_reference(tmp_path):
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=9, n_test=4, max_len=11, D_t=10, D_a=6, D_v=8, seed=1)
    tr, va, te = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.2)
    n = 0
    for textf, visuf, acouf, qmask, umask, label, vids in tr:
        L, B = textf.shape[:2]
        assert visuf.shape == (L, B, 8) and acouf.shape == (L, B, 6) and qmask.shape == (L, B, 2)
        assert umask.shape == (B, L) and label.shape == (B, L) and len(vids) == B
        lens = T.lengths_from_umask(umask)
        assert max(lens) == L
        for b, k in enumerate(lens):
            assert float(textf[k:, b].abs().sum()) == 0 and float(qmask[:k, b].sum()) == k
        n += B
    assert n == 9 - int(0.2 * 9) and sum(len(b[6]) for b in va) == int(0.2 * 9) and sum(len(b[6]) for b in te) == 4


def test_length_bucketing_visits_everything_once_and_cuts_padding():
    rs = np.random.RandomState(0)
    lengths = [int(x) for x in rs.randint(5, 111, size=200)]
    s = D.LengthBucketedBatchSampler(range(200), lengths, batch_size=16, bucket=4, seed=1)
    batches = list(s)
    assert sorted(i for b in batches for i in b) == list(range(200)) and len(batches) == len(s)
    pad = lambda bs: sum(max(lengths[i] for i in b) * len(b) for b in bs) / sum(lengths)
    plain = [list(range(i, min(i + 16, 200))) for i in range(0, 200, 16)]
    assert pad(batches) < 1.3 and pad(plain) > 1.6   # ~24 % padded rows instead of ~70 %
    s.set_epoch(1)
    assert list(s) != batches   # reshuffled per epoch


def _reference_rule(f1s, losses, patience):
    """run_train_erc.py:609-639 verbatim in spirit: returns (best_epoch, best_epoch2, epochs_run)."""
    best_epoch, best_epoch2, pat, pat2, best_f, best_l = -1, -1, 0, 0, 0, None
    ran = 0
    for e, (f, l) in enumerate(zip(f1s, losses)):
        ran += 1
        if e == 0 or best_f < f:
            pat = 0
            best_epoch, best_f = e, f
        else:
            pat += 1
        if best_l is None:
            best_l = l
            best_epoch2 = 0
        else:
            if l < best_l:
                best_epoch2, best_l = e, l
                pat2 = 0
            else:
                pat2 += 1
        if pat >= patience and pat2 >= patience:
            break
    return best_epoch, best_epoch2, ran


@pytest.mark.parametrize("seed", range(6))
def test_dual_patience_epoch_loop(seed):
    rs = np.random.RandomState(seed)
    n = 40
    f1 = list(np.round(50 + np.cumsum(rs.randn(n)) * 0.7, 2))
    loss = list(np.round(1.5 + np.cumsum(rs.randn(n)) * 0.02, 4))
    calls = []

    def run_pass(loader, epoch, train_flag):
        calls.append((loader, epoch, train_flag))
        if loader == "valid":
            return "", [], loss[epoch], 0.0, [], [], f1[epoch], []
        return "", [], 9.0, 60.0 + epoch, [], [], 70.0 + epoch, []

    out = T.fit(None, None, None, "train", "valid", "test", n_epochs=n, patience=3, valid_rate=0.1, run_pass=run_pass,
                log=None)
    be, be2, ran = _reference_rule(f1, loss, 3)
    assert out["epochs_run"] == ran and out["by_f1"]["epoch"] == be and out["by_loss"]["epoch"] == be2
    assert out["by_f1"]["test_fscore"] == 70.0 + be and out["by_loss"]["test_acc"] == 60.0 + be2
    assert calls[:3] == [("train", 0, True), ("valid", 0, False), ("test", 0, False)]
    # valid_rate == 0 selects on the test split (run_train_erc.py:611-612)
    out0 = T.fit(None, None, None, "train", "valid", "test", n_epochs=5, patience=3, valid_rate=0, run_pass=run_pass, log=None)
    assert out0["by_f1"]["epoch"] == 4 and out0["by_f1"]["eval_fscore"] == 74.0
