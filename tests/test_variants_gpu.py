"""The two flags of the GDF path that the reference driver exposes next to the MM-DFN scripts' setting -- bimodal graphs
(--modals av | al | vl; model.py:851-868, model_mm.py:97-106) and context GRUs for the audio / visual streams
(--av_using_lstm; model.py:854-860,1067-1068,1096-1097) -- against goldens exported from the reference
(tests/golden/make_golden.py export_variants): eval log-probabilities, loss, every live gradient's digest, full gradients
of a few parameters, and (CPU) the state_dict key / shape lists."""
import os

import numpy as np
import pytest
import torch

from mm_dfn_amd import FocalLoss, synthetic, train
from test_oracle_golden import GOLD, _digest

DEV = "cuda"
CASES = {
    "av": ("av", False, "3-0-1", dict(B=3, L=20, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512), 111, [20, 13, 7]),
    "al": ("al", False, "3-0-1", dict(B=3, L=20, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512), 112, [20, 13, 7]),
    "vl": ("vl", False, "1-2-1", dict(B=4, L=17, P=3, C=7, nlayers=3, D_t=100, D_a=100, D_v=342), 113, [17, 9, 1, 12]),
    "avl_lstm": ("avl", True, "3-0-1", dict(B=3, L=20, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512), 114, [20, 13, 7]),
    "al_lstm": ("al", True, "1-1-1", dict(B=2, L=15, P=2, C=6, nlayers=2, D_t=100, D_a=1582, D_v=342), 115, [15, 6]),
}


def build(name, dropout=0.0):
    modals, lstm, sw, cfg, seed, lengths = CASES[name]
    m = synthetic.build_model(dropout=dropout, speaker_weights=sw, modals=modals, av_using_lstm=lstm, **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), seed))
    return m, synthetic.make_batch(seed + 1, lengths=lengths, **cfg)


@pytest.mark.parametrize("name", sorted(CASES))
def test_state_dict_keys_of_the_variants_match_the_reference(name):
    m, _ = build(name)
    want = [ln.split() for ln in open(os.path.join(GOLD, "state_dict_keys_variant_%s.txt" % name)).read().splitlines() if ln]
    got = [[k] + [str(d) for d in v.shape] for k, v in m.state_dict().items()]
    assert sorted(map(tuple, got)) == sorted(map(tuple, want))


def test_unsupported_variants_say_so():
    from mm_dfn_amd.dialogue_model import DialogueGNNModel
    mk = lambda **kw: DialogueGNNModel("LSTM", 100, 150, 150, 100, 100, 100, 100, n_speakers=2, max_seq_len=200, window_past=10,
                                       window_future=10, n_classes=6, graph_type="GDF", multi_modal=True, use_crn_speaker=True,
                                       speaker_weights="3-0-1", Deep_GCN_nlayers=2, reason_flag=True, **kw)
    with pytest.raises(NotImplementedError):
        mk(modals="a", att_type="concat_subsequently")
    with pytest.raises(NotImplementedError):
        mk(modals="al", att_type="mfn")
    mk(modals="al", att_type="concat_subsequently", av_using_lstm=True)


@pytest.mark.gpu
@pytest.mark.parametrize("truncate", [False, True])
@pytest.mark.parametrize("name", sorted(CASES))
def test_variants_against_reference_goldens(name, truncate):
    from mm_dfn_amd import gru as fused
    g = np.load(os.path.join(GOLD, "variants.npz"), allow_pickle=False)
    prev, fused.TRUNCATE = fused.TRUNCATE, truncate
    try:
        m, b = build(name)
        m = m.to(DEV).eval()
        run = lambda: m(b["textf"].to(DEV), b["qmask"].to(DEV), b["umask"].to(DEV), b["lengths"], b["acouf"].to(DEV),
                        b["visuf"].to(DEV))[0]
        with torch.no_grad():
            logp = run()
        assert np.abs(logp.cpu().numpy() - g[name + "/log_prob"]).max() < 1e-4
        m.train()           # dropout p = 0
        logp = run()
        loss = FocalLoss(gamma=0.5)(logp, train.flatten_labels(b["label"].to(DEV), b["lengths"]))
        assert abs(loss.item() - float(g[name + "/loss"])) < 1e-5
        train.backward(loss)
    finally:
        fused.TRUNCATE = prev
    grads = {k: p.grad for k, p in m.named_parameters()}
    live = [str(x) for x in g[name + "/live_params"]]
    for k in live:
        want = g[name + "/gd/" + k]
        assert grads[k] is not None, k
        assert abs(_digest(grads[k])[1] - want[1]) / (want[1] + 1e-12) < 2e-4, k
    for k, gr in grads.items():
        if k not in live:
            assert gr is None or float(gr.abs().max()) == 0.0, k
    for k in [x[len(name) + 3:] for x in g.files if x.startswith(name + "/g/")]:
        want = g[name + "/g/" + k]
        assert np.abs(grads[k].cpu().numpy() - want).max() / np.abs(want).max() < 2e-4, k
