"""CPU: the vectorised CPU restatement (oracle/mmdfn_vectorised.py, bench.py's second cpu_baseline mode)
against the reference-structured oracle: log-probs and every gradient, ragged batches, both speaker-weight regimes."""
import pytest
import torch

import mmdfn_oracle as O
import mmdfn_vectorised as V
from mm_dfn_amd import synthetic


@pytest.mark.parametrize("cfg,lengths,seed", [
    (dict(B=4, L=20, P=3, C=6, nlayers=3, D_t=100, D_a=100, D_v=512), [20, 7, 1, 13], 5),
    (dict(B=3, L=9, P=9, C=7, nlayers=2, D_t=60, D_a=40, D_v=32), [9, 9, 4], 6),
])
@pytest.mark.parametrize("weights", [[3.0, 0.0, 1.0], [1.0, 2.0, 0.5], [0.0, 0.0, 0.0]])
def test_vectorised_restatement_equals_oracle(cfg, lengths, seed, weights):
    m = synthetic.build_model(**cfg)
    sd = synthetic.seeded_state_dict(m.state_dict(), seed)
    b = synthetic.make_batch(seed + 1, lengths=lengths, **cfg)
    ocfg = O.default_cfg(cfg["nlayers"], speaker_weights=weights, modal_weight=0.8)
    args = (b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])
    p1 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    p2 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want = O.forward(p1, *args, ocfg, engine="aten")
    w = torch.randn(want.shape, generator=torch.Generator().manual_seed(seed))
    (want * w).sum().backward()
    got = V.forward(p2, *args, ocfg)
    (got * w).sum().backward()
    assert float((got - want).abs().max()) < 2e-6
    for k in p1:
        g1, g2 = p1[k].grad, p2[k].grad
        if g1 is None or float(g1.abs().max()) == 0.0:
            assert g2 is None or float(g2.abs().max()) == 0.0, k
        else:
            assert float((g1 - g2).abs().max() / g1.abs().max()) < 5e-5, k


def test_party_plan_with_non_one_hot_speaker_mask():
    """Two speakers flagged on one utterance: the reference scatters speaker by speaker, the last one wins."""
    q = torch.zeros(5, 1, 3)
    q[0, 0, 0] = q[1, 0, 1] = q[2, 0, 0] = q[3, 0, 2] = 1
    q[2, 0, 2] = 1                                   # utterance 2 carries speakers 0 AND 2
    src, rank, sel = V.party_plan(q)
    assert src[:, 0, 0].tolist()[:2] == [0, 2] and src[:, 0, 2].tolist()[:2] == [2, 3]
    assert sel[2, 0].tolist() == [False, False, True]
    assert not sel[4, 0].any()
