"""Synthetic IEMOCAP/MELD-shaped batches and seeded weights.

Batches follow the reference collate contract (dataloader.py:18-34):
``textf (L,B,D_t), visuf (L,B,D_v), acouf (L,B,D_a), qmask (L,B,P), umask (B,L),
label (B,L)``, sequence-first, zero beyond each dialogue's length.  Everything
is drawn from ``np.random.RandomState`` (bit-stable across numpy versions) so
tests regenerate identical inputs on both sides of a parity check.
"""
import numpy as np
import torch

# BASELINE.json configs (dims: text / audio / visual)
CONFIGS = {
    "cfg1": dict(B=1, L=110, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512),
    "cfg2": dict(B=16, L=110, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512),
    "cfg3": dict(B=32, L=33, P=9, C=7, nlayers=4, D_t=600, D_a=300, D_v=342),
    "cfg4": dict(B=32, L=110, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512),  # per-GPU shard of 256
    # cfg2's batch at the reference's own IEMOCAP feature widths (run_train_erc.py:359-362: 1582-d IS10 audio, 342-d
    # denseface): neither is a multiple of 4, the projections run on row-padded operands (ops.py)
    "cfg2_refdims": dict(B=16, L=110, P=2, C=6, nlayers=2, D_t=100, D_a=1582, D_v=342),
}

# BASELINE.json config 5: long-dialogue stress, six modality streams of 512-d inputs, 8 GCN layers (graph hidden
# d = 100 as in the reference).  Per-GPU batch 8..32 dialogues; M = 6 is beyond the reference's
# trimodal wiring, so this runs through mm_dfn_amd.multistream.MultiStreamGraphModel.
STREAM_CONFIGS = {
    "cfg5": dict(B=8, L=512, P=2, C=6, nlayers=8, D_streams=[512] * 6),
    "cfg5_b32": dict(B=32, L=512, P=2, C=6, nlayers=8, D_streams=[512] * 6),
}


def make_lengths(rs, B, L, ragged, min_len=None):
    if not ragged:
        return [L] * B
    lo = max(1, L // 4) if min_len is None else min_len
    lens = rs.randint(lo, L + 1, size=B)
    lens[rs.randint(0, B)] = L
    return [int(x) for x in lens]


def make_batch(seed, B, L, P, C, D_t, D_a, D_v, ragged=False, lengths=None, device="cpu", **_):
    rs = np.random.RandomState(seed)
    lens = lengths if lengths is not None else make_lengths(rs, B, L, ragged)
    L = max(lens)
    B = len(lens)
    textf = rs.randn(L, B, D_t).astype(np.float32)
    visuf = rs.randn(L, B, D_v).astype(np.float32)
    acouf = rs.randn(L, B, D_a).astype(np.float32)
    spk = rs.randint(0, P, size=(L, B))
    lab = rs.randint(0, C, size=(B, L)).astype(np.int64)
    qmask = np.zeros((L, B, P), np.float32)
    umask = np.zeros((B, L), np.float32)
    for b, n in enumerate(lens):
        textf[n:, b] = 0
        visuf[n:, b] = 0
        acouf[n:, b] = 0
        qmask[np.arange(n), b, spk[:n, b]] = 1
        umask[b, :n] = 1
        lab[b, n:] = 0
    def t(a, feature=False):
        x = torch.from_numpy(a).to(device)
        from . import ops
        if x.is_cuda and ops.is_odd_feature_tensor(x, feature):
            # a feature width that is not a multiple of 4 (1582-d audio, 342-d visual): staged row-padded, as the data
            # pipeline does (ops.py "row padding"; same values, the modules see the (L, B, D) view)
            return ops.pad_rows(x)
        return x

    return dict(textf=t(textf, True), visuf=t(visuf, True), acouf=t(acouf, True), qmask=t(qmask), umask=t(umask), label=t(lab),
                lengths=[int(x) for x in lens])


def seeded_state_dict(reference_state, seed, scale=1.0):
    """Deterministic weights for every key of ``reference_state`` (an ordered
    name->tensor mapping): U(-s, s) with s = scale/sqrt(fan_in), drawn in sorted-key order."""
    rs = np.random.RandomState(seed)
    out = {}
    for k in sorted(reference_state.keys()):
        shape = tuple(reference_state[k].shape)
        fan = shape[-1] if len(shape) > 1 else max(shape[0], 1)
        s = scale / np.sqrt(fan)
        out[k] = torch.from_numpy(rs.uniform(-s, s, size=shape).astype(np.float32))
    return out


def build_model(D_t, D_a, D_v, P, C, nlayers, dropout=0.0, speaker_weights="3-0-1", att_type="concat_subsequently",
                graph_type="GDF", reason_flag=True, modals="avl", av_using_lstm=False, **_):
    """The MM-DFN configuration of run_train_erc.py:418-452 + the IEMOCAP script flags."""
    from .dialogue_model import DialogueGNNModel
    return DialogueGNNModel("LSTM", D_t, 150, 150, 100, 100, 100, 100, n_speakers=P, max_seq_len=200,
                            window_past=10, window_future=10, n_classes=C, dropout=dropout, no_cuda=False,
                            graph_type=graph_type, alpha=0.2, lamda=0.5, D_m_v=D_v, D_m_a=D_a, modals=modals,
                            att_type=att_type, av_using_lstm=av_using_lstm, Deep_GCN_nlayers=nlayers, dataset="IEMOCAP",
                            use_speaker=False, use_modal=False, reason_flag=reason_flag, multi_modal=True,
                            use_crn_speaker=True, speaker_weights=speaker_weights)


def make_stream_batch(seed, B, L, P, C, D_streams, ragged=False, lengths=None, device="cpu", **_):
    """M-stream counterpart of make_batch: ``streams[m]`` is (L, B, D_m), zero beyond each dialogue's length."""
    rs = np.random.RandomState(seed)
    lens = lengths if lengths is not None else make_lengths(rs, B, L, ragged)
    L = max(lens)
    B = len(lens)
    streams = [rs.randn(L, B, int(d)).astype(np.float32) for d in D_streams]
    spk = rs.randint(0, P, size=(L, B))
    lab = rs.randint(0, C, size=(B, L)).astype(np.int64)
    qmask = np.zeros((L, B, P), np.float32)
    umask = np.zeros((B, L), np.float32)
    for b, n in enumerate(lens):
        for s_ in streams:
            s_[n:, b] = 0
        qmask[np.arange(n), b, spk[:n, b]] = 1
        umask[b, :n] = 1
        lab[b, n:] = 0
    def t(a, feature=False):
        x = torch.from_numpy(a).to(device)
        from . import ops
        if x.is_cuda and ops.is_odd_feature_tensor(x, feature):
            # a feature width that is not a multiple of 4 (1582-d audio, 342-d visual): staged row-padded, as the data
            # pipeline does (ops.py "row padding"; same values, the modules see the (L, B, D) view)
            return ops.pad_rows(x)
        return x

    return dict(streams=[t(s_, True) for s_ in streams], qmask=t(qmask), umask=t(umask), label=t(lab),
                lengths=[int(x) for x in lens])


def build_stream_model(D_streams, C, nlayers, dropout=0.0, reason_flag=True, modal_weight=1.0, **_):
    from .multistream import MultiStreamGraphModel
    return MultiStreamGraphModel(D_streams, n_classes=C, nlayers=nlayers, dropout=dropout, lamda=0.5, alpha=0.2,
                                 reason_flag=reason_flag, modal_weight=modal_weight)
