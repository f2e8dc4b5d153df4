"""Import harness for the real MM-DFN reference (TEST INFRASTRUCTURE ONLY).

Only usable where /root/reference exists (the build container).  It never
travels to the GPU box.  It is used to (i) pin the restatement in
``oracle/mmdfn_oracle.py`` against the real reference and (ii) export the
golden vectors under ``tests/golden`` (``tests/golden/make_golden.py``).

Three in-process shims, no edits of the reference tree:
  1. stub ``torch_geometric`` (imported at model.py:11, unused on the GDF path)
  2. ``Tensor.cuda`` -> identity (hard ``.cuda()`` at model_mm.py:98,125)
  3. ``Tensor.__setitem__`` accepting a 2-D ndarray index the way torch<=1.4
     did (model_mm.py:168-172 relies on it).
"""
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("MMDFN_REFERENCE", "/root/reference")
REF_CODE = os.path.join(REF_ROOT, "code")

_installed = False


def available():
    return os.path.isfile(os.path.join(REF_CODE, "model.py"))


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    tg = types.ModuleType("torch_geometric")
    tgn = types.ModuleType("torch_geometric.nn")

    class _Stub(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tgn.RGCNConv = _Stub
    tgn.GraphConv = _Stub
    tg.nn = tgn
    sys.modules.setdefault("torch_geometric", tg)
    sys.modules.setdefault("torch_geometric.nn", tgn)

    torch.Tensor.cuda = lambda self, *a, **k: self

    _orig_setitem = torch.Tensor.__setitem__

    def _setitem(self, idx, val):
        if isinstance(idx, np.ndarray) and idx.ndim == 2:
            idx = tuple(torch.from_numpy(np.ascontiguousarray(r)) for r in idx)
        return _orig_setitem(self, idx, val)

    torch.Tensor.__setitem__ = _setitem
    if REF_CODE not in sys.path:
        sys.path.insert(0, REF_CODE)
    _installed = True


def modules():
    """Returns the reference's (model, model_mm, model_GCN, loss) modules."""
    install()
    import model as ref_model          # noqa: E402
    import model_mm as ref_mm          # noqa: E402
    import model_GCN as ref_gcn        # noqa: E402
    import loss as ref_loss            # noqa: E402
    return ref_model, ref_mm, ref_gcn, ref_loss


def build_reference_model(D_t, D_a, D_v, n_speakers, n_classes, nlayers, dropout=0.0,
                          speaker_weights="3-0-1", modals="avl", reason_flag=True,
                          att_type="concat_subsequently", alpha=0.2, lamda=0.5, graph_type="GDF", av_using_lstm=False):
    """The MM-DFN configuration of run_train_erc.py:418-452 + script flags."""
    ref_model, _, _, _ = modules()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref_model.DialogueGNNModel(
            "LSTM", D_t, 150, 150, 100, 100, 100, 100, n_speakers=n_speakers, max_seq_len=200,
            window_past=10, window_future=10, n_classes=n_classes, dropout=dropout, no_cuda=True,
            graph_type=graph_type, alpha=alpha, lamda=lamda, D_m_v=D_v, D_m_a=D_a, modals=modals,
            att_type=att_type, Deep_GCN_nlayers=nlayers, dataset="IEMOCAP",
            use_speaker=False, use_modal=False, reason_flag=reason_flag, multi_modal=True,
            use_crn_speaker=True, speaker_weights=speaker_weights, av_using_lstm=av_using_lstm)
    return m
