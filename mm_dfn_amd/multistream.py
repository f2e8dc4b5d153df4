"""M-stream dialogue graph model: the MM-DFN graph dynamic-fusion stack over an arbitrary number of modality streams.

The reference wires exactly three streams ('a', 'v', 'l'; model_mm.py:97-106, model.py:1062-1154).  BASELINE.json's
config 5 -- "synthetic long-dialogue stress (L=512, 6 modalities x 512-d, 8 GCN layers)" -- is beyond what it can
express, so this module is the build's own generalisation of the same path with the same per-stream structure:

    per stream m:  X_m = Linear(D_m -> 200)(U_m)                      (model.py:1065,1094,1129)
    strip padding, dialogue-major                                      (model.py:553-565)
    F = MM_GCN over the M streams (adjacency tiles + M(M-1) cross-modal diagonals, GCNII_lyc with the LSTM gate)
                                                                       (model_mm.py:77-120, model_GCN.py:444-488)
    log_softmax(Linear(M*300 -> C)(relu(dropout(F))))                  (model.py:1328-1337)

The recurrent context / speaker-party encoders are specific to the trimodal model (speaker_weights 'a-v-l',
text-only context GRU) and stay in DialogueGNNModel.  The CPU oracle restates this composition
(oracle/mmdfn_oracle.py::forward_streams) from the same reference-pinned pieces.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .dialogue_model import _flat_index
from .mm_gcn import MM_GCN


class MultiStreamGraphModel(nn.Module):
    def __init__(self, D_streams, n_classes=6, nlayers=8, graph_hidden_size=100, hidden=200, dropout=0.5, lamda=0.5,
                 alpha=0.2, reason_flag=True, use_residue=True, modal_weight=1.0, n_speakers=2):
        super().__init__()
        M = len(D_streams)
        if not 2 <= M <= 9:
            raise ValueError("2..9 modality streams (the block-tile kernels index pairs of at most 9 streams)")
        self.dropout = dropout
        self.linears = nn.ModuleList([nn.Linear(int(d), hidden) for d in D_streams])
        self.graph_model = MM_GCN(a_dim=hidden, v_dim=hidden, l_dim=hidden, n_dim=hidden, nlayers=nlayers,
                                  nhidden=graph_hidden_size, nclass=n_classes, dropout=dropout, lamda=lamda, alpha=alpha,
                                  variant=True, return_feature=True, use_residue=use_residue, n_speakers=n_speakers,
                                  modals=["s%d" % m for m in range(M)], use_speaker=False, use_modal=False,
                                  reason_flag=reason_flag, modal_weight=modal_weight)
        width = (hidden + graph_hidden_size) if use_residue else graph_hidden_size
        self.dropout_ = nn.Dropout(dropout)
        self.smax_fc = nn.Linear(width * M, n_classes)

    def project(self, U_list, seq_lengths):
        """[(L, B, D_m)] -> (M, N, 200): the valid rows are selected first and projected afterwards (a Linear acts
        row by row, so this equals projecting the padded grid and stripping it, with fewer rows)."""
        L, B = U_list[0].shape[0], U_list[0].shape[1]
        idx = _flat_index([int(x) for x in seq_lengths], L, B, U_list[0].device)
        rows = [u.reshape(L * B, -1).index_select(0, idx) for u in U_list]
        ys = ops.linear_group(rows, [l.weight for l in self.linears], [l.bias for l in self.linears])
        return torch.stack(ys, 0)

    def forward(self, U_list, qmask, umask, seq_lengths, test_label=False):
        if len(U_list) != len(self.linears):
            raise ValueError("expected %d streams, got %d" % (len(self.linears), len(U_list)))
        with ops.flag_pool((U_list[0].shape[0], tuple(int(x) for x in seq_lengths))):
            feats = self.project(U_list, seq_lengths)
            fused = self.graph_model.forward_streams(feats, seq_lengths, qmask, test_label, stacked_out=True)
            log_prob = ops.head(fused, self.smax_fc.weight, self.smax_fc.bias, self.dropout_.p, self.training)
        return log_prob, None, None, None, None
