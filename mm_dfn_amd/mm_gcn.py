"""MM_GCN: multimodal dialogue graph + GCNII stack (reference model_mm.py:44-180)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph_conv import GCNII_lyc


class MM_GCN(nn.Module):
    def __init__(self, a_dim, v_dim, l_dim, n_dim, nlayers, nhidden, nclass, dropout, lamda, alpha, variant,
                 return_feature, use_residue, new_graph='full', n_speakers=2, modals=None, use_speaker=True,
                 use_modal=False, reason_flag=False, modal_weight=1.0):
        super().__init__()
        self.return_feature = return_feature
        self.use_residue = use_residue
        self.new_graph = new_graph
        self.graph_net = GCNII_lyc(nfeat=n_dim, nlayers=nlayers, nhidden=nhidden, nclass=nclass, dropout=dropout,
                                   lamda=lamda, alpha=alpha, variant=variant, return_feature=return_feature,
                                   use_residue=use_residue, reason_flag=reason_flag)
        # parameters the reference constructs but never reaches on this path; kept
        # so that reference checkpoints load
        self.a_fc = nn.Linear(a_dim, n_dim)
        self.v_fc = nn.Linear(v_dim, n_dim)
        self.l_fc = nn.Linear(l_dim, n_dim)
        self.feature_fc = nn.Linear(n_dim * 3 + nhidden * 3, nhidden) if use_residue else nn.Linear(nhidden * 3, nhidden)
        self.final_fc = nn.Linear(nhidden, nclass)
        self.act_fn = nn.ReLU()
        self.dropout = dropout
        self.alpha = alpha
        self.lamda = lamda
        self.modals = modals
        self.modal_embeddings = nn.Embedding(3, n_dim)
        self.speaker_embeddings = nn.Embedding(n_speakers, n_dim)
        self.a_spk_embs = nn.Embedding(n_speakers, n_dim)
        self.v_spk_embs = nn.Embedding(n_speakers, n_dim)
        self.l_spk_embs = nn.Embedding(n_speakers, n_dim)
        self.use_speaker = use_speaker
        self.use_modal = use_modal
        self.modal_weight = modal_weight

    def _select(self, a, v, l, modals):
        modals = ''.join(modals)
        feats = [t for key, t in (('a', a), ('v', v), ('l', l)) if key in modals]
        if len(feats) < 2:
            raise NotImplementedError("MM_GCN needs at least two modalities")
        return feats

    def create_big_adj(self, a, v, l, dia_len, modals, modal_weight=1.0):
        """Returns the normalised adjacency as a BlockTileAdjacency (never dense)."""
        feats = self._select(a, v, l, modals)
        return ops.build_adjacency(torch.stack(feats, 0), dia_len, modal_weight)

    def forward(self, a, v, l, dia_len, qmask, test_label=False, stacked_out=False):
        if self.use_speaker and 'l' in self.modals:
            flat_q = torch.cat([qmask[:x, i, :] for i, x in enumerate(dia_len)], dim=0)
            l += self.speaker_embeddings(torch.argmax(flat_q, dim=-1))
        if self.use_modal:
            emb = self.modal_embeddings.weight
            if 'a' in self.modals:
                a += emb[0].reshape(1, -1)
            if 'v' in self.modals:
                v += emb[1].reshape(1, -1)
            if 'l' in self.modals:
                l += emb[2].reshape(1, -1)
        adj = self.create_big_adj(a, v, l, dia_len, self.modals, self.modal_weight)
        return self._graph(adj, qmask, test_label, stacked_out)

    def _graph(self, adj, qmask, test_label, stacked_out=False):
        M, N, D = adj.stacked_feats.shape
        features = self.graph_net(adj.stacked_feats.reshape(M * N, D), None, qmask, adj, test_label)
        # cat([F[:N], F[N:2N], ...], -1) (model_mm.py:117) as ONE strided copy: sliced, the backward is M zero-filled
        # (MN, 300) buffers, M slice copies and M-1 adds
        if stacked_out and self.return_feature:
            return features.view(M, N, -1)       # ops.head reads the M blocks in place: the concatenation never exists
        features = features.view(M, N, -1).permute(1, 0, 2).reshape(N, -1)
        if self.return_feature:
            return features
        return F.softmax(self.final_fc(features), dim=-1)

    def forward_streams(self, feats, dia_len, qmask=None, test_label=False, stacked_out=False):
        """M-stream entry: ``feats`` is a list of M (N, D) node-feature matrices -- or one (M, N, D) stack -- for
        2 <= M <= 9 modality streams of the same dialogues.  The reference builds the same graph for subsets of
        'avl' only (model_mm.py:97-106, M <= 3); the block-tile kernels take any M <= 9, which is what BASELINE
        config 5 (six streams) runs.  Output (N, M * (D + nhidden)), stream-major columns like model_mm.py:113."""
        if not torch.is_tensor(feats):
            feats = torch.stack(list(feats), 0)
        if feats.dim() != 3 or not 2 <= feats.shape[0] <= 9:
            raise ValueError("forward_streams expects 2..9 streams of (N, D) features")
        if self.use_speaker or self.use_modal:
            raise NotImplementedError("forward_streams: speaker / modality embeddings exist for 'avl' graphs only")
        return self._graph(ops.build_adjacency(feats, dia_len, self.modal_weight), qmask, test_label, stacked_out)

    def forward_stacked(self, feats, dia_len, qmask, test_label=False, stacked_out=False):
        """Same as forward(a, v, l, ...) for features that are already one (M, N, D) stack in modality order (what the
        fused encoder epilogue writes): skips the unbind / re-stack round trip and its select-backward zero fills.
        Not available with use_speaker / use_modal (they edit single modalities in place, model_mm.py:78-93)."""
        if self.use_speaker or self.use_modal:
            raise NotImplementedError("forward_stacked: use forward(a, v, l, ...) with use_speaker / use_modal")
        if feats.dim() != 3 or feats.shape[0] != len(''.join(self.modals)) or feats.shape[0] < 2:
            raise ValueError("forward_stacked expects a (len(modals), N, D) stack")
        return self._graph(ops.build_adjacency(feats, dia_len, self.modal_weight), qmask, test_label, stacked_out)
