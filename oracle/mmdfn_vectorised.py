"""Vectorised CPU restatement of the MM-DFN hot path  --  TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

SURVEY.md section 8d asks for two CPU numbers on the GPU box's host: the reference-structured restatement
(``mmdfn_oracle.py``: dense (MN x MN) adjacency filled dialogue by dialogue, P separate party-GRU passes, dense
A.H product -- what the reference algorithm costs) and a *vectorised* one (this file: what a careful CPU
implementation of the same mathematics costs -- the conservative side of the GPU/CPU comparison).  Same ``params``
dict (the reference's state_dict keys), same results: ``tests/test_vectorised_cpu.py`` checks it against
``mmdfn_oracle`` (forward and every gradient), which is itself pinned against the reference.

Differences from the reference's op structure (each cites what it replaces):
  * the adjacency is never dense: per dialogue one padded (M, L, L) tile stack from ONE batched Gram product plus the
    M(M-1) cross-modal diagonals as (M, M, L) vectors (model_mm.py:122-180 fills a dense (MN)^2 matrix with
    B (M + M(M-1)) slice-assigns);
  * A.H is a batched (B M) x [L x L] . [L x d] product plus the diagonal terms (model_GCN.py:178 multiplies the dense
    matrix);
  * the 3 P speaker-party GRU passes (model.py:1082,1112,1145) run as ONE batch (the module is shared), driven by a
    cumulative-sum gather/scatter plan instead of B P Python slice loops (model.py:1076-1087); modalities whose
    speaker weight is 0 are skipped (the reference multiplies their result by 0.0).
Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

import mmdfn_oracle as O


def party_plan(qmask):
    """qmask (L, B, P) -> (src, rank, sel): src[k,b,p] = time index of speaker p's k-th utterance in dialogue b
    (L = none), rank[t,b,p] = position of utterance t among speaker p's utterances, sel = the scatter mask (with a
    non-one-hot qmask the reference scatters speaker by speaker, so the last speaker wins)."""
    L, B, P = qmask.shape
    mask = qmask != 0
    rank = torch.cumsum(mask.to(torch.int64), 0) - 1
    t_grid = torch.arange(L, device=qmask.device).view(L, 1, 1).expand(L, B, P)
    src = torch.full((L + 1, B, P), L, dtype=torch.int64, device=qmask.device)
    src.scatter_(0, torch.where(mask, rank, torch.full_like(rank, L)), t_grid)
    src = src[:L]
    later = torch.flip(torch.cumsum(torch.flip(mask, [2]).to(torch.int64), 2), [2]) - mask.to(torch.int64)
    sel = mask & (later == 0)
    return src, rank.clamp_(min=0), sel


def party_gather(X, plan):
    """X (Mn, L, B, H) -> (L, Mn*B*P, H): every speaker's utterances compacted to the front, zero rows behind."""
    src, _, _ = plan
    L, B, P = src.shape
    Mn, H = X.shape[0], X.shape[-1]
    Xp = torch.cat([X, X.new_zeros(Mn, 1, B, H)], 1)
    g_idx = src.view(1, L, B, P, 1).expand(Mn, L, B, P, H)
    S = Xp.unsqueeze(3).expand(Mn, L + 1, B, P, H).gather(1, g_idx)
    return S.permute(1, 0, 2, 3, 4).reshape(L, Mn * B * P, H)


def party_scatter(E, plan, Mn):
    """(L, Mn*B*P, H) party encodings -> (Mn, L, B, H) at the speakers' own positions."""
    _, rank, sel = plan
    L, B, P = rank.shape
    H = E.shape[-1]
    E = E.view(L, Mn, B, P, H).permute(1, 0, 2, 3, 4)
    s_idx = rank.view(1, L, B, P, 1).expand(Mn, L, B, P, H)
    return (E.gather(1, s_idx) * sel.view(1, L, B, P, 1).to(E.dtype)).sum(3)


def encoders(params, U, qmask, lengths, U_a, U_v, cfg, training=False):
    """model.py:1062-1154 + 1183-1209 -> (3, N, 200) in the order a, v, l."""
    w = cfg["speaker_weights"]
    p = cfg.get("dropout", 0.0)
    Xa = F.linear(U_a, params["linear_a.weight"], params["linear_a.bias"])
    Xv = F.linear(U_v, params["linear_v.weight"], params["linear_v.bias"])
    Xl = F.linear(U, params["linear_l.weight"], params["linear_l.bias"])
    ctx = O.bigru2(Xl, params, "lstm_l.", p, training, "aten")
    bases = [Xa, Xv, ctx]
    act = [i for i in range(3) if w[i] != 0.0]
    if act:
        plan = party_plan(qmask)
        S = party_gather(torch.stack([(Xa, Xv, Xl)[i] for i in act], 0), plan)
        Up = party_scatter(O.bigru2(S, params, "rnn_parties.", p, training, "aten"), plan, len(act))
        for slot, i in enumerate(act):
            bases[i] = bases[i] + w[i] * Up[slot]
    L, B = U.shape[0], U.shape[1]
    idx = torch.from_numpy(np.concatenate([np.arange(int(n), dtype=np.int64) * B + j for j, n in enumerate(lengths)]))
    return torch.stack([e.reshape(L * B, -1).index_select(0, idx) for e in bases], 0)


class PaddedGraph:
    """Normalised adjacency of a batch as padded per-dialogue blocks: tiles (B, M, L, L), cross (B, M, M, L) (zero on the
    m == n diagonal), plus the flat <-> padded row maps."""

    def __init__(self, feats, lengths, modal_weight=1.0):
        M, N, D = feats.shape
        B, L = len(lengths), max(lengths)
        lens = torch.tensor(lengths)
        valid = torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)                       # (B, L)
        self.flat_of = torch.nonzero(valid.reshape(-1)).squeeze(1)                      # padded slot of flat row r
        self.B, self.L, self.M, self.N = B, L, M, N
        pad = feats.new_zeros(M, B * L, D).index_copy(1, self.flat_of, feats).view(M, B, L, D).transpose(0, 1)
        norm = torch.sqrt((pad * pad).sum(-1, keepdim=True))
        unit = torch.where(valid.view(B, 1, L, 1), pad / torch.where(norm > 0, norm, torch.ones_like(norm)),
                           torch.zeros_like(pad))                                       # (B, M, L, D)
        vm = valid.view(B, 1, L, 1) & valid.view(B, 1, 1, L)
        S = torch.where(vm, O.angular_sim(unit @ unit.transpose(2, 3)), unit.new_zeros(()))        # (B, M, L, L)
        C = O.angular_sim(torch.einsum("bmld,bnld->bmnl", unit, unit)) * modal_weight             # (B, M, M, L)
        off = ~torch.eye(M, dtype=torch.bool).view(1, M, M, 1)
        C = torch.where(off & valid.view(B, 1, 1, L), C, C.new_zeros(()))
        deg = S.sum(3) + C.sum(2)                                                        # (B, M, L)
        r = torch.where(valid.view(B, 1, L), torch.pow(torch.where(deg > 0, deg, torch.ones_like(deg)), -0.5),
                        deg.new_zeros(()))
        self.tiles = r.unsqueeze(3) * S * r.unsqueeze(2)
        self.cross = r.unsqueeze(2) * C * r.unsqueeze(1)

    def propagate(self, H):
        """H (M*N, d) flat modality-major rows -> A_hat . H in the same layout."""
        M, N, B, L = self.M, self.N, self.B, self.L
        d = H.shape[1]
        Hp = H.new_zeros(M, B * L, d).index_copy(1, self.flat_of, H.view(M, N, d)).view(M, B, L, d).transpose(0, 1)
        out = self.tiles @ Hp + torch.einsum("bmnl,bnld->bmld", self.cross, Hp)
        return out.transpose(0, 1).reshape(M, B * L, d).index_select(1, self.flat_of).reshape(M * N, d)


def gcnii_stack(x, graph, params, prefix, nlayers, lamda, alpha, dropout=0.0, training=False, reason_flag=True):
    """GCNII_lyc.forward (model_GCN.py:444-488) over the padded block graph."""
    x = F.dropout(x, dropout, training)
    h0 = torch.relu(F.linear(x, params[prefix + "fcs.0.weight"], params[prefix + "fcs.0.bias"]))
    cur = F.dropout(h0, dropout, training)
    h = torch.zeros_like(cur)
    c = torch.zeros_like(cur)
    for i in range(nlayers):
        q = cur
        if reason_flag:
            h, c = O.lstm_cell(q, h, c, params[prefix + "rnn.weight_ih_l0"], params[prefix + "rnn.weight_hh_l0"],
                               params[prefix + "rnn.bias_ih_l0"], params[prefix + "rnn.bias_hh_l0"])
            cur = h
        theta = math.log(lamda / (i + 1) + 1)
        hi = graph.propagate(cur)
        out = theta * (torch.cat([hi, h0], 1) @ params[prefix + "convs.%d.weight" % i]) \
            + (1 - theta) * ((1 - alpha) * hi + alpha * h0)
        cur = F.dropout(torch.relu(out), dropout, training)
        if reason_flag:
            cur = cur + q
    return torch.cat([x, cur], -1)


def forward(params, U, qmask, umask, lengths, U_a, U_v, cfg, training=False):
    """DialogueGNNModel.forward for the MM-DFN configuration -> log_prob (N, C)."""
    feats = encoders(params, U, qmask, lengths, U_a, U_v, cfg, training)
    M, N, D = feats.shape
    graph = PaddedGraph(feats, lengths, cfg.get("modal_weight", 1.0))
    out = gcnii_stack(feats.reshape(M * N, D), graph, params, "graph_model.graph_net.", cfg["nlayers"], cfg["lamda"],
                      cfg["alpha"], cfg.get("dropout", 0.0), training, cfg.get("reason_flag", True))
    fused = out.view(M, N, -1).permute(1, 0, 2).reshape(N, -1)
    return O.head(fused, params, cfg.get("dropout", 0.0), training)
