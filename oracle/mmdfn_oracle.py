"""CPU oracle for the MM-DFN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain torch-CPU fp32 restatement of the reference algorithm for the path
SURVEY.md §8 scopes (speaker-aware GRU encoders -> dialogue-graph adjacency ->
GCNII stack with LSTM-gated dynamic fusion -> head -> focal loss).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package (``mm_dfn_amd``) never does.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against the *real reference imported in the build container*
(``oracle/ref_shim.py``): ``tests/test_oracle_vs_reference.py`` compares every
function below with the reference live (skipped where /root/reference is
absent) and ``tests/golden/make_golden.py`` exports reference outputs as
fixtures that travel to the GPU box.

Everything is written functionally over a ``state_dict``-style ``params`` dict
whose keys are the reference's own, and deliberately keeps the
reference's *op structure* (dense (MN x MN) adjacency filled per dialogue,
P separate party-GRU passes, dense A.H product) so that timing it on the GPU
box's host gives a CPU number representative of the reference algorithm
(bench.py ``cpu_baseline.kind == "port"``).

Each function cites the reference lines it restates (paths under
/root/reference/code).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

COS_SHRINK = 0.99999  # model_mm.py:149,166


# ----------------------------------------------------------------------------
# ReLU probe (a test aid of the oracle itself, no counterpart in the reference).  At the BASELINE batch sizes a model
# holds ~10^6 ReLU pre-activations; one of them within fp32 rounding of zero is the normal case, and two correct fp32
# evaluations (this oracle's summation order, a device kernel's) then sit on different linear pieces of the network: same
# forward values, different gradients.  With a probe installed every ReLU site records its pre-activations, and the units
# listed in ``flips`` are evaluated on the OTHER side of their kink, so a test can differentiate the piece the device is
# on (tests/util.relu_flips_from_tap) instead of picking seeds that happen to avoid kinks.
# ----------------------------------------------------------------------------
class ReluProbe:
    def __init__(self, flips=None):
        self.pre = {}                      # site -> pre-activations (detached)
        self.flips = flips or {}           # site -> LongTensor (k, 2) of (row, column) positions to evaluate on the other side


_RELU_PROBE = None


def set_relu_probe(probe):
    global _RELU_PROBE
    prev, _RELU_PROBE = _RELU_PROBE, probe
    return prev


def relu_site(pre, site):
    probe = _RELU_PROBE
    if probe is None:
        return torch.relu(pre)
    probe.pre[site] = pre.detach()
    fl = probe.flips.get(site)
    if fl is None or len(fl) == 0:
        return torch.relu(pre)
    mask = pre.detach() > 0
    mask[fl[:, 0], fl[:, 1]] = ~mask[fl[:, 0], fl[:, 1]]
    return pre * mask.to(pre.dtype)


# ----------------------------------------------------------------------------
# recurrent cells (the equations torch.nn.GRU / torch.nn.LSTM document; the
# reference calls those modules at model.py:866,868 and model_GCN.py:433)
# ----------------------------------------------------------------------------
def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one GRU layer over the full padded length, zero h0.

    x: (T, B, I).  Gate order r, z, n.  n = tanh(W_in x + b_in + r*(W_hn h + b_hn)).
    Returns (T, B, H).
    """
    T, B, _ = x.shape
    H = w_hh.shape[1]
    gi_all = x.reshape(T * B, -1) @ w_ih.t() + b_ih
    gi_all = gi_all.reshape(T, B, 3 * H)
    h = x.new_zeros(B, H)
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gi = gi_all[t]
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 0)


def bigru2(x, params, prefix, dropout=0.0, training=False, engine="manual"):
    """2-layer bidirectional GRU, no packing (model.py:866,868,1082,1132).

    engine="manual": explicit equations above (the kernel's specification).
    engine="aten":   torch.nn.GRU with the same weights (what the reference
                     itself executes on CPU); used for the timed CPU baseline.
    """
    if engine == "aten":
        H = params[prefix + "weight_hh_l0"].shape[1]
        g = torch.nn.GRU(x.shape[-1], H, num_layers=2, bidirectional=True, dropout=dropout)
        g.train(training)
        names = [n for n, _ in g.named_parameters()]
        return torch.func.functional_call(g, {n: params[prefix + n] for n in names}, (x,))[0]
    cur = x
    for layer in range(2):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            tag = "l%d%s" % (layer, suffix)
            outs.append(gru_direction(cur, params[prefix + "weight_ih_" + tag], params[prefix + "weight_hh_" + tag],
                                      params[prefix + "bias_ih_" + tag], params[prefix + "bias_hh_" + tag], rev))
        cur = torch.cat(outs, -1)
        if layer == 0 and dropout > 0 and training:
            cur = F.dropout(cur, dropout, True)
    return cur


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One LSTM step, gate order i, f, g, o (nn.LSTM with seq_len 1, model_GCN.py:466)."""
    H = h.shape[1]
    g = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i = torch.sigmoid(g[:, :H])
    f = torch.sigmoid(g[:, H:2 * H])
    gg = torch.tanh(g[:, 2 * H:3 * H])
    o = torch.sigmoid(g[:, 3 * H:])
    c2 = f * c + i * gg
    h2 = o * torch.tanh(c2)
    return h2, c2


# ----------------------------------------------------------------------------
# speaker-party encoder (model.py:1070-1090 / 1101-1121 / 1134-1154)
# ----------------------------------------------------------------------------
def party_encode(X, qmask, params, dropout=0.0, training=False, engine="manual"):
    """For every speaker p: compact that speaker's utterances to the front of
    a zero (L, B, H) buffer, run the shared ``rnn_parties`` BiGRU over the
    full padded length, scatter the first k outputs back to the speaker's
    positions.  X: (L, B, H), qmask: (L, B, P).  Returns U_p: (L, B, H)."""
    L, B, Hd = X.shape
    P = qmask.shape[2]
    Xb = X.transpose(0, 1)            # (B, L, H)
    qb = qmask.transpose(0, 1)        # (B, L, P)
    idx = [[torch.nonzero(qb[b][:, p]).squeeze(-1) for p in range(P)] for b in range(B)]
    enc = []
    for p in range(P):
        rows = [F.pad(Xb[b][idx[b][p]], (0, 0, 0, L - idx[b][p].numel())) for b in range(B)]
        party = torch.stack(rows, 0)                               # (B, L, H), zeros past k
        enc.append(bigru2(party.transpose(0, 1), params, "rnn_parties.", dropout, training, engine).transpose(0, 1))
    out = []
    for b in range(B):
        acc = torch.zeros(L, enc[0].shape[-1], dtype=X.dtype)
        for p in range(P):
            k = idx[b][p].numel()
            if k > 0:
                acc = acc.index_add(0, idx[b][p], enc[p][b][:k])
        out.append(acc)
    return torch.stack(out, 0).transpose(0, 1)


def flatten_dialogues(enc, lengths):
    """simple_batch_graphify (model.py:553-565): (L,B,H) -> (N,H), dialogue-major."""
    return torch.cat([enc[:lengths[j], j, :] for j in range(enc.shape[1])], 0)


def encoders(params, U, qmask, lengths, U_a, U_v, cfg, training=False, engine="manual"):
    """model.py:1062-1154 + 1183-1209: per-modality projection, text context
    BiGRU, speaker-party BiGRU (all modalities), dialogue-major flatten.
    Returns [features_a, features_v, features_l], each (N, 200)."""
    w = cfg["speaker_weights"]
    p = cfg.get("dropout", 0.0)
    Xa = F.linear(U_a, params["linear_a.weight"], params["linear_a.bias"])
    ea = Xa + w[0] * party_encode(Xa, qmask, params, p, training, engine)
    Xv = F.linear(U_v, params["linear_v.weight"], params["linear_v.bias"])
    ev = Xv + w[1] * party_encode(Xv, qmask, params, p, training, engine)
    Xl = F.linear(U, params["linear_l.weight"], params["linear_l.bias"])
    ctx = bigru2(Xl, params, "lstm_l.", p, training, engine)
    el = ctx + w[2] * party_encode(Xl, qmask, params, p, training, engine)
    return [flatten_dialogues(e, lengths) for e in (ea, ev, el)]


# ----------------------------------------------------------------------------
# dialogue-graph adjacency (model_mm.py:122-180)
# ----------------------------------------------------------------------------
def angular_sim(cosv):
    """1 - acos(0.99999*cos)/pi (model_mm.py:149-150,166-167)."""
    return 1.0 - torch.acos(cosv * COS_SHRINK) / np.pi


def unit_rows(x):
    """x / ||x||_2 per row, no epsilon (model_mm.py:146-147)."""
    return x / torch.sqrt((x * x).sum(1, keepdim=True))


def raw_adjacency_dense(feats, dia_len, modal_weight=1.0):
    """The un-normalised dense (MN x MN) matrix A of model_mm.py:125-174."""
    M = len(feats)
    N = feats[0].shape[0]
    blocks = [[None] * M for _ in range(M)]
    start = 0
    # assemble functionally (no in-place slice-assign) but with the same content
    A = torch.zeros(M * N, M * N, dtype=feats[0].dtype)
    pieces = []
    for Li in dia_len:
        unit = [unit_rows(f[start:start + Li]) for f in feats]
        for m in range(M):
            for n in range(M):
                if m == n:
                    tile = angular_sim(unit[m] @ unit[m].t())
                    pieces.append((m * N + start, n * N + start, tile))
                else:
                    d = angular_sim((unit[m] * unit[n]).sum(1)) * modal_weight
                    pieces.append((m * N + start, n * N + start, torch.diag(d)))
        start += Li
    for r0, c0, t in pieces:
        pad = (c0, M * N - c0 - t.shape[1], r0, M * N - r0 - t.shape[0])
        A = A + F.pad(t, pad)
    return A


def create_big_adj(feats, dia_len, modal_weight=1.0):
    """Dense normalised adjacency D^-1/2 A D^-1/2 (model_mm.py:176-178)."""
    A = raw_adjacency_dense(feats, dia_len, modal_weight)
    r = torch.pow(A.sum(1), -0.5)
    return (r.unsqueeze(1) * A) * r.unsqueeze(0)


def adjacency_tiles(feats, dia_len, modal_weight=1.0):
    """The same Â in the packed block-tile layout the HIP path stores
    (DESIGN.md §layout): per dialogue i, per modality m one L_i x L_i tile,
    plus per unordered modality pair (m<n) one length-N diagonal.
    Returns (tiles flat [sum_i M*L_i^2], cross [npairs, N], rdeg [M, N])."""
    M = len(feats)
    N = feats[0].shape[0]
    S, C = [], {}
    start = 0
    deg = [[] for _ in range(M)]
    for Li in dia_len:
        unit = [unit_rows(f[start:start + Li]) for f in feats]
        s_i = [angular_sim(u @ u.t()) for u in unit]
        c_i = {}
        for m in range(M):
            for n in range(m + 1, M):
                c_i[(m, n)] = angular_sim((unit[m] * unit[n]).sum(1)) * modal_weight
        for m in range(M):
            d = s_i[m].sum(1)
            for n in range(M):
                if n != m:
                    d = d + c_i[(min(m, n), max(m, n))]
            deg[m].append(d)
        S.append(s_i)
        for k, v in c_i.items():
            C.setdefault(k, []).append(v)
        start += Li
    rdeg = torch.stack([torch.pow(torch.cat(d), -0.5) for d in deg], 0)  # (M, N)
    tiles = []
    start = 0
    for i, Li in enumerate(dia_len):
        for m in range(M):
            r = rdeg[m, start:start + Li]
            tiles.append(((r.unsqueeze(1) * S[i][m]) * r.unsqueeze(0)).reshape(-1))
        start += Li
    pairs = sorted(C.keys())
    cross = torch.stack([torch.cat(C[k]) * rdeg[k[0]] * rdeg[k[1]] for k in pairs], 0) if pairs else \
        feats[0].new_zeros(0, N)
    return torch.cat(tiles), cross, rdeg


def tiles_to_dense(tiles, cross, dia_len, M):
    """Expand the packed layout back to the dense (MN x MN) matrix (tests)."""
    N = int(sum(dia_len))
    A = torch.zeros(M * N, M * N, dtype=tiles.dtype)
    off = 0
    start = 0
    for Li in dia_len:
        for m in range(M):
            A[m * N + start:m * N + start + Li, m * N + start:m * N + start + Li] = \
                tiles[off:off + Li * Li].reshape(Li, Li)
            off += Li * Li
        start += Li
    k = 0
    ar = torch.arange(N)
    for m in range(M):
        for n in range(m + 1, M):
            A[m * N + ar, n * N + ar] = cross[k]
            A[n * N + ar, m * N + ar] = cross[k]
            k += 1
    return A


# ----------------------------------------------------------------------------
# GCNII layer + stack (model_GCN.py:176-189, 444-488)
# ----------------------------------------------------------------------------
def graph_convolution(x, adj, h0, lamda, alpha, l, weight):
    """variant=True, residual=False branch of GraphConvolution.forward."""
    theta = math.log(lamda / l + 1)
    hi = adj @ x
    support = torch.cat([hi, h0], 1)
    r = (1 - alpha) * hi + alpha * h0
    return theta * (support @ weight) + (1 - theta) * r


def gcnii_stack(x, adj, params, prefix, nlayers, lamda, alpha, dropout=0.0, training=False,
                reason_flag=True, use_residue=True):
    """GCNII_lyc.forward with an explicit adjacency (return_feature=True)."""
    x = F.dropout(x, dropout, training)
    h0 = relu_site(F.linear(x, params[prefix + "fcs.0.weight"], params[prefix + "fcs.0.bias"]), prefix + "fcs0")
    cur = F.dropout(h0, dropout, training)
    h = torch.zeros_like(cur)
    c = torch.zeros_like(cur)
    for i in range(nlayers):
        q = cur
        if reason_flag:
            h, c = lstm_cell(q, h, c, params[prefix + "rnn.weight_ih_l0"], params[prefix + "rnn.weight_hh_l0"],
                             params[prefix + "rnn.bias_ih_l0"], params[prefix + "rnn.bias_hh_l0"])
            cur = h
        cur = relu_site(graph_convolution(cur, adj, h0, lamda, alpha, i + 1,
                                          params[prefix + "convs.%d.weight" % i]), prefix + "conv%d" % i)
        cur = F.dropout(cur, dropout, training)
        if reason_flag:
            cur = cur + q
    return torch.cat([x, cur], -1) if use_residue else cur


def gcnii_deep(x, dia_len, params, prefix, nlayers, lamda=0.5, alpha=0.1, dropout=0.0, training=False,
               reason_flag=True, use_residue=True):
    """GCNII.forward (model_GCN.py:258-286): unimodal adjacency from x itself (create_big_adj :288-310 = the M = 1
    case), no dropout inside the layer loop (commented out at :275), one dropout after it (:279)."""
    adj = create_big_adj([x], dia_len)
    x = F.dropout(x, dropout, training)
    h0 = torch.relu(F.linear(x, params[prefix + "fcs.0.weight"], params[prefix + "fcs.0.bias"]))
    cur = F.dropout(h0, dropout, training)
    h = torch.zeros_like(cur)
    c = torch.zeros_like(cur)
    for i in range(nlayers):
        q = cur
        if reason_flag:
            h, c = lstm_cell(q, h, c, params[prefix + "rnn.weight_ih_l0"], params[prefix + "rnn.weight_hh_l0"],
                             params[prefix + "rnn.bias_ih_l0"], params[prefix + "rnn.bias_hh_l0"])
            cur = h
        cur = torch.relu(graph_convolution(cur, adj, h0, lamda, alpha, i + 1, params[prefix + "convs.%d.weight" % i]))
        if reason_flag:
            cur = cur + q
    cur = F.dropout(cur, dropout, training)
    return torch.cat([x, cur], -1) if use_residue else cur


def gated_attention_general(a, v, l, params, prefix="gatedatt."):
    """MMGatedAttention.forward, att_type='general', three modalities, eval (model.py:761-781)."""
    lin = lambda name, t: F.linear(t, params[prefix + name + ".weight"], params.get(prefix + name + ".bias"))
    ha, hv, hl = torch.tanh(lin("transform_a", a)), torch.tanh(lin("transform_v", v)), torch.tanh(lin("transform_l", l))
    z_av = torch.sigmoid(lin("transform_av", torch.cat([a, v, a * v], -1)))
    z_al = torch.sigmoid(lin("transform_al", torch.cat([a, l, a * l], -1)))
    z_vl = torch.sigmoid(lin("transform_vl", torch.cat([v, l, v * l], -1)))
    return torch.cat([z_av * ha + (1 - z_av) * hv, z_al * ha + (1 - z_al) * hl, z_vl * hv + (1 - z_vl) * hl], -1)


def forward_deepgcn(params, U, qmask, umask, lengths, U_a, U_v, cfg, training=False, engine="manual",
                    att_type="concat_subsequently"):
    """DialogueGNNModel.forward for graph_type='DeepGCN', multi_modal (model.py:1242-1290): the MM-DFN encoders,
    one unimodal GCNII per modality (lamda 0.5, alpha 0.1 hard-wired at :927-939), fusion, dropout, ReLU, head.
    att_type='gated' is restated for eval only (the module's own input dropout is inactive)."""
    feats = encoders(params, U, qmask, lengths, U_a, U_v, cfg, training, engine)
    e = [gcnii_deep(f, lengths, params, "graph_net_%s." % k, cfg["nlayers"], 0.5, 0.1, cfg.get("dropout", 0.0), training,
                    cfg.get("reason_flag", True)) for f, k in zip(feats, "avl")]
    fused = gated_attention_general(e[0], e[1], e[2], params) if att_type == "gated" else torch.cat(e, -1)
    return head(fused, params, cfg.get("dropout", 0.0), training)


def mm_gcn(feats, dia_len, params, cfg, training=False):
    """MM_GCN.forward, use_speaker/use_modal off (model_mm.py:77-120)."""
    adj = create_big_adj(feats, dia_len, cfg.get("modal_weight", 1.0))
    M = len(feats)
    N = feats[0].shape[0]
    out = gcnii_stack(torch.cat(feats, 0), adj, params, "graph_model.graph_net.", cfg["nlayers"], cfg["lamda"],
                      cfg["alpha"], cfg.get("dropout", 0.0), training, cfg.get("reason_flag", True))
    return torch.cat([out[m * N:(m + 1) * N] for m in range(M)], -1)


# ----------------------------------------------------------------------------
# head + loss (model.py:1328-1337, loss.py:14-34)
# ----------------------------------------------------------------------------
def head(feat, params, dropout=0.0, training=False):
    z = relu_site(F.dropout(feat, dropout, training), "head")
    return F.log_softmax(F.linear(z, params["smax_fc.weight"], params["smax_fc.bias"]), 1)


def focal_loss(log_prob, target, gamma=0.0, alpha=None, size_average=True):
    logpt = log_prob.gather(1, target.view(-1, 1)).view(-1)
    pt = logpt.detach().exp()
    if alpha is not None:
        logpt = logpt * alpha.gather(0, target.view(-1))
    loss = -1 * (1 - pt) ** gamma * logpt
    return loss.mean() if size_average else loss.sum()


def forward(params, U, qmask, umask, lengths, U_a, U_v, cfg, training=False, engine="manual"):
    """DialogueGNNModel.forward for the MM-DFN configuration -> log_prob (N, C)."""
    feats = encoders(params, U, qmask, lengths, U_a, U_v, cfg, training, engine)
    fused = mm_gcn(feats, lengths, params, cfg, training)
    return head(fused, params, cfg.get("dropout", 0.0), training)


def forward_streams(params, U_list, lengths, cfg, training=False):
    """M-stream composition of the restated pieces (per-stream projection model.py:1065, pad strip :553-565, MM_GCN
    model_mm.py:77-120, head model.py:1328-1337) for mm_dfn_amd.multistream.MultiStreamGraphModel.  The reference
    itself stops at three streams (model_mm.py:97-106): for M > 3 this is the oracle's own generalisation of functions
    that are reference-pinned at M = 2 and 3 (create_big_adj / gcnii_stack take any number of feature matrices)."""
    feats = [flatten_dialogues(F.linear(u, params["linears.%d.weight" % m], params["linears.%d.bias" % m]), lengths)
             for m, u in enumerate(U_list)]
    fused = mm_gcn(feats, lengths, params, cfg, training)
    return head(fused, params, cfg.get("dropout", 0.0), training)


def lengths_from_umask(umask):
    """run_train_erc.py:194."""
    return [int((umask[j] == 1).nonzero().tolist()[-1][0]) + 1 for j in range(len(umask))]


def flatten_labels(label, lengths):
    """run_train_erc.py:201."""
    return torch.cat([label[j][:lengths[j]] for j in range(len(label))])


def default_cfg(nlayers=2, dropout=0.0, speaker_weights=(3.0, 0.0, 1.0), lamda=0.5, alpha=0.2,
                modal_weight=1.0, reason_flag=True):
    return dict(nlayers=nlayers, dropout=dropout, speaker_weights=list(speaker_weights), lamda=lamda,
                alpha=alpha, modal_weight=modal_weight, reason_flag=reason_flag)
