"""DialogueGNNModel, MM-DFN configuration (reference model.py:784-1407).

Same constructor and ``forward(U, qmask, umask, seq_lengths, U_a, U_v,
test_label)`` signature and the same 86 ``state_dict`` keys as the reference,
restricted to the configuration MM-DFN trains (base_model='LSTM',
multi_modal=True, graph_type='GDF', use_crn_speaker=True).  Other ablation
branches of the reference constructor raise NotImplementedError.

Host-side structure differs from the reference on purpose (MI355X-first):
  * the B*P*2*3 Python slice-assign loops of the speaker-party encoder
    (model.py:1076-1087) become one device-side gather / scatter driven by a
    prefix sum over qmask, and the 3*P separate ``rnn_parties`` calls
    (model.py:1082,1112,1145) become ONE batched call (weights are shared);
  * padding is stripped with one index_select producing the (M, N, 200)
    modality-major stack the graph kernels consume (model.py:553-565, :98);
  * the adjacency is never dense (layout.py / csrc/adjacency.hip).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gru as fused_gru
from . import ops
from .fusion import MFN, MMGatedAttention
from .graph_conv import GCNII
from .layout import IndexScope, PinnedLRU
from .mm_gcn import MM_GCN

_FLAT_CACHE = PinnedLRU(64)


def _flat_index(lengths, L, B, device):
    """Row ids t*B+b of the (L*B) padded grid in dialogue-major order (simple_batch_graphify)."""
    def rows(lens):
        lens = np.asarray([int(n) for n in lens], dtype=np.int64)
        start = np.cumsum(lens) - lens
        t = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(start, lens)             # position inside the dialogue
        return t * B + np.repeat(np.arange(lens.size, dtype=np.int64), lens)
    scope = IndexScope.current()
    if scope is not None:             # a captured step replayed for other length lists owns (and rewrites) its index tensors
        return scope.tensor(("flat", L, B, str(device)), lengths, rows, device)
    return _FLAT_CACHE.get((tuple(lengths), L, B, str(device)), lambda: torch.from_numpy(rows(lengths)).to(device))


def _flat_inverse(lengths, L, B, device):
    """(L * B,) int64: the row of padded-grid position t*B+b in the dialogue-major stripped order, -1 for padding (the inverse of
    _flat_index; the combine stage's backward writes its outputs by destination with it)."""
    def rows(lens):
        lens = np.asarray([int(n) for n in lens], dtype=np.int64)
        start = np.cumsum(lens) - lens
        t = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(start, lens)
        inv = np.full(L * B, -1, dtype=np.int64)
        inv[t * B + np.repeat(np.arange(lens.size, dtype=np.int64), lens)] = np.arange(int(lens.sum()), dtype=np.int64)
        return inv
    scope = IndexScope.current()
    if scope is not None:
        return scope.tensor(("flat_inv", L, B, str(device)), lengths, rows, device)
    return _FLAT_CACHE.get((tuple(lengths), L, B, str(device), "inv"), lambda: torch.from_numpy(rows(lengths)).to(device))


class _Scalar(nn.Module):
    def __init__(self, i, o, bias):
        super().__init__()
        self.scalar = nn.Linear(i, o, bias=bias)


class _Transform(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.transform = nn.Linear(i, o, bias=True)


class _MlpAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(2 * dim).uniform_(-0.05, 0.05))
        self.w_k = nn.Linear(dim, dim)
        self.w_q = nn.Linear(dim, dim)
        self.proj = nn.Linear(dim, dim)


class _EdgeAttentionParams(nn.Module):
    """Parameters of the reference's MaskedEdgeAttention (model.py:420-445): constructed
    by the reference for every graph type, used only by graph_type='relation'."""

    def __init__(self, dim, max_seq_len):
        super().__init__()
        self.scalar = nn.Linear(dim, max_seq_len, bias=False)
        self.matchatt = _Transform(dim, dim)
        self.simpleatt = _Scalar(dim, 1, False)
        self.att = _MlpAttention(dim)


PROJECT_THEN_GATHER_ROWS = 0


class DialogueGNNModel(nn.Module):

    def __init__(self, base_model, D_m, D_g, D_p, D_e, D_h, D_a, graph_hidden_size, n_speakers, max_seq_len,
                 window_past, window_future, n_classes=7, listener_state=False, context_attention='simple',
                 dropout_rec=0.5, dropout=0.5, nodal_attention=True, avec=False, no_cuda=False,
                 graph_type='relation', use_topic=False, alpha=0.1, lamda=0.5, multiheads=6,
                 graph_construct='direct', use_GCN=False, use_residue=True, dynamic_edge_w=False, D_m_v=512,
                 D_m_a=100, modals='avl', att_type='gated', av_using_lstm=False, Deep_GCN_nlayers=64,
                 dataset='IEMOCAP', use_speaker=True, use_modal=False, reason_flag=False, multi_modal=True,
                 use_crn_speaker=False, speaker_weights='1-1-1', modal_weight=1.0):
        super().__init__()
        if base_model != 'LSTM' or not multi_modal or graph_type not in ('GDF', 'DeepGCN'):
            raise NotImplementedError("mm_dfn_amd implements the MM-DFN hot path (graph_type='GDF') and its unimodal-graph "
                                      "sibling 'DeepGCN' only, with base_model='LSTM', multi_modal=True (got %r, %r, %r)"
                                      % (base_model, multi_modal, graph_type))
        if graph_type == 'GDF' and att_type not in ('concat_subsequently', 'mfn'):
            raise NotImplementedError("att_type must be 'concat_subsequently' (the MM-DFN scripts) or 'mfn'")
        if graph_type == 'DeepGCN' and att_type not in ('concat_subsequently', 'gated'):
            raise NotImplementedError("DeepGCN: att_type must be 'concat_subsequently' or 'gated'")
        present = [m for m in 'avl' if m in modals]
        if sorted(modals) != sorted(present) or len(present) < 2:
            raise NotImplementedError("modals must name two or three of 'a', 'v', 'l' (model_mm.py:97-106), got %r" % (modals,))
        if len(present) < 3 and (graph_type != 'GDF' or att_type != 'concat_subsequently'):
            raise NotImplementedError("two-modality graphs: graph_type='GDF' with att_type='concat_subsequently' only")
        if 2 * D_e != 200:
            raise NotImplementedError("the reference hard-codes a 200-wide encoder (model.py:847-849,1074); D_e must be 100")
        self.base_model = base_model
        self.no_cuda = no_cuda
        self.graph_type = graph_type
        self.alpha = alpha
        self.lamda = lamda
        self.dropout = dropout
        self.use_residue = use_residue
        self.return_feature = True
        self.modals = [x for x in modals]
        self.present = present                  # modalities in the order the graph stacks them: a, v, l (model_mm.py:98-104)
        self.av_using_lstm = av_using_lstm
        self.use_speaker = use_speaker
        self.use_modal = use_modal
        self.att_type = att_type
        self.reason_flag = reason_flag
        self.multi_modal = multi_modal
        self.n_speakers = n_speakers
        self.use_crn_speaker = use_crn_speaker
        self.speaker_weights = list(map(float, speaker_weights.split('-')))
        self.modal_weight = modal_weight
        self.dataset = dataset
        self.window_past = window_past
        self.window_future = window_future
        self.nodal_attention = nodal_attention

        hidden = 2 * D_e
        # model.py:851-868: only the modules of the modalities in ``modals`` exist (and with them the state_dict keys); the
        # audio / visual streams get a context GRU of their own with av_using_lstm (keys lstm_a.* / lstm_v.*)
        gru = lambda: nn.GRU(input_size=hidden, hidden_size=D_e, num_layers=2, bidirectional=True, dropout=dropout)
        if 'a' in present:
            self.linear_a = nn.Linear(D_m_a, hidden)
            if av_using_lstm:
                self.lstm_a = gru()
        if 'v' in present:
            self.linear_v = nn.Linear(D_m_v, hidden)
            if av_using_lstm:
                self.lstm_v = gru()
        if 'l' in present:
            self.linear_l = nn.Linear(D_m, hidden)
            self.lstm_l = gru()
        self.rnn_parties = gru()
        self.att_model = _EdgeAttentionParams(hidden, max_seq_len)
        if graph_type == 'DeepGCN':
            # one unimodal GCNII per modality, lamda / alpha hard-wired (model.py:922-941)
            mk = lambda: GCNII(nfeat=hidden, nlayers=Deep_GCN_nlayers, nhidden=graph_hidden_size, nclass=n_classes,
                               dropout=dropout, lamda=0.5, alpha=0.1, variant=True, return_feature=True,
                               use_residue=use_residue, reason_flag=reason_flag)
            self.graph_net_a, self.graph_net_v, self.graph_net_l = mk(), mk(), mk()
        else:
            self.graph_model = MM_GCN(a_dim=hidden, v_dim=hidden, l_dim=hidden, n_dim=hidden, nlayers=Deep_GCN_nlayers,
                                      nhidden=graph_hidden_size, nclass=n_classes, dropout=dropout, lamda=lamda,
                                      alpha=alpha, variant=True, return_feature=True, use_residue=use_residue,
                                      n_speakers=n_speakers, modals=self.modals, use_speaker=use_speaker,
                                      use_modal=use_modal, reason_flag=reason_flag, modal_weight=modal_weight)
        self.gatedatt = MMGatedAttention(hidden + graph_hidden_size, graph_hidden_size, att_type='general')
        self.dropout_ = nn.Dropout(dropout)
        if att_type == 'mfn':
            self.mfn = MFN()                                      # model.py:991-994
            self.smax_fc = nn.Linear(400, n_classes)
        elif att_type == 'gated':
            self.smax_fc = nn.Linear(100 * len(self.modals), n_classes)   # model.py:985-989 (three modalities)
        else:
            width = (hidden + graph_hidden_size) if use_residue else graph_hidden_size
            self.smax_fc = nn.Linear(width * len(self.modals), n_classes)

    # ------------------------------------------------------------------ encoders
    def _run_grus(self, xs, grus):
        """Always the fused HIP recurrence (raises on CPU tensors: there is no fallback)."""
        return fused_gru.bigru2(xs, grus, self.dropout, self.training)

    def encode(self, U, qmask, seq_lengths, U_a, U_v):
        """Projection + context BiGRUs (text; audio / visual too with av_using_lstm) + speaker-party BiGRU (every modality with
        a non-zero speaker weight) -> (M, N, 200) dialogue-major stack of the modalities in ``modals``, order a, v, l
        (model.py:1062-1154,1183-1209).  All GRUs are independent and share every recurrence launch; the party gather /
        scatter / pad-strip are the fused K3/K4 kernels (csrc/encoder_glue.hip)."""
        present = self.present
        raw = {'a': U_a, 'v': U_v, 'l': U}
        lin = {m: getattr(self, "linear_" + m) for m in present}
        L, B = raw[present[0]].shape[0], raw[present[0]].shape[1]
        P = qmask.shape[2]
        wts = dict(zip('avl', self.speaker_weights))
        # modalities with a context GRU of their own (model.py:1067-1068,1096-1097,1132)
        ctx_mods = [m for m in present if m == 'l' or self.av_using_lstm]
        ctx_grus = [getattr(self, "lstm_" + m) for m in ctx_mods]
        # only modalities with a non-zero speaker weight go through the party encoder: the reference also encodes the others
        # and multiplies the result by 0 (model.py:1090,1121,1154 with '3-0-1'), which changes neither the features nor any
        # gradient
        act = [m for m in present if self.use_crn_speaker and wts[m] != 0.0]
        # valid-length launches of the party encoder (gru.py TRUNCATE): their all-padding sequence depends on the weights only
        # and starts on a side stream, under the gather
        truncate = bool(act) and P <= 16 and P * L <= 2048 and fused_gru.wants_truncation(len(ctx_mods) * B + len(act) * B * P)
        use_table = truncate and fused_gru.USE_TABLE
        if use_table:
            fork = torch.cuda.Event()
            fork.record()
        proj = ops.linear_group([raw[m] for m in present], [lin[m].weight for m in present], [lin[m].bias for m in present])
        X = dict(zip(present, proj))
        idx = _flat_index([int(x) for x in seq_lengths], L, B, proj[0].device)
        inv = _flat_inverse([int(x) for x in seq_lengths], L, B, proj[0].device)
        table = None
        if use_table:
            # (launched behind the projections in program order, so the main stream's first kernel is not held up by the
            # side branch's launches; ordered behind the point BEFORE them)
            table = fused_gru.start_party_table(self.rnn_parties, L, after=fork)
        weights = [wts[m] if m in act else 0.0 for m in present]
        if not act:
            outs = self._run_grus([X[m] for m in ctx_mods], ctx_grus) if ctx_mods else []
            base = dict(X)
            base.update(zip(ctx_mods, outs))
            rank = torch.full((L, B, P), -1, dtype=torch.int32, device=proj[0].device)
            return ops.party_combine([base[m] for m in present], None, rank, idx, [0.0] * len(present), inv)
        party = None
        if len(act) * L * B * P >= PROJECT_THEN_GATHER_ROWS or P >= 4:
            # first party-GRU layer: gather(X) W_ih^T + b == gather(X W_ih^T) + b (padding rows = b), so the input
            # contraction runs over the n_act*L*B projected utterances, not over the n_act*L*P*B party rows of
            # which all but one in P are zero (the reference projects every padded party row, model.py:1082).
            # Pays off once the party batch is large (measured: cfg4 2.27 -> 2.22 ms, cfg3 2.25 -> 2.19 ms; at
            # cfg2's 7040 party rows the extra small launches cost more than the halved GEMMs save: 1.146 vs
            # 1.125 ms per step, tools/ab_project_then_gather.py, round 2)
            w_ih, b_ih, _ = fused_gru._layer_params(self.rnn_parties, 0)
            # the context GRUs' first-layer input contractions read the same projected utterances and do not depend on the
            # party branch: they ride in its grouped projection launch (ops.project_gather riders) instead of one launch each
            riders, ridx = [], {}
            for m, gru in zip(ctx_mods, ctx_grus):
                if m in act and gru.input_size == X[m].shape[-1] and gru.bias:
                    cw, cb, _ = fused_gru._layer_params(gru, 0)
                    ridx[m] = len(riders)
                    riders.append((act.index(m), cw[0], cw[1], cb[0], cb[1], fused_gru._stacked_view(*cw)))
            gi_p, rank, *rest = ops.project_gather([X[m] for m in act], qmask, w_ih[0], w_ih[1], b_ih[0], b_ih[1],
                                                   fused_gru._stacked_view(*w_ih), fused_gru._stacked_view(*b_ih), riders=riders)
            passed, rode = rest[:len(act)], rest[len(act):]
            X.update(zip(act, passed))          # the gathered modalities come back as identities (one consumer each)
            if truncate:
                party = (len(ctx_mods), rank, table)
            outs = fused_gru.bigru2([X[m] for m in ctx_mods] + [None], ctx_grus + [self.rnn_parties], self.dropout,
                                    self.training, gi0=[rode[ridx[m]] if m in ridx else None for m in ctx_mods] + [gi_p],
                                    party=party)
        else:
            # the gathered modalities come back as identities (passthrough): the combine stage below reads those, so
            # each projected modality has one consumer and its two gradient paths meet inside the gather's backward
            S, rank, *passed = ops.party_gather([X[m] for m in act], qmask, passthrough=True)
            X.update(zip(act, passed))
            if truncate:
                party = (len(ctx_mods), rank, table)
            outs = fused_gru.bigru2([X[m] for m in ctx_mods] + [S], ctx_grus + [self.rnn_parties], self.dropout,
                                    self.training, party=party)
        base = dict(X)
        base.update(zip(ctx_mods, outs[:-1]))
        return ops.party_combine([base[m] for m in present], outs[-1], rank, idx, weights, inv)

    # ------------------------------------------------------------------ forward
    def forward(self, U, qmask, umask, seq_lengths, U_a=None, U_v=None, test_label=False):
        # the bf16 piece planes of the GRU input weights follow whatever the optimizer did since the last pass: the stale ones
        # are re-cut by one grouped launch (ops.refresh_planes: a host-side version check when nothing changed)
        if U.is_cuda:
            ops.refresh_planes()
        # one pool of dropout keep flags per forward: every dropout site of the step shares one generator launch
        with ops.flag_pool((U.shape[0], tuple(int(x) for x in seq_lengths))):
            return self._forward(U, qmask, umask, seq_lengths, U_a, U_v, test_label)

    def _forward(self, U, qmask, umask, seq_lengths, U_a, U_v, test_label):
        if ('a' in self.present and U_a is None) or ('v' in self.present and U_v is None):
            raise ValueError("modals=%r needs %s" % (''.join(self.present), "U_a and U_v" if len(self.present) == 3 else
                                                     "U_a" if 'a' in self.present and U_a is None else "U_v"))
        feats = self.encode(U, qmask, seq_lengths, U_a, U_v)
        if self.graph_type == 'DeepGCN':
            # model.py:1242-1290: three independent unimodal graphs, fused after the graph stage;
            # NB dropout THEN ReLU on the fused features, as in the GDF head
            ea = self.graph_net_a(feats[0], seq_lengths, qmask)
            ev = self.graph_net_v(feats[1], seq_lengths, qmask)
            el = self.graph_net_l(feats[2], seq_lengths, qmask)
            fused = (self.gatedatt(ea, ev, el, self.modals) if self.att_type == 'gated'
                     else torch.cat([ea, ev, el], dim=-1))
            return ops.head(fused, self.smax_fc.weight, self.smax_fc.bias, self.dropout_.p, self.training), None, None, None, None
        # without the memory-fusion stage the head reads the (M, N, 300) graph output in place (stacked_out): the
        # cat([a, v, l], -1) of model_mm.py:113-117 and its backward are never materialised
        stacked = self.att_type != 'mfn'
        if self.use_speaker or self.use_modal:
            by = dict(zip(self.present, (feats[i] for i in range(len(self.present)))))
            fused = self.graph_model(by.get('a', []), by.get('v', []), by.get('l', []), seq_lengths, qmask, test_label, stacked)
        else:
            fused = self.graph_model.forward_stacked(feats, seq_lengths, qmask, test_label, stacked)
        if test_label:
            # model.py:1297-1301: the fused (N, 900) graph output of the inference pass, in the reference's column order
            import os
            import numpy as np
            dump = fused.permute(1, 0, 2).reshape(fused.shape[1], -1) if fused.dim() == 3 else fused
            index = 15
            print('# deepGCN layer ' + str(index))
            out_dir = self.graph_model.graph_net.test_output_dir
            os.makedirs(out_dir, exist_ok=True)
            np.save(os.path.join(out_dir, "1080_v2_test_output_multi_{}".format(index)), dump.detach().cpu().numpy())
        if self.att_type == 'mfn':
            # re-pad (N, 900) -> (L, B, 900), memory fusion over time, strip again (model.py:1303-1326)
            L, B = U.shape[0], U.shape[1]
            idx = _flat_index([int(x) for x in seq_lengths], L, B, fused.device)
            padded = fused.new_zeros(L * B, fused.shape[1]).index_copy(0, idx, fused).view(L, B, -1)
            fused = self.mfn(padded).reshape(L * B, -1).index_select(0, idx)
        if test_label:
            # model.py:1331-1335: the class scores behind smax_fc are dumped as well, so the head runs stage by stage here
            import os
            import numpy as np
            flat = fused.permute(1, 0, 2).reshape(fused.shape[1], -1) if fused.dim() == 3 else fused
            scores = ops.linear(torch.relu(F.dropout(flat, self.dropout_.p, self.training)), self.smax_fc.weight,
                                self.smax_fc.bias)
            out_dir = self.graph_model.graph_net.test_output_dir
            np.save(os.path.join(out_dir, "1080_v3_test_output_multi_after_relu-fc_{}".format(15)),
                    scores.detach().cpu().numpy())
            return torch.log_softmax(scores, 1), None, None, None, None
        # dropout -> ReLU -> smax_fc -> log_softmax (model.py:1328-1337) as one fused launch each way
        log_prob = ops.head(fused, self.smax_fc.weight, self.smax_fc.bias, self.dropout_.p, self.training)
        return log_prob, None, None, None, None
