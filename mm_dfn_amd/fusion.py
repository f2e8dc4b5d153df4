"""Secondary fusion modules named by the north star, reference signatures kept:

* ``MFN`` (model_fusion.py:10-120): memory fusion network applied after the GDF graph when
  ``--mm_fusion_mthd mfn`` (model.py:1303-1326).  Non-default.  Composed from torch-ROCm ops (the three
  LSTM input projections are hoisted out of the time loop and the three recurrent projections are one batched
  product); it is not one of the hand-written kernels.
* ``MMGatedAttention`` ('general', model.py:718-781): unreachable under graph_type='GDF' in the reference
  (shape bug); provided at module level only, with the reference's state_dict keys.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MFN(nn.Module):
    def __init__(self, d=300, config=None):
        super().__init__()
        self.d_l, self.d_a, self.d_v = d, d, d
        self.dh_l, self.dh_a, self.dh_v = 100, 100, 100
        total_h = self.dh_l + self.dh_a + self.dh_v
        self.mem_dim = 100
        att_in = total_h * 2
        gamma_in = att_in + self.mem_dim
        final_out = total_h + self.mem_dim
        self.lstm_l = nn.LSTMCell(self.d_l, self.dh_l)
        self.lstm_a = nn.LSTMCell(self.d_a, self.dh_a)
        self.lstm_v = nn.LSTMCell(self.d_v, self.dh_v)
        self.att1_fc1 = nn.Linear(att_in, 100)
        self.att1_fc2 = nn.Linear(100, att_in)
        self.att1_dropout = nn.Dropout(0.2)
        self.att2_fc1 = nn.Linear(att_in, 100)
        self.att2_fc2 = nn.Linear(100, self.mem_dim)
        self.att2_dropout = nn.Dropout(0.2)
        self.gamma1_fc1 = nn.Linear(gamma_in, 100)
        self.gamma1_fc2 = nn.Linear(100, self.mem_dim)
        self.gamma1_dropout = nn.Dropout(0.2)
        self.gamma2_fc1 = nn.Linear(gamma_in, 100)
        self.gamma2_fc2 = nn.Linear(100, self.mem_dim)
        self.gamma2_dropout = nn.Dropout(0.2)
        self.out_fc1 = nn.Linear(final_out, 100)      # constructed by the reference, unused in forward
        self.out_fc2 = nn.Linear(100, 1)
        self.out_dropout = nn.Dropout(0.2)

    def forward(self, x):
        """x: (T, n, 3d) -> (T, n, 400) = [h_l | h_a | h_v | mem]."""
        T, n = x.shape[0], x.shape[1]
        cells = (self.lstm_l, self.lstm_a, self.lstm_v)
        parts = (x[:, :, :self.d_l], x[:, :, self.d_l:self.d_l + self.d_a], x[:, :, self.d_l + self.d_a:])
        # hoisted input projections for all timesteps: (3, T, n, 400)
        gx = torch.stack([F.linear(p, c.weight_ih, c.bias_ih + c.bias_hh) for p, c in zip(parts, cells)], 0)
        w_hh = torch.stack([c.weight_hh.t() for c in cells], 0)            # (3, 100, 400)
        h = x.new_zeros(3, n, 100)
        c = x.new_zeros(3, n, 100)
        mem = x.new_zeros(n, self.mem_dim)
        hs, mems = [], []
        for t in range(T):
            g = gx[:, t] + torch.bmm(h, w_hh)                             # (3, n, 400) gate order i, f, g, o
            i, f, gg, o = g.chunk(4, -1)
            c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c_new)
            prev_cs = torch.cat([c[0], c[1], c[2]], 1)
            new_cs = torch.cat([c_new[0], c_new[1], c_new[2]], 1)
            c_star = torch.cat([prev_cs, new_cs], 1)
            att = F.softmax(self.att1_fc2(self.att1_dropout(F.relu(self.att1_fc1(c_star)))), dim=1)
            attended = att * c_star
            c_hat = torch.tanh(self.att2_fc2(self.att2_dropout(F.relu(self.att2_fc1(attended)))))
            both = torch.cat([attended, mem], 1)
            g1 = torch.sigmoid(self.gamma1_fc2(self.gamma1_dropout(F.relu(self.gamma1_fc1(both)))))
            g2 = torch.sigmoid(self.gamma2_fc2(self.gamma2_dropout(F.relu(self.gamma2_fc1(both)))))
            mem = g1 * mem + g2 * c_hat
            c = c_new
            hs.append(torch.cat([h[0], h[1], h[2]], 1))
            mems.append(mem)
        return torch.cat([torch.stack(hs), torch.stack(mems)], -1)


class MMGatedAttention(nn.Module):
    def __init__(self, mem_dim, cand_dim, att_type='general'):
        super().__init__()
        if att_type != 'general':
            raise NotImplementedError("only att_type='general' is constructed by the reference model (model.py:982)")
        self.mem_dim, self.cand_dim, self.att_type = mem_dim, cand_dim, att_type
        self.dropouta = nn.Dropout(0.5)
        self.dropoutv = nn.Dropout(0.5)
        self.dropoutl = nn.Dropout(0.5)
        self.transform_l = nn.Linear(mem_dim, cand_dim)
        self.transform_v = nn.Linear(mem_dim, cand_dim)
        self.transform_a = nn.Linear(mem_dim, cand_dim)
        self.transform_av = nn.Linear(mem_dim * 3, 1)
        self.transform_al = nn.Linear(mem_dim * 3, 1)
        self.transform_vl = nn.Linear(mem_dim * 3, 1)

    def forward(self, a, v, l, modals=None):
        modals = modals if modals is not None else ['a', 'v', 'l']
        a = self.dropouta(a) if len(a) != 0 else a
        v = self.dropoutv(v) if len(v) != 0 else v
        l = self.dropoutl(l) if len(l) != 0 else l
        ha = torch.tanh(self.transform_a(a)) if 'a' in modals else a
        hv = torch.tanh(self.transform_v(v)) if 'v' in modals else v
        hl = torch.tanh(self.transform_l(l)) if 'l' in modals else l
        out = []
        if 'a' in modals and 'v' in modals:
            z = torch.sigmoid(self.transform_av(torch.cat([a, v, a * v], -1)))
            h_av = z * ha + (1 - z) * hv
            if 'l' not in modals:
                return h_av
            out.append(h_av)
        if 'a' in modals and 'l' in modals:
            z = torch.sigmoid(self.transform_al(torch.cat([a, l, a * l], -1)))
            h_al = z * ha + (1 - z) * hl
            if 'v' not in modals:
                return h_al
            out.append(h_al)
        if 'v' in modals and 'l' in modals:
            z = torch.sigmoid(self.transform_vl(torch.cat([v, l, v * l], -1)))
            h_vl = z * hv + (1 - z) * hl
            if 'a' not in modals:
                return h_vl
            out.append(h_vl)
        return torch.cat(out, -1)
