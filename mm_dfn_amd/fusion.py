"""Secondary fusion modules named by the north star, reference signatures kept:

* ``MFN`` (model_fusion.py:10-120): memory fusion network applied after the GDF graph when
  ``--mm_fusion_mthd mfn`` (model.py:1303-1326).  Non-default.  Every dense product of a timestep (three recurrent
  LSTM projections; the attention and gamma MLPs) is a grouped launch of the few-row MFMA kernel
  (csrc/linear_small.hip, forward and input gradients; weight gradients through the step's gemm_tn batch), the LSTM
  cells run on the fused gate kernel of the GCN stack (csrc/gcn_pointwise.hip) and the attention / memory update on
  csrc/fusion.hip: per step 4 grouped GEMM launches + 3 pointwise launches, no library GEMM.  The three input
  projections are hoisted out of the time loop (one grouped launch for all timesteps).
* ``MMGatedAttention`` ('general', model.py:718-781): unreachable under graph_type='GDF' in the reference
  (shape bug); provided at module level, with the reference's state_dict keys: one grouped launch
  for the three transforms, one fused kernel per modality pair (gate dot product, sigmoid, tanh, blend) each way.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


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
        G = ops.linear_group
        # hoisted input projections of the three cells for all timesteps: one grouped launch, (T, n, 400) each
        gx = G([p.reshape(T * n, p.shape[-1]) for p in parts], [c.weight_ih for c in cells],
               [c.bias_ih + c.bias_hh for c in cells], hip=True)
        gx = [g.view(T, n, -1) for g in gx]
        w_hh = [c.weight_hh for c in cells]
        h = [x.new_zeros(n, 100) for _ in range(3)]
        c = None                                                            # (3 n, 100); None = zero state
        prev_cs = x.new_zeros(n, 300)
        mem = x.new_zeros(n, self.mem_dim)
        hs, mems = [], []
        for t in range(T):
            rec = G(h, w_hh, [None] * 3, hip=True)                           # h_m W_hh_m^T, gate order i, f, g, o
            gates = torch.cat([gx[m][t] + rec[m] for m in range(3)], 0)     # (3 n, 400)
            h_new, c_new = ops.lstm_pointwise(gates, c)
            new_cs = c_new.view(3, n, 100).permute(1, 0, 2).reshape(n, 300)
            c_star = torch.cat([prev_cs, new_cs], 1)
            a1 = self.att1_dropout(G([c_star], [self.att1_fc1.weight], [self.att1_fc1.bias], act=1, hip=True)[0])
            z = G([a1], [self.att1_fc2.weight], [self.att1_fc2.bias], hip=True)[0]
            attended = ops.softmax_scale(z, c_star)
            both = torch.cat([attended, mem], 1)
            y = G([attended, both, both], [self.att2_fc1.weight, self.gamma1_fc1.weight, self.gamma2_fc1.weight],
                  [self.att2_fc1.bias, self.gamma1_fc1.bias, self.gamma2_fc1.bias], act=1, hip=True)
            y = [self.att2_dropout(y[0]), self.gamma1_dropout(y[1]), self.gamma2_dropout(y[2])]
            u, v1, v2 = G(y, [self.att2_fc2.weight, self.gamma1_fc2.weight, self.gamma2_fc2.weight],
                          [self.att2_fc2.bias, self.gamma1_fc2.bias, self.gamma2_fc2.bias], hip=True)
            mem = ops.mfn_mem(u, v1, v2, mem)
            c, prev_cs = c_new, new_cs
            hv = h_new.view(3, n, 100)
            h = [hv[0], hv[1], hv[2]]
            hs.append(hv.permute(1, 0, 2).reshape(n, 300))
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
        xs = {'a': a, 'v': v, 'l': l}
        tr = {'a': self.transform_a, 'v': self.transform_v, 'l': self.transform_l}
        use = [m for m in ('a', 'v', 'l') if m in modals]
        # the transforms' pre-activations in one grouped launch; tanh is applied inside the pair kernel
        pre = dict(zip(use, ops.linear_group([xs[m] for m in use], [tr[m].weight for m in use], [tr[m].bias for m in use],
                                             hip=True)))
        gate = {('a', 'v'): self.transform_av, ('a', 'l'): self.transform_al, ('v', 'l'): self.transform_vl}
        out = [ops.gated_pair(xs[m], xs[n], pre[m], pre[n], g.weight, g.bias)
               for (m, n), g in gate.items() if m in modals and n in modals]
        if len(out) == 1:
            return out[0]
        return torch.cat(out, -1)
