"""The GCNII "dynamic fusion" stack (reference GCNII_lyc.forward, model_GCN.py:444-488) as ONE autograd node.

Forward = 1 + 3 launches per layer (input stage; per layer: LSTM gate K8, propagate K6, GCNII update K7 -- K6 + K7 as ONE
launch for short dialogues, csrc/gcn_small.hip), backward =
1 + 3 per layer (K7', K6, K8') + ONE adjacency-gradient contraction for the whole stack (round 5: the layers' propagated
states and their gradients are column blocks of two (R, nl H) buffers) -- every stage is a fused kernel of
csrc/gcn_stack.hip / propagate.hip / tile_dot.hip.  Written as a single ``torch.autograd.Function`` with a
hand-scheduled backward because the stack's dataflow has three fan-outs that autograd would serve with accumulation
kernels: h0 feeds every layer (the kernels accumulate dh0 in place), the adjacency feeds every layer (tile_outer
accumulates dA in place) and the LSTM cell is shared by all layers (its weight gradients are extra segments of one
reduction in the end-of-backward batch, ops.queue_wgrad).  Parameters must be leaf tensors (their gradients are
written to ``.grad`` by that batch); GCNII_lyc falls back to the op-by-op path otherwise, and for launches of more
than ROW_LIMIT rows (beyond anything measured: the fused node is ahead of the op-by-op path with its bf16-piece GEMMs from
5 280 to 98 304 rows, tools/time_stack_paths.py).
"""
import math

PRECUT = __import__("os").environ.get("MMDFN_GATE_PRECUT", "1") == "1"     # (0: A/B aid, every workgroup cuts the cell's weights itself)
FUSE_PROP_LAYER = __import__("os").environ.get("MMDFN_FUSE_PROP_LAYER", "1") == "1"   # (0: A/B aid, propagate and the layer update as two launches)

import torch

from . import _hip, ops
from .ops_pad import _lay_args

ROW_LIMIT = 131072        # measured to 98 304 rows (cfg5 B = 32): the fused node is 8-9 % ahead of the op-by-op path at every size (round 4)
# test tap: a list to which every forward of the fused node appends references to its ReLU decisions (h0, the per-layer
# gate masks, the output) -- tests/util.relu_flips_from_tap compares them with the oracle's pre-activations to find the
# units whose pre-activation is within rounding of zero and landed on the other side.  None (the default): nothing is kept.
TAP = None


def eligible(x, nfeat, nhidden, nlayers, params, lamda=1.0):
    # lamda > 0: theta_l = ln(lamda / l + 1) > 0, which the backward kernel divides by (c1 = (1-theta)(1-alpha)/theta)
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[0] <= ROW_LIMIT and nlayers >= 1 and lamda > 0
            and nhidden % 4 == 0 and 4 <= nhidden <= 100 and nfeat % 4 == 0 and 4 <= nfeat <= 256
            and (not torch.is_grad_enabled() or all(ops._leaf(p) for p in params)))


class _GcnStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tiles, cross, masks, mscale, lay, symmetric, lamda, alpha, reason, use_residue, W0, b0, w_ih, w_hh,
                b_ih, b_hh, *convW):
        lib = _hip.lib()
        P, st = _hip.ptr, _hip.stream()
        x = x.contiguous()
        R, F = x.shape
        H = W0.shape[0]
        nl = len(convW)
        dev = x.device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        mx = m0 = None
        ml = [None] * nl
        if masks is not None:
            mx = masks[:R * F].view(R, F)
            m0 = masks[R * F:R * (F + H)].view(R, H)
            ml = [masks[R * (F + H) + i * R * H: R * (F + H) + (i + 1) * R * H].view(R, H) for i in range(nl)]
        out = new(R, F + H) if use_residue else new(R, H)
        xd = out if use_residue else new(R, F)                 # x_d lives in the left F columns of the residue output
        ldxd = F + H if use_residue else F
        h0, cur = new(R, H), new(R, H)
        _hip.check(lib.mmdfn_gcn_input_fwd(P(x), P(mx), P(W0), P(b0), P(m0), P(xd), P(h0), P(cur), R, F, H, ldxd, mscale, st),
                   "mmdfn_gcn_input_fwd")
        h = c = None
        layers = []
        # the layers' hidden states are column blocks of ONE (R, nl H) buffer: what every layer propagates (zin_l) is then the
        # Y operand of a single adjacency-gradient contraction of width nl H in the backward pass (the adjacency is shared by
        # the layers, model_GCN.py:461-472), instead of nl read-modify-write passes over the tile array
        zin_all = new(R, nl * H) if reason else None
        planes = None
        if PRECUT and reason and nl > 1 and lib.mmdfn_lstm_gate_takes_planes(R, H):
            # many-row launches (bf16-piece form): the cell's weights are cut ONCE for all layers of this forward pass
            planes = new(int(lib.mmdfn_lstm_gate_planes_workspace(H)))
            _hip.check(lib.mmdfn_lstm_gate_cut_weights(P(w_ih), P(w_hh), P(planes), H, st), "mmdfn_lstm_gate_cut_weights")
        for i in range(nl):
            q = cur
            rec = dict(q=q, h_prev=h, c_prev=c)
            if reason:
                gates, h_new, c_new = new(R, 4 * H), zin_all[:, i * H:(i + 1) * H], new(R, H)
                _hip.check(lib.mmdfn_lstm_gate_fwd_pre(P(q), P(h), P(c), P(w_ih), P(w_hh), P(b_ih), P(b_hh), P(gates), P(h_new),
                                                       P(c_new), R, H, nl * H, P(planes), st), "mmdfn_lstm_gate_fwd_pre")
                rec.update(gates=gates, c_new=c_new)
                h, c = h_new, c_new
                zin = h_new
            else:
                zin = q
            last = i == nl - 1
            if last and use_residue:
                dst, ldo = out[:, F:], F + H
            elif last:
                dst, ldo = out, H
            else:
                dst, ldo = new(R, H), H
            gmask = new(R, H)
            theta = math.log(lamda / (i + 1) + 1)
            # short dialogues: propagate + layer update of a strip of rows in ONE launch (csrc/gcn_small.hip; hi is still written
            # out for the backward pass); -2 = shape not covered: the two launches
            hi = new(R, H)
            rc = lib.mmdfn_prop_layer_fwd(P(tiles), P(cross), P(zin), zin.stride(0), *_lay_args(lay), lay.B, lay.M, lay.N,
                                          lay.max_len, P(h0), P(convW[i]), P(q if reason else None), P(ml[i]), P(hi), P(dst),
                                          P(gmask), theta, alpha, H, ldo, mscale, st) if FUSE_PROP_LAYER else -2
            if rc == -2:
                hi = ops.propagate_raw(tiles, cross, zin, lay, out=hi)
                _hip.check(lib.mmdfn_gcnii_layer_fwd(P(hi), P(h0), P(convW[i]), P(q if reason else None), P(ml[i]), P(dst),
                                                     P(gmask), theta, alpha, R, H, ldo, mscale, st), "mmdfn_gcnii_layer_fwd")
            else:
                _hip.check(rc, "mmdfn_prop_layer_fwd")
            rec.update(zin=zin, hi=hi, gmask=gmask, theta=theta)
            layers.append(rec)
            cur = dst
        if TAP is not None:
            TAP.append(dict(h0=h0, gmask=[rec["gmask"] for rec in layers], out=out.detach()))
        ctx.layers = layers
        # (xd is a detached alias: holding the output itself would tie this node and its output into a reference cycle)
        ctx.misc = dict(lay=lay, symmetric=symmetric, alpha=alpha, reason=reason, use_residue=use_residue, R=R, F=F, H=H,
                        mx=mx, m0=m0, xd=xd.detach(), h0=h0, tiles=tiles, cross=cross, mscale=mscale, zin_all=zin_all)
        ctx.params = (W0, b0, w_ih, w_hh, b_ih, b_hh, convW)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _hip.lib()
        P, st = _hip.ptr, _hip.stream()
        m = ctx.misc
        R, F, H = m["R"], m["F"], m["H"]
        W0, b0, w_ih, w_hh, b_ih, b_hh, convW = ctx.params
        lay, tiles, cross = m["lay"], m["tiles"], m["cross"]
        dev = dout.device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        dout = dout.contiguous()
        if m["use_residue"]:
            dcur, lddo, dxd, lddxd = dout[:, F:], F + H, dout, F + H
        else:
            dcur, lddo, dxd, lddxd = dout, H, None, 0
        want_adj = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        # weight gradients: queued for the step's one-launch batch (ops.wgrad_batch scope, hook-free leaf parameters), or
        # computed here and returned to autograd (plain loss.backward(), torch.autograd.grad, hooked parameters)
        plist = [p for p in (W0, b0, w_ih, w_hh, b_ih, b_hh) if p is not None] + list(convW)
        batched = ops.wgrad_batching() and not any(ops._hooked(p) for p in plist)
        inline = {}                                    # id(param) -> gradient (in-line mode)

        def wgrad(A, B, weight, biases=(), rows=None):
            if batched:
                ops.queue_wgrad(A, B, weight, biases, rows=rows)
                return
            dw, db = ops._wgrad_inline(A, B, bool(biases))
            if rows is not None:
                g = inline.get(id(weight))
                if g is None:
                    g = inline[id(weight)] = torch.zeros_like(weight)
                g[rows[0]:rows[1]] += dw
            else:
                inline[id(weight)] = dw if id(weight) not in inline else inline[id(weight)] + dw
            for b in biases:
                inline[id(b)] = db if id(b) not in inline else inline[id(b)] + db

        dh0 = new(R, H)
        acc_h0 = 0
        dtiles = dcross = None
        dh_carry = dc_carry = None
        nl = len(ctx.layers)
        # one adjacency-gradient launch for the whole stack: dhi_l are column blocks of one buffer, like zin_l (see forward)
        one_outer = want_adj and m["zin_all"] is not None
        dhi_all = new(R, nl * H) if one_outer else None
        for i in reversed(range(nl)):
            L = ctx.layers[i]
            dP = new(R, H)
            dhi = dhi_all[:, i * H:(i + 1) * H] if one_outer else new(R, H)
            _hip.check(lib.mmdfn_gcnii_layer_bwd_ld(P(dcur), P(L["gmask"]), P(convW[i]), P(dP), P(dhi), P(dh0), L["theta"],
                                                    m["alpha"], R, H, lddo, acc_h0, dhi.stride(0), st), "mmdfn_gcnii_layer_bwd_ld")
            acc_h0 = 1
            # dW_i = [hi | h0]^T dP: two row ranges of the (2H, H) parameter, the concatenated operand never exists
            wgrad(L["hi"], dP, convW[i], rows=(0, H))
            wgrad(m["h0"], dP, convW[i], rows=(H, 2 * H))
            dz = ops.propagate_raw(tiles, cross, dhi, lay, transpose=not m["symmetric"])
            if want_adj and not one_outer:
                dtiles, dcross = ops.tile_outer_raw(dhi, L["zin"], lay, dtiles, dcross)
            if m["reason"]:
                has_h = L["h_prev"] is not None
                dG, dq = new(R, 4 * H), new(R, H)
                dh_prev = new(R, H) if has_h else None
                dc_prev = new(R, H) if has_h else None
                _hip.check(lib.mmdfn_lstm_gate_bwd(P(L["gates"]), P(L["c_prev"]), P(L["c_new"]), P(dz), P(dh_carry),
                                                   P(dc_carry), P(w_ih), P(w_hh), P(dcur), P(dG), P(dc_prev), P(dq),
                                                   P(dh_prev), R, H, 1 if has_h else 0, lddo, st), "mmdfn_lstm_gate_bwd")
                wgrad(dG, L["q"], w_ih, [b_ih, b_hh])
                if has_h:
                    wgrad(dG, L["h_prev"], w_hh)
                dcur, lddo = dq, H
                dh_carry, dc_carry = dh_prev, dc_prev
            else:
                dcur, lddo = dz, H
        if one_outer:
            # dA = [dhi_1 | .. | dhi_nl] [zin_1 | .. | zin_nl]^T on the tile / diagonal pattern: the tile array is written once
            dtiles, dcross = ops.tile_outer_raw(dhi_all, m["zin_all"], lay)
        dpre, dx = new(R, H), new(R, F)
        _hip.check(lib.mmdfn_gcn_input_bwd(P(dcur), P(m["m0"]), P(dh0), P(m["h0"]), P(W0), P(dxd), P(m["mx"]), P(dpre), P(dx), R,
                                           F, H, lddxd, m["mscale"], st), "mmdfn_gcn_input_bwd")
        xd = m["xd"][:, :F] if m["use_residue"] else m["xd"]
        wgrad(dpre, xd, W0, [b0] if b0 is not None else [])
        pg = [None if p is None else inline.get(id(p)) for p in (W0, b0, w_ih, w_hh, b_ih, b_hh)]
        return (dx, dtiles, dcross) + (None,) * 8 + tuple(pg) + tuple(inline.get(id(w)) for w in convW)


def gcn_stack(x, adj, masks, mscale, lamda, alpha, reason_flag, use_residue, W0, b0, lstm, convs):
    """x (R, F) -> [x (.) m_x | cur] (R, F + H) (or cur alone without use_residue).  ``masks``: one flat fp32 tensor of
    0 / 1 keep flags for x, h0 and every layer (R F + R H + nl R H floats; the kernels multiply by ``mscale`` = 1/(1-p)),
    or None (eval / p = 0)."""
    w_ih = w_hh = b_ih = b_hh = None
    if reason_flag:
        w_ih, w_hh, b_ih, b_hh = lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0
    return _GcnStack.apply(x, adj.tiles, adj.cross, masks, float(mscale), adj.layout, adj.symmetric, float(lamda), float(alpha),
                           bool(reason_flag), bool(use_residue), W0, b0, w_ih, w_hh, b_ih, b_hh, *convs)
