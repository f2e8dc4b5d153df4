"""K6 + K7 forward of a layer in one launch for short dialogues (csrc/gcn_small.hip, mmdfn_prop_layer_fwd) against the two launches
it replaces (mmdfn_propagate + mmdfn_gcnii_layer_fwd, model_GCN.py:178-189) on the same operands: hi, the layer output and the
saved gate mask.  The two forms differ by summation order only (fp32): 2e-6 relative to the largest entry; ReLU decisions may flip
where the pre-activation is within that distance of zero -- such elements are compared through the pre-activation bound."""
import math

import numpy as np
import pytest
import torch

from mm_dfn_amd import _hip, gcn_stack, ops
from mm_dfn_amd.layout import DialogueLayout
from mm_dfn_amd.ops_pad import _lay_args
from util import random_block_adjacency

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [([1], 3, 100), ([31, 32, 33], 3, 100), ([110, 64, 97, 5], 3, 100), ([128, 2], 2, 64), ([50, 20], 1, 100),
         ([110] * 9, 3, 100), ([17, 96], 3, 96), ([40], 3, 4)]


@pytest.mark.parametrize("lengths,M,H", CASES)
@pytest.mark.parametrize("variant", ["plain", "q+mask", "strided"])
def test_fused_layer_forward_matches_the_two_launches(lengths, M, H, variant):
    rs = np.random.RandomState(len(lengths) * 7 + M + H)
    N = sum(lengths)
    R = M * N
    lay = DialogueLayout.get(lengths, M, torch.device(DEV))
    adj = random_block_adjacency(len(lengths) * 11 + M, lengths, M, DEV)[0]
    t = lambda *sh: torch.from_numpy(rs.randn(*sh).astype(np.float32)).to(DEV)
    ldz = 2 * H if variant == "strided" else H
    ldo = H + 8 if variant == "strided" else H
    zbuf = t(R, ldz)
    zin = zbuf[:, ldz - H:] if variant == "strided" else zbuf
    h0, W = t(R, H), t(2 * H, H) * 0.2
    q = t(R, H) if variant != "plain" else None
    mk = (torch.from_numpy((rs.uniform(size=(R, H)) > 0.3).astype(np.float32)).to(DEV)) if variant != "plain" else None
    theta, alpha, ms = math.log(0.5 / 2 + 1), 0.1, 1.0 / 0.7
    P, st, lib = _hip.ptr, _hip.stream(), _hip.lib()
    # reference: the two launches
    hi0 = ops.propagate_raw(adj.tiles, adj.cross, zin, lay)
    out0 = torch.full((R, ldo), float("nan"), device=DEV)
    gm0 = torch.empty(R, H, device=DEV)
    _hip.check(lib.mmdfn_gcnii_layer_fwd(P(hi0), P(h0), P(W), P(q), P(mk), P(out0), P(gm0), theta, alpha, R, H, ldo, ms, st), "k7")
    # the fused launch
    hi1 = torch.full((R, H), float("nan"), device=DEV)
    out1 = torch.full((R, ldo), float("nan"), device=DEV)
    gm1 = torch.full((R, H), float("nan"), device=DEV)
    rc = lib.mmdfn_prop_layer_fwd(P(adj.tiles), P(adj.cross), P(zin), zin.stride(0), *_lay_args(lay), lay.B, lay.M, lay.N, lay.max_len,
                                  P(h0), P(W), P(q), P(mk), P(hi1), P(out1), P(gm1), theta, alpha, H, ldo, ms, st)
    assert rc == 0
    torch.cuda.synchronize()
    assert not torch.isnan(hi1).any() and not torch.isnan(out1[:, :H]).any() and not torch.isnan(gm1).any()
    if ldo > H:
        assert torch.isnan(out1[:, H:]).all()                       # nothing written beyond the H columns
    scale = float(hi0.abs().max())
    assert float((hi1 - hi0).abs().max()) <= 2e-6 * scale
    # elements whose ReLU decision agrees must agree in value; the others sit within rounding of pre = 0 (out differs by <= noise)
    same = (gm1 > 0) == (gm0 > 0)
    oscale = float(out0[:, :H].abs().max())
    assert float(((out1[:, :H] - out0[:, :H]).abs() * same).max()) <= 4e-6 * oscale
    assert float((gm1 - gm0).abs()[same].max() if same.any() else 0.0) == 0.0
    flips = int((~same).sum())
    assert flips <= max(2, R * H // 20000)
    if flips:
        qq = q if q is not None else torch.zeros_like(h0)
        # a flipped element: relu(pre) is either 0 or |pre| ~ rounding noise, so out - q is tiny in both forms
        assert float((out1[:, :H] - qq).abs()[~same].max()) <= 1e-4 * oscale and float((out0[:, :H] - qq).abs()[~same].max()) <= 1e-4 * oscale


def test_stack_node_with_and_without_the_fused_layer_launch(monkeypatch):
    """The fused GCN stack node (forward + backward) with the one-launch layer forward against the two-launch form."""
    from mm_dfn_amd import GCNII_lyc, synthetic
    rs = np.random.RandomState(3)
    lengths, M = [110, 64, 33, 80], 3
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(M, N, 200).astype(np.float32)).to(DEV)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(gcn_stack, "FUSE_PROP_LAYER", fused)
        torch.manual_seed(0)
        net = GCNII_lyc(nfeat=200, nlayers=2, nhidden=100, nclass=6, dropout=0.0, lamda=0.5, alpha=0.1, variant=True,
                        return_feature=True, use_residue=True, reason_flag=True).to(DEV).train()
        x = feats.clone().requires_grad_(True)
        adj = ops.build_adjacency(x, lengths, 1.0)
        out = net(adj.stacked_feats.reshape(M * N, 200), lengths, None, adj)
        (out * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
        res.append((out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters() if p.grad is not None]))
    (o1, g1, p1), (o0, g0, p0) = res
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(o1, o0) < 5e-6 and rel(g1, g0) < 5e-5
    for a, b in zip(p1, p0):
        assert rel(a, b) < 5e-5
