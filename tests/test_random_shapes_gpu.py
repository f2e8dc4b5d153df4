"""GPU: seeded random shapes for the kernels added in round 2, each against a float64 restatement (or the unfused path):
fused GCN stack (rows, widths, layers, masks, residue), head on stacked / plain features, multi-tensor mask-scale, the
GRU recurrence over random sequence counts and lengths (both backward kernels), gather / combine with passthrough."""
import numpy as np
import pytest
import torch

from mm_dfn_amd import GCNII_lyc, gcn_stack, gru as fused, ops
from util import abs_err, random_block_adjacency, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.randn(*shape) * scale).astype(np.float32)).to(DEV)


@pytest.mark.parametrize("seed", range(8))
def test_fused_stack_random_shapes_equal_the_op_by_op_path(seed):
    torch.manual_seed(1000 + seed)               # (the weights below come from torch's generator)
    rs = np.random.RandomState(1000 + seed)
    M = int(rs.choice([1, 2, 3, 6]))
    lengths = [int(x) for x in rs.randint(1, 70, size=rs.randint(1, 5))]
    F, H = 4 * int(rs.randint(1, 60)), 4 * int(rs.randint(1, 26))
    nl = int(rs.randint(1, 5))
    use_res, reason = bool(rs.randint(2)), bool(rs.randint(2))
    adj, _, _, _ = random_block_adjacency(seed, lengths, M, DEV)
    R = M * sum(lengths)
    x = _t(rs, R, F)
    net = GCNII_lyc(nfeat=F, nlayers=nl, nhidden=H, nclass=6, dropout=0.5, lamda=0.5, alpha=0.1, variant=True,
                    return_feature=True, use_residue=use_res, reason_flag=reason).to(DEV)
    for p in net.parameters():
        p.data.normal_(0.0, 0.3)
    net.eval()                                   # no dropout: the two paths must agree to rounding
    assert gcn_stack.eligible(x, F, H, nl, net._stack_params())
    xa = x.clone().requires_grad_(True)
    ya = net._forward_stack(xa, adj)
    w = _t(rs, *ya.shape)
    (ya * w).sum().backward()
    ga = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    for p in net.parameters():
        p.grad = None
    xb = x.clone().requires_grad_(True)
    prev = gcn_stack.ROW_LIMIT
    gcn_stack.ROW_LIMIT = 0                      # forces the op-by-op path
    try:
        yb = net._forward_fused(xb, adj)
    finally:
        gcn_stack.ROW_LIMIT = prev
    (yb * w).sum().backward()
    # (two different fp32 evaluation orders of a stack up to four layers deep with N(0, 0.3) weights)
    assert rel_err(ya, yb) < 5e-5
    assert rel_err(xa.grad, xb.grad) < 2e-4
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert rel_err(ga[k], p.grad) < 5e-4, k


@pytest.mark.parametrize("seed", range(8))
def test_head_random_shapes(seed):
    rs = np.random.RandomState(2000 + seed)
    M, N, Wm, C = int(rs.randint(1, 7)), int(rs.randint(1, 3000)), 4 * int(rs.randint(1, 80)), int(rs.randint(1, 9))
    p = float(rs.choice([0.0, 0.3, 0.5]))
    F3 = _t(rs, M, N, Wm).requires_grad_(True)
    W, b = _t(rs, C, M * Wm, scale=0.05).requires_grad_(True), _t(rs, C).requires_grad_(True)
    if not ops.head_supported(F3, W):
        pytest.skip("wider than the head kernel's LDS budget")
    mask = torch.from_numpy((rs.uniform(size=(N, M * Wm)) > p).astype(np.float32)).to(DEV) if p > 0 else None
    ms = 1.0 / (1.0 - p)
    G = _t(rs, N, C)
    logp = ops._Head.apply(F3, mask, ms, W, b)
    (logp * G).sum().backward()
    Fd = F3.detach().permute(1, 0, 2).reshape(N, M * Wm).double().cpu().requires_grad_(True)
    Wd, bd = W.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    z = torch.relu(Fd * (mask.double().cpu() * ms if mask is not None else 1.0))
    want = torch.log_softmax(z @ Wd.t() + bd, 1)
    (want * G.double().cpu()).sum().backward()
    assert rel_err(logp, want) < 5e-6
    assert rel_err(F3.grad, Fd.grad.view(N, M, Wm).permute(1, 0, 2)) < 2e-5
    assert rel_err(W.grad, Wd.grad) < 2e-5 and rel_err(b.grad, bd.grad) < 2e-5


@pytest.mark.parametrize("seed", range(6))
def test_gru_random_sequence_counts_and_lengths(seed, kernel_variants):
    """Forward against torch's nn.GRU (eval), backward of both one-sequence-per-workgroup kernels against each other and
    of the rows-per-workgroup kernels against autograd through torch's GRU."""
    torch.manual_seed(3000 + seed)
    rs = np.random.RandomState(3000 + seed)
    ngroups = int(rs.randint(1, 4))
    shapes = [(int(rs.randint(1, 60)), int(rs.choice([1, 2, 7, 33, 90, 200]))) for _ in range(ngroups)]
    grus = []
    for i in range(ngroups):
        g = torch.nn.GRU(200, 100, num_layers=2, bidirectional=True)
        for p in g.parameters():
            p.data.uniform_(-0.15, 0.15)
        grus.append(g)
    xs = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)) for T, R in shapes]
    ws = [torch.from_numpy(rs.randn(T, R, 200).astype(np.float32)) for T, R in shapes]
    want, wx, wp = [], [], []
    for g, x, w in zip(grus, xs, ws):
        xr = x.clone().requires_grad_(True)
        y = g(xr)[0]
        (y * w).sum().backward()
        want.append(y.detach()); wx.append(xr.grad); wp.append({k: p.grad.clone() for k, p in g.named_parameters()})
        for p in g.parameters():
            p.grad = None
    for mode in ("1", "0"):
        kernel_variants.setenv("MMDFN_GRU_KPART_BWD", mode)
        gd = [torch.nn.GRU(200, 100, num_layers=2, bidirectional=True).to(DEV) for _ in grus]
        for a, b in zip(gd, grus):
            a.load_state_dict(b.state_dict())
        xg = [x.to(DEV).requires_grad_(True) for x in xs]
        ys = fused.bigru2(xg, gd, 0.0, True)
        sum((y * w.to(DEV)).sum() for y, w in zip(ys, ws)).backward()
        for i in range(ngroups):
            assert abs_err(ys[i], want[i]) < 5e-6
            assert rel_err(xg[i].grad, wx[i]) < 5e-5
            for k, p in gd[i].named_parameters():
                assert rel_err(p.grad, wp[i][k]) < 2e-4, (mode, k)


@pytest.mark.parametrize("seed", range(4))
def test_mask_scale_random_groups(seed):
    rs = np.random.RandomState(4000 + seed)
    n = int(rs.randint(1, 7))
    shapes = [tuple(int(v) for v in rs.randint(1, 40, size=rs.randint(1, 4))) + (4,) for _ in range(n)]
    xs = [_t(rs, *sh).requires_grad_(True) for sh in shapes]
    masks = [torch.from_numpy((rs.uniform(size=int(np.prod(sh))) > 0.4).astype(np.float32)).to(DEV) for sh in shapes]
    ws = [_t(rs, *sh) for sh in shapes]
    outs = ops.mask_scale(xs, masks, 1.0 / 0.6)
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    for x, m, w, o in zip(xs, masks, ws, outs):
        assert torch.equal(o, x.detach() * m.view_as(x) * (1.0 / 0.6))
        assert torch.equal(x.grad, w * m.view_as(x) * (1.0 / 0.6))
