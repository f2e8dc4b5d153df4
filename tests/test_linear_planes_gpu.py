"""GPU: the projection kernel that takes its weight as bf16 piece planes (csrc/linear_planes.hip) and the plane registry that
keeps those planes in step with the parameters (ops.weight_planes / refresh_planes / invalidate_planes)."""
import pytest
import torch

from mm_dfn_amd import ops

pytestmark = pytest.mark.gpu


def _ref(x, w1, w2, b1, b2, transposed, act, base):
    W = torch.cat([w1, w2]) if w2 is not None else w1
    W = W.double()
    y = x.double() @ (W if transposed else W.t())
    if not transposed:
        b = [t for t in (b1, b2) if t is not None]
        if b:
            y = y + torch.cat(b).double()
    if act:
        y = y.clamp_min(0)
    if base is not None:
        y = y + base.double()
    return y


@pytest.mark.parametrize("R,K,n1,n2", [(7040, 200, 300, 300), (1760, 200, 300, 300), (130, 200, 300, 300), (64, 8, 40, 0),
                                       (1000, 212, 33, 0), (517, 600, 100, 100), (3, 100, 17, 19), (19008, 200, 300, 300)])
def test_linear_planes_against_fp64(R, K, n1, n2):
    g = torch.Generator(device="cuda").manual_seed(R + K)
    x = torch.randn(R, K, device="cuda", generator=g)
    w1 = torch.randn(n1, K, device="cuda", generator=g) * 0.1
    w2 = torch.randn(n2, K, device="cuda", generator=g) * 0.1 if n2 else None
    b1 = torch.randn(n1, device="cuda", generator=g)
    b2 = torch.randn(n2, device="cuda", generator=g) if n2 else None
    y = ops.linear_planes_raw(x, w1, w2, b1, b2)
    want = _ref(x, w1, w2, b1, b2, False, 0, None)
    scale = float(want.abs().max())
    # fp32-level error: six exact bf16 piece products per MAC, fp32 accumulation (<= 4x the exact-f32 MFMA kernel's error)
    assert float((y.double() - want).abs().max()) <= 2e-6 * scale
    exact = ops.linear_group_raw([dict(x=x, w=w1, w2=w2, b=b1, b2=b2)])[0] if (K <= 768 and R <= 8192) else None
    if exact is not None:
        e_exact = float((exact.double() - want).abs().max())
        assert float((y.double() - want).abs().max()) <= max(4.0 * e_exact, 1e-6 * scale)
    # ReLU + accumulate
    base = torch.randn(R, n1 + n2, device="cuda", generator=g)
    out = base.clone()
    ops.linear_planes_raw(x, w1, w2, b1, b2, act=1, out=out, accumulate=True)
    want = _ref(x, w1, w2, b1, b2, False, 1, base)
    assert float((out.double() - want).abs().max()) <= 2e-6 * max(scale, float(want.abs().max()))


@pytest.mark.parametrize("R,K,n1,n2", [(7040, 200, 300, 300), (1760, 200, 300, 300), (333, 100, 52, 0), (200, 36, 24, 40)])
def test_linear_planes_transposed_operand_is_the_input_gradient(R, K, n1, n2):
    """dX = dY . [w1; w2]: the contraction runs over the stored ROWS (both blocks), the output has the stored columns."""
    g = torch.Generator(device="cuda").manual_seed(7 * R + K)
    dy = torch.randn(R, n1 + n2, device="cuda", generator=g)
    w1 = torch.randn(n1, K, device="cuda", generator=g) * 0.1
    w2 = torch.randn(n2, K, device="cuda", generator=g) * 0.1 if n2 else None
    dx = ops.linear_planes_raw(dy, w1, w2, transposed=True)
    want = _ref(dy, w1, w2, None, None, True, 0, None)
    assert dx.shape == (R, K)
    assert float((dx.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_row_strided_input_and_nan_beyond_k():
    """X rows wider than K (a column block of a wider matrix): what lies beyond K -- NaN here -- must not reach the result."""
    g = torch.Generator(device="cuda").manual_seed(3)
    big = torch.full((500, 120), float("nan"), device="cuda")
    big[:, :100] = torch.randn(500, 100, device="cuda", generator=g)
    x = big[:, :100]
    w = torch.randn(40, 100, device="cuda", generator=g)
    y = ops.linear_planes_raw(x, w)
    want = x.double() @ w.double().t()
    assert torch.isfinite(y).all() and float((y.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_planes_follow_the_parameters():
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(256, 64, device="cuda", generator=g)
    w1 = torch.nn.Parameter(torch.randn(48, 64, device="cuda", generator=g))
    w2 = torch.nn.Parameter(torch.randn(48, 64, device="cuda", generator=g))

    def check():
        y = ops.linear_planes_raw(x, w1, w2)
        want = x.double() @ torch.cat([w1.detach(), w2.detach()]).double().t()
        return float((y.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())

    assert check()
    e = ops.weight_planes(w1, w2)
    before = e.buf.clone()
    assert check() and torch.equal(e.buf, before)          # nothing changed: no re-cut needed, same planes
    with torch.no_grad():
        w2.mul_(-2.0)                                      # an in-place update autograd's version counter sees
    assert check()
    w1.data.view(-1)[:5].zero_()                           # ... and one it does not see (a raw kernel's write, as FlatAdam's)
    assert not check()
    ops.invalidate_planes()
    assert check()
    w1.data.view(-1)[5:9].fill_(3.0)
    ops.invalidate_planes()
    buf0 = e.buf.clone()
    ops.refresh_planes()                                   # what a model's forward pass does first: stale entries, one launch
    assert not torch.equal(e.buf, buf0) and check()
    buf0 = e.buf.clone()
    ops.refresh_planes()                                   # nothing stale: nothing launched
    assert torch.equal(e.buf, buf0)
    # a captured use: the graph holds no cut; what the capture recorded is re-checked (and re-cut) in front of a replay
    out = torch.empty(256, 96, device="cuda")
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with ops.planes_recording() as used:
        with torch.cuda.graph(gr):
            ops.linear_planes_raw(x, w1, w2, out=out)
    assert len(used) == 1
    with torch.no_grad():
        w1.mul_(0.5)
    ops.refresh_planes(used)
    gr.replay()
    torch.cuda.synchronize()
    want = x.double() @ torch.cat([w1.detach(), w2.detach()]).double().t()
    assert float((out.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    del used
    # entries die with their parameters
    n = len(ops._PLANES)
    del w1, w2, e, check, gr
    import gc
    gc.collect()
    ops.refresh_planes()
    assert len(ops._PLANES) < n


def test_linear2_node_on_planes_matches_autograd():
    """ops.linear2 (the GRU input contraction node) above PLANES_MIN_ROWS: output and input gradient from the plane kernels, weight
    gradients as before."""
    g = torch.Generator(device="cuda").manual_seed(5)
    R = max(ops.PLANES_MIN_ROWS, 1024) + 37
    x = torch.randn(R, 200, device="cuda", generator=g, requires_grad=True)
    prm = [torch.nn.Parameter(torch.randn(*s, device="cuda", generator=g) * 0.1) for s in ((300, 200), (300, 200), (300,), (300,))]
    y = ops.linear2(x, *prm)
    dy = torch.randn_like(y)
    y.backward(dy)
    xr = x.detach().double().requires_grad_()
    pr = [p.detach().double().requires_grad_() for p in prm]
    yr = xr @ torch.cat(pr[:2]).t() + torch.cat(pr[2:])
    yr.backward(dy.double())
    assert float((y.double() - yr).abs().max()) <= 2e-6 * float(yr.abs().max())
    assert float((x.grad.double() - xr.grad).abs().max()) <= 2e-6 * float(xr.grad.abs().max())
    for p, q in zip(prm, pr):
        assert float((p.grad.double() - q.grad).abs().max()) <= 3e-6 * float(q.grad.abs().max())


def test_captured_step_with_flat_adam_keeps_planes_in_step(monkeypatch):
    """A cfg2-size captured step (its party-GRU input products run on the plane kernels) replayed between FlatAdam steps: the graph
    holds no cut, FlatAdam's kernel writes the weights behind autograd's version counters, so the planes stay right only because
    FlatAdam invalidates them and CapturedStep.replay() re-cuts what its capture recorded.  Same losses as the run without the plane
    form (a stale plane set would miss a 1e-2 learning-rate step: orders of magnitude above the tolerance)."""
    from mm_dfn_amd import FocalLoss, ops_linear, synthetic, train
    from mm_dfn_amd.graphs import CapturedStep
    from mm_dfn_amd.optim import FlatAdam
    cfg = dict(synthetic.CONFIGS["cfg2"])
    batch = synthetic.make_batch(2021, device="cuda", **cfg)
    label = train.flatten_labels(batch["label"], batch["lengths"])
    loss_f = FocalLoss(gamma=0.5)

    def run(min_rows):
        monkeypatch.setattr(ops_linear, "PLANES_MIN_ROWS", min_rows)
        model = synthetic.build_model(dropout=0.0, **cfg)
        model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
        model = model.cuda().train()

        def fwd_bwd():
            logp = model(batch["textf"], batch["qmask"], batch["umask"], batch["lengths"], batch["acouf"], batch["visuf"])[0]
            loss = loss_f(logp, label)
            train.backward(loss)
            return loss
        model.zero_grad(set_to_none=True)
        fwd_bwd()
        opt = FlatAdam(model, lr=1e-2, weight_decay=1e-4)
        opt.bucket.flatten()
        opt._materialise()
        cap = CapturedStep(model, fwd_bwd, warmup=2, bucket=opt.bucket)
        used = len(cap._planes)
        losses = []
        for _ in range(4):
            losses.append(float(cap.replay()))
            opt.step(grads_already_flat=True)
        return losses, used

    with_planes, used = run(4096)
    without, unused = run(1 << 30)
    assert used >= 2 and unused == 0
    assert with_planes[0] != with_planes[-1]                       # the steps do move the loss
    for a, b in zip(with_planes, without):
        assert abs(a - b) <= 2e-4 * abs(b), (with_planes, without)


DEV = "cuda"


def test_grouped_launch_is_the_bits_of_the_single_launches():
    """mmdfn_linear_planes_group (ABI 16): every problem keeps its own tile form, so its result is bit-identical to its own launch."""
    g = torch.Generator(device=DEV).manual_seed(11)
    shapes = [(7040, 200, 300, 300), (1760, 200, 300, 300), (130, 100, 300, 300), (4100, 600, 100, 100)]   # (R, K, n1, n2)
    probs = []
    for R, K, n1, n2 in shapes:
        probs.append(dict(x=torch.randn(R, K, device=DEV, generator=g), w1=torch.randn(n1, K, device=DEV, generator=g) * 0.1,
                          w2=torch.randn(n2, K, device=DEV, generator=g) * 0.1, b1=torch.randn(n1, device=DEV, generator=g),
                          b2=torch.randn(n2, device=DEV, generator=g)))
    got = ops.linear_planes_group_raw(probs)
    for pr, y in zip(probs, got):
        want = ops.linear_planes_raw(pr["x"], pr["w1"], pr["w2"], pr["b1"], pr["b2"])
        assert torch.equal(y, want)
    # the input-gradient orientation (no biases), two problems
    dys = [torch.randn(7040, 600, device=DEV, generator=g), torch.randn(1760, 600, device=DEV, generator=g)]
    got = ops.linear_planes_group_raw([dict(x=dy, w1=pr["w1"], w2=pr["w2"]) for dy, pr in zip(dys, probs[:2])], transposed=True)
    for dy, pr, dx in zip(dys, probs[:2], got):
        assert torch.equal(dx, ops.linear_planes_raw(dy, pr["w1"], pr["w2"], transposed=True))


def test_linear2_group_node_against_separate_nodes():
    """_Linear2Group (the context + party input contractions of a GRU layer as one node) against one linear2 per group:
    outputs, input gradients and weight / bias gradients."""
    from mm_dfn_amd import train as T
    g = torch.Generator(device=DEV).manual_seed(12)
    mk = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.2)
    groups = []
    for rows in ((110, 16), (110, 64)):
        x = mk(*rows, 200).requires_grad_(True)
        prm = [torch.nn.Parameter(mk(300, 200)), torch.nn.Parameter(mk(300, 200)), torch.nn.Parameter(mk(300)), torch.nn.Parameter(mk(300))]
        groups.append([x] + prm)
    cot = [mk(*grp[0].shape[:-1], 600) for grp in groups]

    def run(joint):
        for grp in groups:
            for t in grp:
                t.grad = None
        if joint:
            ys = ops.linear2_group([tuple(grp) for grp in groups])
            assert ys is not None
        else:
            ys = [ops.linear2(*grp) for grp in groups]
        loss = sum((y * c).sum() for y, c in zip(ys, cot))
        T.backward(loss)
        return [y.detach().clone() for y in ys], [[t.grad.detach().clone() for t in grp] for grp in groups]

    y1, g1 = run(True)
    y0, g0 = run(False)
    for a, b in zip(y1, y0):
        # (the 1 760-row group alone takes the exact-f32 few-row kernel, inside the group the bf16-piece plane kernel:
        # both at fp32 rounding level)
        assert float((a - b).abs().max() / b.abs().max()) < 2e-6
    for ga, gb in zip(g1, g0):
        for a, b in zip(ga, gb):
            assert float((a - b).abs().max() / b.abs().max()) < 3e-6


def test_linear2_group_with_the_dropout_folded_in():
    """masks / scale: y_g = (x_g * mask_g * scale) W^T + b with the dropout's backward in the input-gradient launch's epilogue,
    against ops.mask_scale followed by the same node."""
    from mm_dfn_amd import train as T
    g = torch.Generator(device=DEV).manual_seed(13)
    mk = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.2)
    groups = []
    for rows in ((110, 16), (110, 64)):
        x = mk(*rows, 200).requires_grad_(True)
        prm = [torch.nn.Parameter(mk(300, 200)), torch.nn.Parameter(mk(300, 200)), torch.nn.Parameter(mk(300)), torch.nn.Parameter(mk(300))]
        groups.append([x] + prm)
    masks = [(torch.rand(grp[0].numel(), device=DEV, generator=g) < 0.6).float() for grp in groups]
    cot = [mk(*grp[0].shape[:-1], 600) for grp in groups]

    def run(folded):
        for grp in groups:
            for t in grp:
                t.grad = None
        if folded:
            ys = ops.linear2_group([tuple(grp) for grp in groups], masks=masks, scale=1.0 / 0.6)
        else:
            xs = ops.mask_scale([grp[0] for grp in groups], masks, 1.0 / 0.6)
            ys = ops.linear2_group([(x,) + tuple(grp[1:]) for x, grp in zip(xs, groups)])
        assert ys is not None
        T.backward(sum((y * c).sum() for y, c in zip(ys, cot)))
        return [y.detach().clone() for y in ys], [[t.grad.detach().clone() for t in grp] for grp in groups]

    y1, g1 = run(True)
    y0, g0 = run(False)
    for a, b in zip(y1, y0):
        assert torch.equal(a, b)
    for ga, gb in zip(g1, g0):
        for a, b in zip(ga, gb):
            assert float((a - b).abs().max() / b.abs().max()) < 1e-6
        # a dropped input element receives no gradient
    for grp, m, ga in zip(groups, masks, g1):
        assert float(ga[0].reshape(-1)[m == 0].abs().max()) == 0.0
