"""Shared helpers for the parity tests (oracle = checker, HIP path = thing checked)."""
import numpy as np
import torch

import mmdfn_oracle as O
from mm_dfn_amd import synthetic
from mm_dfn_amd.layout import BlockTileAdjacency, DialogueLayout, pair_list


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def abs_err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def random_block_adjacency(seed, lengths, M, device):
    """Random NON-symmetric tiles + cross diagonals; returns (BlockTileAdjacency on device, dense CPU matrix)."""
    rs = np.random.RandomState(seed)
    lay = DialogueLayout.get(lengths, M, device)
    tiles = [torch.from_numpy(rs.uniform(-1, 1, size=(L, L)).astype(np.float32)) for L in lengths for _ in range(M)]
    cross = torch.from_numpy(rs.uniform(-1, 1, size=(lay.npairs, lay.N)).astype(np.float32))
    adj = BlockTileAdjacency.from_parts(lay, tiles, cross, device=device)
    N = lay.N
    dense = torch.zeros(M * N, M * N)
    it = iter(tiles)
    start = 0
    for L in lengths:
        for m in range(M):
            dense[m * N + start:m * N + start + L, m * N + start:m * N + start + L] = next(it)
        start += L
    ar = torch.arange(N)
    for k, (m, n) in enumerate(pair_list(M)):
        dense[m * N + ar, n * N + ar] = cross[k]
        dense[n * N + ar, m * N + ar] = cross[k]
    return adj, dense, tiles, cross


def oracle_params(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def model_and_batch(cfg, seed, ragged, device, lengths=None, dropout=0.0):
    model = synthetic.build_model(dropout=dropout, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), seed))
    batch = synthetic.make_batch(seed + 1, ragged=ragged, lengths=lengths, **cfg)
    return model.to(device), batch


def relu_flips_from_tap(tap, probe, prefix, M, N, limit=16, band=1e-5):
    """ReLU units on which the device and the oracle sit on different sides of the kink.  ``tap``: one entry of
    mm_dfn_amd.gcn_stack.TAP (the fused node's h0, per-layer gate masks and output); ``probe``: the oracle's ReluProbe of
    the same forward.  Every disagreement must be a pre-activation within ``band`` of zero (anything larger is a real
    error and fails here) and there may be at most ``limit`` of them.  Returns {site: LongTensor (k, 2)} for
    O.ReluProbe(flips=...)."""
    flips, total = {}, 0

    def site(name, dev_on, pre):
        nonlocal total
        diff = (dev_on.cpu() != (pre > 0)).nonzero()
        if len(diff):
            worst = float(pre[diff[:, 0], diff[:, 1]].abs().max())
            assert worst < band, "%s: device and oracle disagree on a ReLU whose pre-activation is %.3g" % (name, worst)
            flips[name] = diff
            total += len(diff)

    site(prefix + "fcs0", tap["h0"] > 0, probe.pre[prefix + "fcs0"])
    for i, g in enumerate(tap["gmask"]):
        site(prefix + "conv%d" % i, g > 0, probe.pre[prefix + "conv%d" % i])
    out = tap["out"]                                   # (M N, W) stacked -> the head's (N, M W) input
    W = out.shape[1]
    head_on = (out > 0).view(M, N, W).permute(1, 0, 2).reshape(N, M * W)
    site("head", head_on, probe.pre["head"])
    assert total <= limit, "%d ReLU units differ between device and oracle" % total
    return flips
