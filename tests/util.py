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
