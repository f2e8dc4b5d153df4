"""K5s (csrc/adjacency_small.hip): the strip-workgroup form of the adjacency build for short dialogues against

  * the CPU oracle (create_big_adj, reference model_mm.py:122-180) -- the parity bar of tests/test_graph_kernels_gpu.py:
    adjacency 2e-5 absolute (acos near the unit diagonal), gradients 1e-4 relative;
  * the many-launch form of csrc/adjacency.hip on the same inputs (both strip heights forced through the tuning build's
    switches): the two forms differ by summation order only -- adjacency entries within 4e-6, d(features) within 2e-6 relative
    with the stack's gradient added in, 6e-5 for the adjacency path alone (each form's own distance from fp64).
"""
import numpy as np
import pytest
import torch

import mmdfn_oracle as O
from mm_dfn_amd import _hip, ops
from mm_dfn_amd.layout import DialogueLayout
from mm_dfn_amd.ops_pad import _lay_args
from util import abs_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (lengths, M, D): one row, one strip + one row, exactly one / two / four strips, the longest tile, ragged batches, one and
# two modalities, the widest features, more than eight dialogues (block decode), a width that is not a multiple of 16
SHAPES = [([1], 3, 200), ([33, 1, 32], 3, 200), ([64, 65], 3, 200), ([128, 3], 3, 200), ([127, 128], 2, 64),
          ([110, 97, 64, 33, 80, 71, 45, 27, 102, 58, 39], 3, 200), ([50, 20], 1, 100), ([17, 96], 3, 256),
          ([40, 41, 42], 3, 36), ([110] * 16, 3, 200)]


def _buffers(lay, M, N, D, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    f32 = dict(dtype=torch.float32, device=DEV)
    nan = float("nan")
    return dict(feats=torch.randn(M, N, D, device=DEV, generator=g), unit=torch.full((M, N, D), nan, **f32),
                norm=torch.full((M, N), nan, **f32), cosg=torch.full((lay.tile_elems,), nan, **f32),
                cdot=torch.full((lay.npairs, N), nan, **f32), rdeg=torch.full((M, N), nan, **f32),
                tiles=torch.full((lay.tile_elems,), nan, **f32), cross=torch.full((lay.npairs, N), nan, **f32),
                dtiles=torch.randn(lay.tile_elems, device=DEV, generator=g), dcross=torch.randn(lay.npairs, N, device=DEV, generator=g),
                wsym=torch.empty(lay.tile_elems, **f32), etile=torch.empty(lay.tile_elems, **f32),
                ecross=torch.empty(lay.npairs, N, **f32), ddeg=torch.empty(M, N, **f32), dunit=torch.empty(M, N, D, **f32),
                dfeats=torch.full((M, N, D), nan, **f32), addend=torch.randn(M, N, D, device=DEV, generator=g))


def _run(b, lay, M, N, D, mw, addend=True):
    P = _hip.ptr
    rc = _hip.lib().mmdfn_adj_build(P(b["feats"]), P(b["unit"]), P(b["norm"]), P(b["cosg"]), P(b["cdot"]), P(b["rdeg"]),
                                    P(b["tiles"]), P(b["cross"]), *_lay_args(lay), lay.B, M, N, D, lay.max_len, mw, _hip.stream())
    _hip.check(rc, "mmdfn_adj_build")
    rc = _hip.lib().mmdfn_adj_build_bwd(P(b["dtiles"]), P(b["dcross"]), P(b["unit"]), P(b["norm"]), P(b["cosg"]), P(b["cdot"]),
                                        P(b["rdeg"]), P(b["tiles"]), P(b["cross"]), P(b["wsym"]), P(b["etile"]), P(b["ecross"]),
                                        P(b["ddeg"]), P(b["dunit"]), P(b["dfeats"]), P(b["addend"] if addend else None),
                                        *_lay_args(lay), lay.B, M, N, D, lay.max_len, mw, _hip.stream())
    _hip.check(rc, "mmdfn_adj_build_bwd")
    torch.cuda.synchronize()


def _valid_tile_mask(lay, M):
    """True where a tile-array element is a stored entry or a padding column the kernels must write (zero)."""
    mask = torch.zeros(lay.tile_elems, dtype=torch.bool)
    for i, L in enumerate(lay.lengths):
        ld, base = int(lay.ld_host[i]), int(lay.tile_base_host[i])
        mask[base:base + M * L * ld] = True
    return mask.to(DEV)


@pytest.mark.parametrize("lengths,M,D", SHAPES)
@pytest.mark.parametrize("sr", [32, 64])
def test_strip_form_matches_the_many_launch_form(lengths, M, D, sr, kernel_variants):
    N = sum(lengths)
    lay = DialogueLayout.get(lengths, M, torch.device(DEV))
    kernel_variants.setenv("MMDFN_ADJ_SMALL", "0")
    ref = _buffers(lay, M, N, D, 7)
    _run(ref, lay, M, N, D, 0.7)
    kernel_variants.setenv("MMDFN_ADJ_SMALL", "1")
    kernel_variants.setenv("MMDFN_ADJ_SR", str(sr))
    got = _buffers(lay, M, N, D, 7)
    _run(got, lay, M, N, D, 0.7)
    written = _valid_tile_mask(lay, M)
    for name, tol in (("tiles", 4e-6), ("cosg", 2e-6), ("cross", 4e-6), ("cdot", 2e-6), ("rdeg", 2e-6), ("unit", 1e-6), ("norm", 1e-4)):
        a, o = got[name], ref[name]
        if name in ("tiles", "cosg"):
            a, o = a[written], o[written]
        if a.numel() == 0:
            continue                                   # (one modality: no cross diagonals)
        assert not torch.isnan(a).any(), name          # every element the many-launch form writes is written (padding columns = 0)
        assert float((a - o).abs().max()) <= tol, name
    assert not torch.isnan(got["dfeats"]).any()
    assert rel_err(got["dfeats"], ref["dfeats"]) < 2e-6
    # padding columns of every tile row hold zeros
    for i, L in enumerate(lengths):
        ld, base = int(lay.ld_host[i]), int(lay.tile_base_host[i])
        if ld > L:
            t = got["tiles"][base:base + M * L * ld].view(M * L, ld)
            assert float(t[:, L:].abs().max()) == 0.0
    # without the second gradient path
    got2 = _buffers(lay, M, N, D, 7)
    _run(got2, lay, M, N, D, 0.7, addend=False)
    kernel_variants.setenv("MMDFN_ADJ_SMALL", "0")
    ref2 = _buffers(lay, M, N, D, 7)
    _run(ref2, lay, M, N, D, 0.7, addend=False)
    # (the adjacency gradient alone: both forms sit 2-5e-5 from an fp64 evaluation -- the fp32 forward's acos near the unit
    # diagonal -- and so up to that far from each other; tools/adj_forms_vs_f64.py prints the two errors side by side)
    assert rel_err(got2["dfeats"], ref2["dfeats"]) < 6e-5


@pytest.mark.parametrize("lengths,M,D", [([33, 1, 32], 3, 200), ([110, 64, 33], 3, 200), ([128, 3], 2, 64), ([40, 41, 42], 3, 36)])
@pytest.mark.parametrize("modal_weight", [1.0, 0.7])
def test_strip_form_against_the_oracle(lengths, M, D, modal_weight):
    """Production library (the form is chosen from the shape: all of these take the strip form)."""
    rs = np.random.RandomState(31)
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(M, N, D).astype(np.float32))
    R = torch.from_numpy(rs.randn(M * N, M * N).astype(np.float32))
    fo = feats.clone().requires_grad_(True)
    want = O.create_big_adj([fo[m] for m in range(M)], lengths, modal_weight)
    (want * R).sum().backward()
    fg = feats.to(DEV).requires_grad_(True)
    adj = ops.build_adjacency(fg, lengths, modal_weight)
    assert abs_err(adj.to_dense(), want.detach()) < 2e-5
    (adj.to_dense() * R.to(DEV)).sum().backward()
    assert rel_err(fg.grad, fo.grad) < 1e-4


def test_strip_form_is_bit_reproducible_and_independent_of_the_batch():
    """Same dialogue alone and inside a batch (other strip height, other block position): identical bits; two runs: identical bits."""
    rs = np.random.RandomState(5)
    M, D = 3, 200
    lengths = [110, 64, 97, 33, 80, 71, 45, 27, 102]
    N = sum(lengths)
    feats = torch.from_numpy(rs.randn(M, N, D).astype(np.float32)).to(DEV)
    a1 = ops.build_adjacency(feats, lengths, 1.0)
    a2 = ops.build_adjacency(feats, lengths, 1.0)
    assert torch.equal(a1.tiles, a2.tiles) and torch.equal(a1.cross, a2.cross)
    start = sum(lengths[:2])
    solo = ops.build_adjacency(feats[:, start:start + 97].contiguous(), [97], 1.0)
    lay = a1.layout
    base, ld = int(lay.tile_base_host[2]), int(lay.ld_host[2])
    assert torch.equal(a1.tiles[base:base + M * 97 * ld], solo.tiles)
    assert torch.equal(a1.cross[:, start:start + 97], solo.cross)
