"""CPU: host-side logic of the drop-in modules (no HIP calls)."""
import os

import numpy as np
import pytest
import torch

import mmdfn_oracle as O
import mmdfn_vectorised as V
from mm_dfn_amd import synthetic, train
from mm_dfn_amd.dialogue_model import DialogueGNNModel
from mm_dfn_amd.layout import BlockTileAdjacency, DialogueLayout, pair_list

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_state_dict_keys_match_reference_fixture():
    want = [l.split() for l in open(os.path.join(GOLD, "state_dict_keys_iemocap.txt")).read().strip().splitlines()]
    m = synthetic.build_model(D_t=100, D_a=1582, D_v=342, P=2, C=6, nlayers=2)
    got = [[k, "x".join(map(str, v.shape))] for k, v in m.state_dict().items()]
    assert sorted(map(tuple, got)) == sorted(map(tuple, want))
    assert len(got) == 86


def test_layout_offsets():
    lay = DialogueLayout([5, 3, 8], 3, "cpu")
    assert lay.N == 16 and lay.max_len == 8 and lay.npairs == 3
    assert lay.row_start.tolist() == [0, 5, 8, 16]
    assert lay.tile_base.tolist() == [0, 3 * 5 * 8, 3 * 5 * 8 + 3 * 3 * 4, 3 * 5 * 8 + 3 * 3 * 4 + 3 * 8 * 8]
    assert lay.nnz == sum(3 * L * L + 6 * L for L in (5, 3, 8))
    assert lay.propagate_bytes(100) == 4 * lay.nnz + 8 * 3 * 16 * 100
    # L=110, M=3, d=100 -> 411 840 B per dialogue-layer
    assert DialogueLayout([110], 3, "cpu").propagate_bytes(100) == 411840


def test_block_tile_dense_roundtrip():
    rs = np.random.RandomState(0)
    lengths, M = [4, 7, 1], 3
    lay = DialogueLayout(lengths, M, "cpu")
    tiles = [torch.from_numpy(rs.randn(L, L).astype(np.float32)) for L in lengths for _ in range(M)]
    cross = torch.from_numpy(rs.randn(lay.npairs, lay.N).astype(np.float32))
    adj = BlockTileAdjacency.from_parts(lay, tiles, cross)
    flat = torch.cat([t.reshape(-1) for t in tiles])
    # oracle packs without row padding; compare through the dense form
    dense = adj.to_dense()
    N = lay.N
    assert torch.equal(dense[0:4, 0:4], tiles[0])
    assert torch.equal(dense[N + 4:N + 11, N + 4:N + 11], tiles[4])
    assert torch.equal(dense[2 * N + 11, 2 * N + 11], tiles[8][0, 0])
    for k, (m, n) in enumerate(pair_list(M)):
        assert torch.equal(dense[m * N + 3, n * N + 3], cross[k, 3]) and torch.equal(dense[n * N + 3, m * N + 3], cross[k, 3])
    assert float(dense[0, 5]) == 0.0


@pytest.mark.parametrize("P,lengths", [(2, [15, 9, 1, 6]), (9, [12, 12, 3]), (3, [1])])
def test_party_gather_scatter_match_oracle(P, lengths):
    """The cumulative-sum gather/scatter plan (oracle/mmdfn_vectorised.py: the index-op composition the K3/K4 kernels
    are checked against on the GPU) == the reference's per-(b,p) loops.  The GRU is applied by the TEST (torch CPU)."""
    cfg = dict(B=len(lengths), L=max(lengths), P=P, C=6, nlayers=2, D_t=100, D_a=32, D_v=64)
    m = synthetic.build_model(**cfg).eval()
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 3))
    b = synthetic.make_batch(4, lengths=lengths, **cfg)
    X = [torch.randn(max(lengths), len(lengths), 200) for _ in range(3)]
    with torch.no_grad():
        plan = V.party_plan(b["qmask"])
        S = V.party_gather(torch.stack(X, 0), plan)
        got = V.party_scatter(m.rnn_parties(S)[0], plan, 3)
        for i in range(3):
            want = O.party_encode(X[i], b["qmask"], dict(m.state_dict()), engine="aten")
            assert (got[i] - want).abs().max() < 1e-5


def test_model_on_cpu_fails_loudly():
    from mm_dfn_amd import _hip
    cfg = dict(B=2, L=5, P=2, C=6, nlayers=2, D_t=100, D_a=32, D_v=64)
    m = synthetic.build_model(**cfg).eval()
    b = synthetic.make_batch(4, lengths=[5, 3], **cfg)
    with pytest.raises(_hip.HipLibraryError):
        m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])


def test_party_plan_non_one_hot_last_speaker_wins():
    """The reference scatters speaker by speaker (model.py:1084-1087): with two speakers flagged on one
    utterance the later speaker's encoding overwrites the earlier one."""
    cfg = dict(B=1, L=6, P=2, C=6, nlayers=2, D_t=100, D_a=32, D_v=64)
    m = synthetic.build_model(**cfg).eval()
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 5))
    b = synthetic.make_batch(6, lengths=[6], **cfg)
    q = b["qmask"].clone()
    q[2, 0, :] = 1.0
    X = torch.randn(6, 1, 200)
    with torch.no_grad():
        plan = V.party_plan(q)
        got = V.party_scatter(m.rnn_parties(V.party_gather(X.unsqueeze(0), plan))[0], plan, 1)[0]
        # literal restatement of the reference loops
        U_ = X.transpose(0, 1)
        q_ = q.transpose(0, 1)
        parts = [torch.zeros_like(U_) for _ in range(2)]
        for p in range(2):
            idx = torch.nonzero(q_[0][:, p]).squeeze(-1)
            parts[p][0][:idx.numel()] = U_[0][idx]
        E = [m.rnn_parties(parts[p].transpose(0, 1))[0].transpose(0, 1) for p in range(2)]
        want = torch.zeros(1, 6, 200)
        for p in range(2):
            idx = torch.nonzero(q_[0][:, p]).squeeze(-1)
            want[0][idx] = E[p][0][:idx.numel()]
    assert (got.transpose(0, 1) - want).abs().max() < 1e-6


def test_lengths_and_label_flatten():
    b = synthetic.make_batch(1, B=4, L=9, P=2, C=6, D_t=8, D_a=8, D_v=8, lengths=[9, 2, 5, 1])
    assert train.lengths_from_umask(b["umask"]) == [9, 2, 5, 1] == O.lengths_from_umask(b["umask"])
    assert torch.equal(train.flatten_labels(b["label"], [9, 2, 5, 1]), O.flatten_labels(b["label"], [9, 2, 5, 1]))


def test_synthetic_batch_contract():
    b = synthetic.make_batch(2, ragged=True, **synthetic.CONFIGS["cfg3"])
    L, B = b["textf"].shape[:2]
    assert b["visuf"].shape == (L, B, 342) and b["acouf"].shape == (L, B, 300) and b["qmask"].shape == (L, B, 9)
    assert b["umask"].shape == (B, L) and b["label"].shape == (B, L) and max(b["lengths"]) == L
    for j, n in enumerate(b["lengths"]):
        assert float(b["textf"][n:, j].abs().sum()) == 0 and float(b["qmask"][n:, j].sum()) == 0
        assert torch.equal(b["qmask"][:n, j].sum(1), torch.ones(n))


def test_out_of_scope_configurations_raise():
    with pytest.raises(NotImplementedError):
        DialogueGNNModel('DialogRNN', 100, 150, 150, 100, 100, 100, 100, 2, 200, 10, 10, graph_type='GDF')
    with pytest.raises(NotImplementedError):
        DialogueGNNModel('LSTM', 100, 150, 150, 100, 100, 100, 100, 2, 200, 10, 10, graph_type='relation')


def test_pinned_lru_evicts_one_at_a_time_and_records_capture_users():
    from mm_dfn_amd.layout import PinnedLRU, recording
    c = PinnedLRU(3)
    made = []
    mk = lambda k: (lambda: made.append(k) or ("val", k))
    with recording(c) as used:
        a = c.get("a", mk("a"))
        c.get("b", mk("b"))
    assert used == [("val", "a"), ("val", "b")]
    c.get("c", mk("c"))
    c.get("a", mk("a"))                   # hit: refreshes 'a'
    c.get("d", mk("d"))                   # evicts 'b' only (least recently used)
    assert list(c.data) == ["c", "a", "d"] and made == ["a", "b", "c", "d"]
    assert c.get("a", mk("a")) is a
    c.get("b", mk("b"))                   # 'b' is rebuilt, 'c' goes
    assert made[-1] == "b" and "c" not in c.data and len(c) == 3
    # many signatures later the objects recorded during the "capture" are still the caller's to keep alive
    for i in range(10):
        c.get(i, mk(i))
    assert used[0] is a and "a" not in c.data


def test_dialogue_layout_cache_is_lru():
    from mm_dfn_amd import layout
    layout._LAYOUT_CACHE.clear()
    first = DialogueLayout.get([3, 2], 3, "cpu")
    for n in range(70):
        DialogueLayout.get([n + 1], 3, "cpu")
        assert DialogueLayout.get([3, 2], 3, "cpu") is first      # kept by use, not wiped with the rest
    assert len(layout._LAYOUT_CACHE) <= 64


def test_focal_loss_alpha_table_must_cover_the_classes():
    from mm_dfn_amd import FocalLoss
    f = FocalLoss(gamma=1.0, alpha=0.25)                     # scalar alpha -> 2-entry table (loss.py:9)
    assert f.alpha.numel() == 2
    lp = torch.log_softmax(torch.randn(4, 2), 1)
    assert torch.isfinite(f(lp, torch.tensor([0, 1, 1, 0])))


def test_flag_pool_shares_one_draw_per_step(monkeypatch):
    """ops.keep_flags inside a flag_pool scope: the first step draws per request, later steps with the same key draw once
    and hand out disjoint 16-byte aligned slices; outside a scope every request is its own draw.  (Host logic only: the draw
    itself -- the device's Philox kernel, tests/test_gru_gpu.py -- is replaced by a CPU stand-in here.)"""
    import torch
    from mm_dfn_amd import ops, ops_flags
    monkeypatch.setattr(ops_flags, "draw_flags", lambda n, p, device: torch.empty(n, device=device).bernoulli_(1.0 - p))
    dev = torch.device("cpu")
    a = ops.keep_flags(10, 0.5, dev)
    assert a.shape == (10,) and set(a.unique().tolist()) <= {0.0, 1.0}
    bufs = []
    for step in range(3):
        with ops.flag_pool(("k", 1)):
            x = ops.keep_flags(1001, 0.5, dev)
            y = ops.keep_flags(64, 0.5, dev)
            z = ops.keep_flags(7, 0.25, dev)          # another p: its own buffer
            with ops.flag_pool("inner"):               # nested scopes join the outer one
                u = ops.keep_flags(30, 0.5, dev)
        same = x.untyped_storage().data_ptr() == y.untyped_storage().data_ptr() == u.untyped_storage().data_ptr()
        assert same == (step > 0)
        assert z.untyped_storage().data_ptr() != x.untyped_storage().data_ptr()
        if step > 0:
            assert y.storage_offset() == 1004 and u.storage_offset() == 1068
            assert x.untyped_storage().nbytes() == 4 * (1004 + 64 + 32)
        bufs.append(x)
        assert 0.3 < float(x.mean()) < 0.7
    assert bufs[1].untyped_storage().data_ptr() != bufs[2].untyped_storage().data_ptr()   # a fresh buffer per step
    assert ops_flags._FLAG_SCOPE is None
    # a step that asks for more than the hint: the overflow request draws its own buffer
    with ops.flag_pool(("k", 1)):
        x = ops.keep_flags(1001, 0.5, dev)
        big = ops.keep_flags(5000, 0.5, dev)
    assert big.shape == (5000,) and big.untyped_storage().data_ptr() != x.untyped_storage().data_ptr()


def test_bucket_order_keeps_gru_direction_pairs_adjacent_and_flat_views_stack():
    """distributed.bucket_order: every weight_ih / bias_ih parameter of a bidirectional GRU is directly followed by its
    *_reverse twin (FlatAdam lays its flat buffer out in this order), so gru._stacked_view can hand the fused path ONE
    (600, K) operand without a copy; non-adjacent CPU parameters are never re-pointed (None -> the copying path)."""
    import torch
    from mm_dfn_amd import distributed, gru, synthetic
    cfg = dict(synthetic.CONFIGS["cfg1"]) if "cfg1" in synthetic.CONFIGS else dict(B=2, L=8, P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    m = synthetic.build_model(dropout=0.0, **cfg)
    live = [p for p in m.parameters()]
    order = distributed.bucket_order(m, live)
    names = {id(p): n for n, p in m.named_parameters()}
    ns = [names[id(p)] for p in order]
    assert sorted(ns) == sorted(names.values()) and len(set(ns)) == len(ns)
    pairs = 0
    for i, n in enumerate(ns):
        if (".weight_ih_l" in "." + n or ".bias_ih_l" in "." + n) and not n.endswith("_reverse") and n + "_reverse" in ns:
            assert ns[i + 1] == n + "_reverse", (n, ns[i + 1])
            pairs += 1
    assert pairs >= 8            # two GRUs x two layers x (weight, bias)
    # a flat buffer in that order makes the pairs adjacent
    flat = torch.cat([p.detach().reshape(-1) for p in order])
    off = 0
    views = {}
    for p in order:
        views[names[id(p)]] = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    wf, wr = views["lstm_l.weight_ih_l0"], views["lstm_l.weight_ih_l0_reverse"]
    assert gru._adjacent(wf, wr)
    sv = gru._stacked_view(wf, wr)
    assert sv.shape == (600, wf.shape[1]) and torch.equal(sv, torch.cat([wf, wr], 0)) and sv.data_ptr() == wf.data_ptr()
    bf, br = views["lstm_l.bias_ih_l0"], views["lstm_l.bias_ih_l0_reverse"]
    assert torch.equal(gru._stacked_view(bf, br), torch.cat([bf, br], 0))
    # the module's own (separately allocated, CPU) parameters: not adjacent, and not re-pointed on the CPU
    before = m.lstm_l.weight_ih_l0.data_ptr()
    assert gru._stacked_view(m.lstm_l.weight_ih_l0, m.lstm_l.weight_ih_l0_reverse) is None
    assert m.lstm_l.weight_ih_l0.data_ptr() == before


def test_backward_with_cached_unit_seed_matches_plain_backward():
    import torch
    from mm_dfn_amd import train
    w = torch.randn(5, requires_grad=True)
    (w * w).sum().backward()
    g0 = w.grad.clone()
    w.grad = None
    train.backward((w * w).sum())
    assert torch.equal(w.grad, g0)
    w.grad = None
    train.backward((w * w).sum())            # second call: the cached seed
    assert torch.equal(w.grad, g0)


def test_only_feature_tensors_are_row_padded():
    """Row padding (ops.py) is for contraction operands: an (L, B, D) feature tensor with D % 4 != 0.  The speaker mask
    (L, B, P) has the same rank and dtype and stays as it is -- padding it made every step copy it back."""
    from mm_dfn_amd import ops
    assert ops.is_odd_feature_tensor(torch.zeros(5, 3, 342), True)
    assert ops.is_odd_feature_tensor(torch.zeros(5, 3, 1582), True)
    assert ops.is_odd_feature_tensor(torch.zeros(5, 3, 10), True)               # a narrow feature stream is still a feature stream
    assert not ops.is_odd_feature_tensor(torch.zeros(5, 3, 100), True)
    assert not ops.is_odd_feature_tensor(torch.zeros(5, 3, 2), False)           # qmask, two speakers
    assert not ops.is_odd_feature_tensor(torch.zeros(5, 3, 9), False)           # qmask, MELD
    assert not ops.is_odd_feature_tensor(torch.zeros(5, 3, 17), False)          # qmask of a 17-speaker corpus: the role decides
    assert not ops.is_odd_feature_tensor(torch.zeros(3, 5), False)              # umask
    assert not ops.is_odd_feature_tensor(torch.zeros(5, 3, 342, dtype=torch.int64), True)
    # the role is the slot in the reference's batch tuple (textf, visuf, acouf, qmask, umask, label)
    assert ops.FEATURE_SLOTS == (0, 1, 2)


def test_gru_launch_form_thresholds():
    """Which form a GRU launch takes is decided from the row count alone (no device data): up to one workgroup slot per
    sequence-direction the plain launches, above that the valid-length (merged-chain) launches, and beyond
    MFMA_MIN_CHAINS sequence-directions the MFMA form at full length (csrc/gru_mfma.hip: its launch time does not depend on
    the chain count, so truncation has nothing to remove there)."""
    from mm_dfn_amd import gru
    if gru.TRUNCATE != "auto":
        pytest.skip("MMDFN_GRU_TRUNCATE overrides the rule")
    assert not gru.wants_truncation(80)                     # cfg2: 160 chains on 256 CUs
    assert gru.wants_truncation(288)                        # cfg4: 576 chains
    assert gru.wants_truncation(gru.MFMA_MIN_CHAINS // 2)   # the last size the scalar kernels keep
    assert not gru.wants_truncation(gru.MFMA_MIN_CHAINS // 2 + 1)
    assert not gru.wants_truncation(960)                    # cfg3: 1 920 chains -> MFMA form


def test_foreign_slab_stacks_get_gradient_destinations():
    """ops._ext_destinations: slab stacks of other kernels that ride on the weight-gradient batch's reduction launch get a fresh
    .grad (written) or accumulate into the existing one; a stack with its own buffer (the two bias halves of the
    project-then-gather node) is written even though its stand-in already holds a buffer."""
    from mm_dfn_amd import ops
    w = torch.nn.Parameter(torch.zeros(6, 8))
    b = torch.nn.Parameter(torch.zeros(6))
    e = dict(part=torch.zeros(4, 6, 8), colpart=torch.zeros(4, 6), splits=4, M=6, N=8, weight=w, bias=b)
    (_, C, cs, acc), = ops._ext_destinations([e])
    assert acc == 0 and C.data_ptr() == w.grad.data_ptr() and cs.data_ptr() == b.grad.data_ptr() and tuple(C.shape) == (6, 8)
    (_, C2, cs2, acc2), = ops._ext_destinations([e])                       # second backward without zero_grad: accumulate
    assert acc2 == 1 and C2.data_ptr() == C.data_ptr()
    w3 = torch.nn.Parameter(torch.zeros(6, 8))                             # one of the two exists: the missing one starts at zero
    w3.grad = torch.ones(6, 8)
    b3 = torch.nn.Parameter(torch.zeros(6))
    (_, C4, cs4, acc4), = ops._ext_destinations([dict(e, weight=w3, bias=b3)])
    assert acc4 == 1 and float(cs4.abs().sum()) == 0.0                     # the fresh bias buffer is zero-filled, then added to
    buf = torch.empty(10)
    half = ops._HalvesGrad(buf)
    (_, C5, cs5, acc5), = ops._ext_destinations([dict(part=None, colpart=torch.zeros(3, 10), splits=3, M=10, N=0, weight=None,
                                                      bias=half, acc=0)])
    assert C5 is None and cs5 is buf and acc5 == 0


def test_rejected_retarget_leaves_layout_and_scope_untouched():
    """a batch that does not fit a bucketed step's index scope must be refused BEFORE any host state or device array
    is rewritten (the captured step keeps describing the batch its arrays hold)."""
    from mm_dfn_amd.layout import IndexScope
    scope = IndexScope()
    with scope:
        lay = DialogueLayout.get([4, 3, 3], 3, "cpu")
        pos = scope.tensor(("pos", 0), [4, 3, 3], lambda lens: np.cumsum(np.asarray(lens, dtype=np.int64)), "cpu")
    snap = (list(lay.lengths), lay.B, lay.N, lay.max_len, lay.row_start_host.copy(), lay._i32.clone(), lay.tile_base.clone(),
            pos.clone())
    for bad in ([5, 3, 2], [4, 3, 2], [4, 3, 2, 1], []):
        with pytest.raises((RuntimeError, ValueError)):
            scope.retarget(bad)
        assert (list(lay.lengths), lay.B, lay.N, lay.max_len) == snap[:4]
        assert np.array_equal(lay.row_start_host, snap[4])
        assert torch.equal(lay._i32, snap[5]) and torch.equal(lay.tile_base, snap[6]) and torch.equal(pos, snap[7])
    scope.retarget([3, 4, 3])                              # same (B, N, max_len): accepted, everything follows
    assert lay.lengths == [3, 4, 3] and lay._i32[:3].tolist() == [3, 4, 3] and pos.tolist() == [3, 7, 10]


def test_grad_addends_of_a_backward_that_raised_do_not_block_later_passes():
    """addends queued by a backward pass that raised before its end-of-backward callback must not leave the
    'callback queued' state behind (later passes would silently drop their addends)."""
    from mm_dfn_amd import ops
    p = torch.nn.Parameter(torch.zeros(3))
    ops._GRAD_ADDENDS.append((p, torch.ones(3), None))      # what a failed pass leaves behind
    ops._GRAD_ADDENDS_ARMED[0] = True
    with ops.wgrad_batch():
        assert not ops._GRAD_ADDENDS and not ops._GRAD_ADDENDS_ARMED[0]
    with pytest.raises(ZeroDivisionError):
        with ops.wgrad_batch():
            ops._GRAD_ADDENDS.append((p, torch.ones(3), None))
            ops._GRAD_ADDENDS_ARMED[0] = True
            1 / 0
    assert not ops._GRAD_ADDENDS and not ops._GRAD_ADDENDS_ARMED[0]
