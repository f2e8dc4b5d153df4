"""GPU: pickle -> loaders -> (prefetched) batches -> train_or_eval_graph_model / fit on the HIP model."""
import numpy as np
import pytest
import torch

from mm_dfn_amd import FocalLoss, synthetic
from mm_dfn_amd import data as D
from mm_dfn_amd import train as T

pytestmark = pytest.mark.gpu
CFG = dict(P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)


def _model(seed=5, dropout=0.0):
    m = synthetic.build_model(dropout=dropout, **CFG)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), seed))
    return m.cuda()


def test_prefetcher_yields_the_loader_batches_on_device(tmp_path):
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=13, n_test=5, max_len=30, seed=2)
    _, _, te = D.get_IEMOCAP_loaders(p, batch_size=2, valid_rate=0.0)
    plain = list(te)
    pre = list(D.DevicePrefetcher(te, depth=2))
    assert len(plain) == len(pre) == 3
    for a, b in zip(plain, pre):
        for x, y in zip(a[:6], b[:6]):
            assert y.is_cuda and torch.equal(x, y.cpu())
        assert a[6] == b[6]


def test_eval_pass_same_through_prefetcher_and_bucketing(tmp_path):
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=4, n_test=21, max_len=40, seed=4)
    m = _model()
    loss_f = FocalLoss(gamma=0.5)
    names = ['hap', 'sad', 'neu', 'ang', 'exc', 'fru']
    _, _, te = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0)
    _, _, te_b = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0, bucketed=True)
    r0 = T.train_or_eval_graph_model(m, loss_f, te, cuda_flag=True, target_names=names)
    r1 = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(te), cuda_flag=False, target_names=names)
    r2 = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(te_b), cuda_flag=False, target_names=names)
    assert r0[2] == r1[2] and r0[3] == r1[3] and r0[6] == r1[6] and np.array_equal(r0[5], r1[5])
    # bucketing regroups the dialogues (dialogues are independent, so per-utterance predictions are unchanged):
    # same multiset of (label, pred) pairs, same accuracy / F1
    assert r2[3] == r0[3] and r2[6] == r0[6]
    assert sorted(zip(r0[4].tolist(), r0[5].tolist())) == sorted(zip(r2[4].tolist(), r2[5].tolist()))


def test_fit_trains_and_stops(tmp_path):
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=20, n_test=6, max_len=24, seed=6)
    m = _model(dropout=0.1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
    tr, va, te = D.get_IEMOCAP_loaders(p, batch_size=8, valid_rate=0.2, bucketed=True)
    wrap = D.DevicePrefetcher
    out = T.fit(m, FocalLoss(gamma=0.5), opt, wrap(tr), wrap(va), wrap(te), n_epochs=6, patience=2, valid_rate=0.2,
                log=None)
    h = out["history"]
    assert 1 <= out["epochs_run"] <= 6 and all(np.isfinite(h["train_loss"]))
    assert h["train_loss"][-1] < h["train_loss"][0]      # it learns the (memorisable) synthetic labels
    assert 0 <= out["by_f1"]["epoch"] < out["epochs_run"]


def test_load_reference_weights_roundtrip(tmp_path):
    m = _model(seed=9)
    path = str(tmp_path / "sd.pt")
    torch.save({k: v.cpu() for k, v in m.state_dict().items()}, path)
    m2 = synthetic.build_model(**CFG).cuda()
    T.load_reference_weights(m2, path)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_focal_loss_on_device_rejects_bad_alpha_and_poisons_bad_labels():
    lp = torch.log_softmax(torch.randn(5, 6, device="cuda"), 1).requires_grad_(True)
    with pytest.raises(IndexError):
        FocalLoss(gamma=1.0, alpha=0.25)(lp, torch.zeros(5, dtype=torch.int64, device="cuda"))   # 2-entry table, 6 classes
    tgt = torch.tensor([0, 5, 6, 1, 2], device="cuda")                                           # 6 is out of range
    loss = FocalLoss(gamma=0.5)(lp, tgt)
    assert torch.isnan(loss)                  # the reference raises in gather (loss.py:23); a kernel poisons instead
    loss.backward()
    assert torch.isnan(lp.grad[2]).all() and torch.isfinite(lp.grad[[0, 1, 3, 4]]).all()


def test_flat_adam_state_dict_roundtrip_and_lr_schedule():
    from mm_dfn_amd.optim import FlatAdam
    b = synthetic.make_batch(3, lengths=[9, 4], B=2, L=9, **CFG)
    loss_f = FocalLoss(gamma=0.5)

    def step(m, opt):
        opt.zero_grad()
        lp = m(b["textf"].cuda(), b["qmask"].cuda(), b["umask"].cuda(), b["lengths"], b["acouf"].cuda(), b["visuf"].cuda())[0]
        loss_f(lp, T.flatten_labels(b["label"].cuda(), b["lengths"])).backward()
        opt.step()

    m1, m2 = _model(11), _model(11)
    o1 = FlatAdam(m1, lr=1e-3, weight_decay=1e-4)
    for _ in range(2):
        step(m1, o1)
    sd_opt, sd_model = o1.state_dict(), {k: v.clone() for k, v in m1.state_dict().items()}
    assert sd_opt["step"] == 2 and "smax_fc.weight" in sd_opt["state"] and "gatedatt.transform_l.weight" not in sd_opt["state"]
    o1.param_groups[0]["lr"] = 5e-4                        # what an LR scheduler does
    step(m1, o1)
    # resume: fresh model + optimizer, weights and moments restored, same third step
    m2.load_state_dict(sd_model)
    o2 = FlatAdam(m2, lr=123.0)
    m2.zero_grad(set_to_none=True)
    lp = m2(b["textf"].cuda(), b["qmask"].cuda(), b["umask"].cuda(), b["lengths"], b["acouf"].cuda(), b["visuf"].cuda())[0]
    loss_f(lp, T.flatten_labels(b["label"].cuda(), b["lengths"])).backward()
    o2.bucket.flatten()                                     # fixes the flat layout
    o2.load_state_dict(sd_opt)
    assert o2.param_groups[0]["lr"] == 1e-3
    o2.param_groups[0]["lr"] = 5e-4
    step(m2, o2)
    for (k, a), (_, c) in zip(m1.named_parameters(), m2.named_parameters()):
        assert float((a - c).abs().max()) < 1e-7, k


def test_prefetcher_stages_straight_into_captured_static_buffers(tmp_path):
    """Pass loop + StepGraphCache + DevicePrefetcher: from the second pass on every batch is copied from pinned host memory
    directly into its captured step's static input buffers (the step receives those very tensors), deferred metrics survive
    a signature that repeats inside a pass, and the results equal the eager pass loop's."""
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=4, n_test=18, max_len=24, seed=8)
    names = ['hap', 'sad', 'neu', 'ang', 'exc', 'fru']
    _, _, te = D.get_IEMOCAP_loaders(p, batch_size=3, valid_rate=0.0)
    batches = [list(b) for b in te]
    batches = batches + [batches[1]]                 # one signature twice in a pass: its outputs must be copied out in time
    m = _model(19)
    loss_f = FocalLoss(gamma=0.5)
    want = T.train_or_eval_graph_model(m, loss_f, batches, cuda_flag=True, target_names=names)
    cache = T.StepGraphCache(m, loss_f)
    pre = D.DevicePrefetcher(batches, depth=2)
    first = T.train_or_eval_graph_model(m, loss_f, pre, target_names=names, graph_cache=cache)
    assert cache.misses == len(batches) - 1 and cache.hits == 1
    claimed = []
    orig = cache.claim_static
    cache.claim_static = lambda *a, **k: claimed.append(orig(*a, **k)) or claimed[-1]
    second = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(batches, depth=2), target_names=names, graph_cache=cache)
    # every batch found its captured entry; the repeated signature was staged the ordinary way while its buffers were promised
    assert len(claimed) == len(batches) and sum(c is not None for c in claimed) >= len(batches) - 1
    for got in (first, second):
        assert got[3] == want[3] and got[6] == want[6] and np.array_equal(got[5], want[5])
        assert abs(got[2] - want[2]) < 1e-3


def test_prefetcher_with_consecutive_identical_signatures(tmp_path):
    """several batches of ONE signature inside the prefetch window (uniform-length data, batch_size 1,
    repeated length tuples).  The static buffers of a captured step are handed to the loader only while no earlier batch of
    that signature is still waiting to be stepped; every batch must be replayed with ITS OWN data (different features and
    labels per batch, so a batch replayed under another one's name changes predictions and loss)."""
    p = D.write_synthetic_pickle(str(tmp_path / "u.pkl"), n_train=2, n_test=12, max_len=9, min_len=9, seed=5)
    names = ['hap', 'sad', 'neu', 'ang', 'exc', 'fru']
    _, _, te = D.get_IEMOCAP_loaders(p, batch_size=2, valid_rate=0.0)
    batches = [list(b) for b in te]                  # six batches, every dialogue 9 utterances long: one signature
    assert len(batches) == 6
    m = _model(23)
    loss_f = FocalLoss(gamma=0.5)
    want = T.train_or_eval_graph_model(m, loss_f, batches, cuda_flag=True, target_names=names)
    for depth in (2, 3):
        cache = T.StepGraphCache(m, loss_f)
        for _ in range(3):                           # pass 1 captures, passes 2-3 replay with claims in flight
            got = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(batches, depth=depth), target_names=names,
                                              graph_cache=cache)
            assert np.array_equal(got[5], want[5]) and np.array_equal(got[4], want[4])
            assert np.allclose(got[7][1], want[7][1], atol=1e-4)          # per-batch losses, in order
        assert cache.misses == 1 and not cache.queued


def test_dropout_stream_is_not_rewound_by_captured_steps():
    """building a captured step must consume no random numbers and replays must move torch's generator along,
    so that two signatures captured and stepped back to back (every step of a first epoch is a cache miss) draw their keep
    flags from DISJOINT counter ranges, and torch.manual_seed restarts the stream for replays as it does for eager steps."""
    from mm_dfn_amd import ops
    m = _model(17, dropout=0.5).train()
    loss_f = FocalLoss(gamma=0.5)
    cache = T.StepGraphCache(m, loss_f)
    idx = torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]

    def state():
        torch.cuda.synchronize()
        dev = ops._FLAG_STATE[idx][0].cpu().tolist()
        return dev[1], int(gen.get_offset())

    def run(seed, lengths):
        b = synthetic.make_batch(seed, lengths=list(lengths), device="cuda", B=len(lengths), L=max(lengths), **CFG)
        before = state() if idx in ops._FLAG_STATE else None
        loss, logp, _ = cache.step((b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]), list(lengths), True)
        return before, state(), logp.detach().clone()

    torch.manual_seed(123)
    _, s0, _ = run(1, (9, 4))                       # the very first draw on the device + first capture
    assert s0[0] == s0[1] and s0[0] > 0             # device offset == torch offset, one step's worth consumed
    b1, s1, lp_a = run(2, (11, 5, 3))               # a second signature: cache miss, warm-up + capture + replay
    assert b1 == s0                                 # (nothing moved in between)
    assert s1[0] == s1[1] and s1[0] > s0[0]         # continues behind the first step: not rewound to the capture's start
    b2, s2, lp_a2 = run(2, (11, 5, 3))              # replay of the second signature
    assert s2[0] == s2[1] and s2[0] - s1[0] == s1[0] - s0[0]      # every step of a signature consumes the same amount
    assert float((lp_a - lp_a2).abs().max()) > 1e-6              # different masks on the same inputs
    # torch.manual_seed restarts the stream for replays too (the reference re-seeds every pass, run_train_erc.py:164)
    torch.manual_seed(123)
    _, s3, _ = run(1, (9, 4))
    assert s3 == s0
    _, s4, lp_b = run(2, (11, 5, 3))
    assert s4 == s1 and torch.equal(lp_b, lp_a)


def test_captured_step_pins_cached_index_tensors_and_detects_moved_parameters():
    from mm_dfn_amd import layout, dialogue_model
    from mm_dfn_amd.graphs import CapturedStep
    from mm_dfn_amd.optim import FlatAdam
    m = _model(13).train()
    cfg = dict(B=2, L=9, **CFG)
    b = synthetic.make_batch(3, lengths=[9, 4], device="cuda", **cfg)
    label = T.flatten_labels(b["label"], b["lengths"])
    loss_f = FocalLoss(gamma=0.5)

    def fwd_bwd():
        lp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
        loss = loss_f(lp, label)
        loss.backward()
        return loss

    cap = CapturedStep(m, fwd_bwd, warmup=1)
    want = float(cap.replay())
    g0 = m.smax_fc.weight.grad.clone()
    assert any(isinstance(x, layout.DialogueLayout) for x in cap._pinned) and any(torch.is_tensor(x) for x in cap._pinned)
    # flood both caches with other batch signatures: the captured step's index tensors are evicted from the caches ...
    for n in range(80):
        layout.DialogueLayout.get([n + 1, 2], 3, "cuda")
        dialogue_model._flat_index([n + 1, 2], n + 1, 2, torch.device("cuda"))
    junk = [torch.full((4096,), -7, dtype=torch.int64, device="cuda") for _ in range(64)]   # ... and memory is re-used
    assert all(x not in layout._LAYOUT_CACHE.data.values() for x in cap._pinned if isinstance(x, layout.DialogueLayout))
    assert float(cap.replay()) == want and torch.equal(m.smax_fc.weight.grad, g0)
    del junk
    # re-pointing the parameter storages after the capture is caught instead of silently training stale memory
    opt = FlatAdam(m, lr=1e-3)
    opt.bucket.flatten()
    opt._materialise()
    with pytest.raises(RuntimeError, match="storage moved"):
        cap.replay()


def test_pass_loop_with_captured_step_cache_equals_eager(tmp_path):
    """train_or_eval_graph_model(graph_cache=StepGraphCache): ragged batches of changing signature, two epochs (the
    second one is all replays), training trajectory and metrics identical to the eager loop (dropout 0)."""
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=22, n_test=9, max_len=30, seed=8)
    names = ['hap', 'sad', 'neu', 'ang', 'exc', 'fru']
    loss_f = FocalLoss(gamma=0.5)
    runs = []
    for use_cache in (False, True):
        m = _model(21)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
        tr, _, te = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0)
        cache = T.StepGraphCache(m, loss_f) if use_cache else None
        hist = []
        for e in range(2):
            r_tr = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(tr), e, True, opt, False, 'avl', names,
                                               graph_cache=cache)
            r_te = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(te), e, False, None, False, 'avl', names,
                                               graph_cache=cache)
            hist.append((r_tr[2], r_tr[3], r_te[2], r_te[3], r_te[5].copy()))
        runs.append((m, hist, cache))
    (m0, h0, _), (m1, h1, cache) = runs
    for a, b in zip(h0, h1):
        assert abs(a[0] - b[0]) < 2e-4 and abs(a[2] - b[2]) < 2e-4 and a[1] == b[1] and a[3] == b[3]
        assert (a[4] == b[4]).mean() > 0.99
    for (k, x), (_, y) in zip(m0.named_parameters(), m1.named_parameters()):
        assert float((x - y).abs().max()) < 2e-5, k
    n_tr, n_te = 6, 3                                   # ceil(22/4), ceil(9/4) batches per pass
    assert cache.misses == n_tr + n_te                  # the per-pass reseed repeats the shuffle: epoch 2 only replays
    assert cache.hits == n_tr + n_te


def test_captured_step_cache_with_flat_adam_recaptures_after_the_flat_layout(tmp_path):
    """StepGraphCache + FlatAdam: the optimizer re-points every parameter into its flat buffer at its first step, which
    makes the entries captured before that stale; the cache notices (CapturedStep checks the storages), captures those
    signatures again and training continues on the same trajectory as torch.optim.Adam in the eager loop."""
    from mm_dfn_amd.optim import FlatAdam
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=14, n_test=5, max_len=24, seed=9)
    loss_f = FocalLoss(gamma=0.5)
    runs = []
    for flat in (False, True):
        m = _model(22)
        opt = FlatAdam(m, lr=1e-3, weight_decay=1e-5) if flat else torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
        tr, _, _ = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0)
        cache = T.StepGraphCache(m, loss_f) if flat else None
        losses = []
        for e in range(3):
            r = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(tr), e, True, opt, False, 'avl', None, graph_cache=cache)
            losses.append(r[2])
        runs.append((m, losses, cache))
    (m0, l0, _), (m1, l1, cache) = runs
    assert all(abs(a - b) < 3e-4 for a, b in zip(l0, l1)), (l0, l1)
    # 12 Adam steps of lr 1e-3 move a weight by up to 1.2e-2.  The two runs differ in the optimizer arithmetic (torch's multi-tensor
    # Adam vs the fused flat kernel), and Adam normalises every gradient: an entry whose gradient is a rounding error away from
    # zero can move by a different +-lr per step in the two runs, so the maximum over a tensor is a lottery (5e-5 .. 1.3e-4 seen
    # with different summation orders of the weight-gradient kernel) while the mean deviation stays at 1e-9 .. 2e-6.  Checked:
    # the mean deviation (tight) and the worst entry against the distance one step can move it.
    for (k, x), (_, y) in zip(m0.named_parameters(), m1.named_parameters()):
        d = (x - y).abs()
        assert float(d.mean()) < 1e-5, k
        assert float(d.max()) < 1e-3, k
    assert cache.recaptures >= 1                          # the entry captured before the first optimizer step
    assert cache.hits >= 2 * 4 - 1                         # epochs 2 and 3 replay


def _run_rccl_case(name):
    """The RCCL cases run in a process of their own, and the child's EXIT CODE counts (round 6), not only its marker: every case
    releases its captured graphs (CapturedStep.close()) and then leaves through destroy_process_group(), so an abort anywhere --
    the teardown included -- fails the test instead of being hidden behind a marker printed earlier.
    Why a child at all (round-6 findings, tools/rccl_teardown_repro.py): init -> capture-with-all-reduce -> replay ->
    destroy survives 80 / 80 rounds in a fresh process whatever the order of graph release and group teardown (graphs kept alive
    included), so the teardown order was NOT what aborted in round 5.  Inside the long pytest process the same cases abort about
    once in three runs, and the dump shows a NATIVE thread (c10d watchdog / RCCL service thread) aborting while the main thread is
    inside dist.all_reduce under stream capture (graphs.py tail(), not the teardown) -- a race between the communication
    library's helper threads and a capture in a process that has already created and destroyed groups.  It takes the whole pytest
    session with it, so the cases keep their own process; MMDFN_RCCL_INPROC=1 runs them in-process for hunting."""
    import os
    if os.environ.get("MMDFN_RCCL_INPROC", "0") == "1":
        globals()[name]()
        return
    import subprocess
    import sys
    tests = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(tests)
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import test_trainer_gpu as t; t.%s()"
            % (root, os.path.join(root, "oracle"), tests, name))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "RCCL-CASE-OK" in out.stdout, \
        "rc %s\n%s\n%s" % (out.returncode, out.stdout[-2000:], out.stderr[-6000:])


def _init_rccl_group():
    """init_process_group("nccl") at world size 1 on a free local port (a port handed out by bind(0) can be taken again by the
    time the store listens on it: retried)."""
    import os
    import socket
    import torch.distributed as dist
    last = None
    for _ in range(5):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        try:
            dist.init_process_group(backend="nccl")
            return
        except dist.DistNetworkError as exc:
            last = exc
    raise last


def _rccl_teardown(cap):
    """Graphs first, then the group: a captured collective keeps RCCL work alive for as long as its hipGraphExec exists."""
    import gc
    import torch.distributed as dist
    if cap is not None:
        cap.close()
    gc.collect()
    torch.cuda.synchronize()
    dist.destroy_process_group()


def test_real_model_gradient_bucket_through_rccl_and_flat_adam():
    """The data-parallel machinery on the real model and device tensors, world size 1 over RCCL (backend "nccl"):
    captured step -> flat bucket pack -> all-reduce (also as a node of the captured graph) -> FlatAdam, against
    the plain eager step with torch Adam."""
    _run_rccl_case("_rccl_case_bucket_and_flat_adam")


def _rccl_case_bucket_and_flat_adam():
    import os
    import socket
    import torch.distributed as dist
    from mm_dfn_amd import distributed
    from mm_dfn_amd.graphs import CapturedStep
    from mm_dfn_amd.optim import FlatAdam
    _init_rccl_group()
    cap = None
    try:
        cfg = dict(B=3, L=12, **CFG)
        b = synthetic.make_batch(31, lengths=[12, 5, 9], device="cuda", **cfg)
        label = T.flatten_labels(b["label"], b["lengths"])
        loss_f = FocalLoss(gamma=0.5)
        m_ref, m_dp = _model(33).train(), _model(33).train()
        o_ref = torch.optim.Adam(m_ref.parameters(), lr=1e-3, weight_decay=1e-4)

        def fb(m):
            lp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
            loss = loss_f(lp, label)
            loss.backward()
            return loss

        bucket = distributed.GradientBucket(m_dp, average=True)
        m_dp.zero_grad(set_to_none=True)
        fb(m_dp)
        opt = FlatAdam(m_dp, lr=1e-3, weight_decay=1e-4, bucket=bucket)
        bucket.flatten()
        opt._materialise()
        live = [p for p in m_dp.parameters() if p.grad is not None]
        assert bucket.flat.numel() == sum(distributed.slot_size(p) for p in live) and len(live) == 48   # 16-byte aligned slots
        try:
            cap = CapturedStep(m_dp, lambda: fb(m_dp), warmup=1, bucket=bucket, reduce_in_graph=True)
            in_graph = True
        except Exception:                                   # a runtime without collective capture: eager all-reduce
            torch.cuda.synchronize()
            cap = CapturedStep(m_dp, lambda: fb(m_dp), warmup=1, bucket=bucket)
            in_graph = False
        for _ in range(3):
            o_ref.zero_grad(set_to_none=True)
            l_ref = fb(m_ref)
            o_ref.step()
            l_dp = cap.replay()
            if not in_graph:
                bucket.reduce_flat()
            opt.step(grads_already_flat=True)
            assert abs(float(l_ref) - float(l_dp)) < 5e-6
        for (k, x), (_, y) in zip(m_ref.named_parameters(), m_dp.named_parameters()):
            assert float((x - y).abs().max()) < 5e-6, k
        assert dist.get_backend() == "nccl"
        torch.cuda.synchronize()
        print("RCCL-CASE-OK", flush=True)
    finally:
        _rccl_teardown(cap)


def _step_inputs(seed=11, lengths=(20, 13, 7)):
    b = synthetic.make_batch(seed, lengths=list(lengths), device="cuda", B=len(lengths), L=max(lengths), **CFG)
    return b, T.flatten_labels(b["label"], b["lengths"])


def _loss(m, b, flat):
    logp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
    return FocalLoss(gamma=0.5)(logp, flat)


def test_weight_gradient_batch_is_opt_in_and_matches_plain_autograd():
    """The one-launch weight-gradient batch runs only under ops.wgrad_batch() (train.backward): a plain loss.backward()
    and torch.autograd.grad() get every parameter gradient from autograd itself, with the same values, and
    autograd.grad() writes no .grad behind the caller's back."""
    from mm_dfn_amd import ops
    m = _model().train()
    b, flat = _step_inputs()
    T.backward(_loss(m, b, flat))                       # batched: .grad written by the end-of-backward launch pair
    live = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    assert "graph_model.graph_net.rnn.weight_ih_l0" in live and "lstm_l.weight_hh_l0" in live
    m.zero_grad(set_to_none=True)
    _loss(m, b, flat).backward()                        # plain autograd: in-line weight gradients
    for n, p in m.named_parameters():
        assert (p.grad is not None) == (n in live), n
        if n in live:
            scale = float(live[n].abs().max()) + 1e-12
            assert float((p.grad - live[n]).abs().max()) / scale < 1e-5, n
    m.zero_grad(set_to_none=True)
    params = [p for n, p in m.named_parameters() if n in live]
    grads = torch.autograd.grad(_loss(m, b, flat), params)
    assert all(g is not None for g in grads)
    assert all(p.grad is None for p in m.parameters())  # nothing written on the side
    for g, (n, p) in zip(grads, [(n, p) for n, p in m.named_parameters() if n in live]):
        assert float((g - live[n]).abs().max()) / (float(live[n].abs().max()) + 1e-12) < 1e-5, n
    # input-only autograd.grad: no parameter gradient is computed or stored
    x = b["textf"].clone().requires_grad_(True)
    logp = m(x, b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
    (gx,) = torch.autograd.grad(logp.sum(), [x])
    assert gx is not None and all(p.grad is None for p in m.parameters())
    assert not ops._WGQ["outs"] and not ops._WGQ["armed"]


def test_slab_stacks_of_the_head_and_the_gather_bias_ride_on_the_batch_reduction():
    """Round 5: the classifier's dW / db slabs (mmdfn_head_bwd_partial) and the column-sum slabs of the project-then-gather
    node's bias gradient (mmdfn_colsum_partial) are summed by the reduction launch of the step's weight-gradient batch
    (mmdfn_gemm_tn_batch_ext): no head_reduce / colsum_final launch in the trace, same gradients as plain autograd (checked by
    the test above), and a second backward without zero_grad accumulates (the bias halves then take the autograd path)."""
    from torch.profiler import ProfilerActivity, profile
    from mm_dfn_amd import ops, ops_wgrad  # (the queue's own module: a name that is REBOUND has to be patched where it is looked up)
    m = _model().train()
    b, flat = _step_inputs()
    seen = []
    orig = ops_wgrad._prepare_wgrad_batch

    def spy(batch, ext_items=None, **kw):
        seen.append((len(batch), len(ext_items or [])))
        return orig(batch, ext_items, **kw)
    ops_wgrad._prepare_wgrad_batch = spy
    try:
        torch.manual_seed(3)
        T.backward(_loss(m, b, flat))
    finally:
        ops_wgrad._prepare_wgrad_batch = orig
    assert seen and sum(e for _, e in seen) == 2 and seen[-1][0] > 0, seen        # both stacks ride on the last batch
    g1 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    assert "smax_fc.weight" in g1 and "smax_fc.bias" in g1
    torch.manual_seed(3)
    T.backward(_loss(m, b, flat))                       # no zero_grad: every gradient doubles
    for n, p in m.named_parameters():
        if n in g1:
            scale = float(g1[n].abs().max()) + 1e-12
            assert float((p.grad - 2 * g1[n]).abs().max()) / scale < 2e-5, n
    m.zero_grad(set_to_none=True)
    T.backward(_loss(m, b, flat))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        m.zero_grad(set_to_none=True)
        T.backward(_loss(m, b, flat))
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("head_bwd_kernel" in n for n in names) and any("gemm_tn_batch_reduce" in n for n in names), names
    assert not [n for n in names if "head_reduce" in n or "colsum_final" in n], names
    assert not ops._WGQ["outs"] and not ops._WGQ["ext"] and not ops._WGQ["armed"]


def test_parameter_hooks_fire_under_the_weight_gradient_batch():
    m = _model().train()
    b, flat = _step_inputs()
    seen = []
    w = m.graph_model.graph_net.convs[0].weight
    h = w.register_hook(lambda g: seen.append(tuple(g.shape)))
    T.backward(_loss(m, b, flat))
    h.remove()
    assert seen == [tuple(w.shape)] and w.grad is not None
    assert m.lstm_l.weight_hh_l0.grad is not None       # un-hooked parameters still went through the batch


def test_two_part_bucket_on_the_real_model_starts_its_first_collective_inside_backward():
    """GradientBucket(parts=2) on the real model through RCCL (world size 1): from the second step on the graph / head
    half is packed and reduced where the adjacency builder's backward ends (the hook fires inside loss.backward), the
    gradients equal the one-part bucket's, eager and as nodes of a captured step."""
    _run_rccl_case("_rccl_case_two_part_bucket")


def _rccl_case_two_part_bucket():
    import os
    import socket
    import torch.distributed as dist
    from mm_dfn_amd import distributed
    from mm_dfn_amd.graphs import CapturedStep
    _init_rccl_group()
    cap = None
    try:
        b, flat = _step_inputs(lengths=(20, 13, 7))
        grads = {}
        for parts in (1, 2):
            m = _model(13).train()
            bucket = distributed.GradientBucket(m, average=True, parts=parts)
            early = []
            for _ in range(3):
                m.zero_grad(set_to_none=True)
                bucket.arm()
                T.backward(_loss(m, b, flat))
                early.append(bucket._early == "packed")
                bucket.all_reduce()
            assert early == ([False, True, True] if parts == 2 else [False] * 3)
            grads[parts] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
            if parts == 2:
                names = {id(p): n for n, p in m.named_parameters()}
                order = [names[id(p)] for p in bucket.params]
                ne = sum(n.startswith(distributed.EARLY_PREFIXES) for n in order)
                assert 0 < ne < len(order) and all(n.startswith(distributed.EARLY_PREFIXES) for n in order[:ne])
                assert "lstm_l.weight_hh_l0" in order[ne:] and "graph_model.graph_net.rnn.weight_ih_l0" in order[:ne]

                def fwd_bwd():
                    bucket.arm()
                    loss = _loss(m, b, flat)
                    T.backward(loss)
                    return loss
                cap = CapturedStep(m, fwd_bwd, warmup=1, bucket=bucket, reduce_in_graph=True)
                cap.replay()
                torch.cuda.synchronize()
                for n, g in grads[2].items():
                    assert float((dict(m.named_parameters())[n].grad - g).abs().max()) <= 1e-6 * float(g.abs().max() + 1e-30), n
        for n, g in grads[1].items():
            assert torch.equal(g, grads[2][n]), n
        torch.cuda.synchronize()
        print("RCCL-CASE-OK", flush=True)
    finally:
        _rccl_teardown(cap)


def test_rccl_teardown_thirty_rounds_in_process():
    """thirty times in ONE process -- init_process_group("nccl"), a captured step whose graph holds the
    gradient all-reduce, replays, CapturedStep.close(), destroy_process_group() -- without an abort, a hang or a changed loss."""
    _run_rccl_case("_rccl_case_thirty_rounds")


def _rccl_case_thirty_rounds():
    import torch.distributed as dist
    from mm_dfn_amd import distributed
    from mm_dfn_amd.graphs import CapturedStep
    b, flat = _step_inputs(lengths=(12, 5, 9))
    losses = []
    for it in range(30):
        _init_rccl_group()
        cap = None
        try:
            m = _model(13).train()
            bucket = distributed.GradientBucket(m, average=True)
            m.zero_grad(set_to_none=True)
            T.backward(_loss(m, b, flat))
            bucket.flatten()

            def fwd_bwd():
                loss = _loss(m, b, flat)
                T.backward(loss)
                return loss
            cap = CapturedStep(m, fwd_bwd, warmup=1, bucket=bucket, reduce_in_graph=True)
            for _ in range(2):
                loss = cap.replay()
            losses.append(float(loss))
        finally:
            _rccl_teardown(cap)
        with pytest.raises(RuntimeError):
            cap.replay()                                   # closed
        assert not dist.is_initialized()
    assert len(losses) == 30 and max(losses) - min(losses) == 0.0
    print("RCCL-CASE-OK", flush=True)


def test_weight_gradient_queue_survives_a_backward_that_raises():
    """A backward pass that raises leaves queued segments behind (the engine never runs the end-of-backward callback);
    the next step must not inherit them (every queued parameter silently lost its .grad for the rest of
    the process)."""
    from mm_dfn_amd import ops

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    m = _model().train()
    b, flat = _step_inputs()
    good = _model().train()
    T.backward(_loss(good, b, flat))
    want = {n: p.grad.clone() for n, p in good.named_parameters() if p.grad is not None}
    # the failing pass: the head and the graph stack queue their segments, then the encoder side raises
    x = b["textf"].clone().requires_grad_(True)
    logp = m(Boom.apply(x), b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
    with pytest.raises(RuntimeError, match="boom"):
        T.backward(FocalLoss(gamma=0.5)(logp, flat))
    assert not ops._WGQ["outs"] and not ops._WGQ["armed"] and ops._WGQ["scope"] == 0
    m.zero_grad(set_to_none=True)
    T.backward(_loss(m, b, flat))
    for n, p in m.named_parameters():
        assert (p.grad is not None) == (n in want), n
        if n in want:
            assert float((p.grad - want[n]).abs().max()) / (float(want[n].abs().max()) + 1e-12) < 1e-5, n


REF_DIMS = dict(P=2, C=6, nlayers=2, D_t=100, D_a=1582, D_v=342)        # the reference's IEMOCAP features (run_train_erc.py:359-362)
CFG3 = dict(P=9, C=7, nlayers=4, D_t=600, D_a=300, D_v=342)            # BASELINE cfg3 (MELD-like)
RAGGED_CFG3 = (33, 3, 17, 31, 9, 27, 12, 33, 5, 21, 30, 8, 16, 2, 25, 11, 33, 7, 19, 29, 4, 14, 23, 10, 32, 6, 18, 28, 13, 22, 1, 26)


@pytest.mark.parametrize("lengths,cfg", [((20, 13, 7), CFG), (tuple([110] * 16), CFG), ((20, 13, 7), REF_DIMS),
                                         (tuple([110] * 16), REF_DIMS), (RAGGED_CFG3, CFG3)],
                         ids=["small", "cfg2", "small-1582-342", "cfg2-1582-342", "cfg3"])
def test_training_step_launches_no_library_gemm(lengths, cfg):
    """Every dense product of a training step (forward, input gradients, weight gradients) runs on this package's MFMA
    kernels: no Tensile (`Cijk_*`: hipBLASLt / rocBLAS) kernel and no ATen matmul in the device trace -- for a few-row batch,
    the BASELINE cfg2 shape (16 x 110: the context GRU's 1 760-row and the party GRU's 7 040-row products),
    the reference's own feature widths (1582-d audio, 342-d visual: contraction widths that are not multiples of 4 run on
    row-padded operands, ops.py) and BASELINE cfg3."""
    from torch.profiler import ProfilerActivity, profile
    m = synthetic.build_model(dropout=0.1, **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 5))
    m = m.cuda().train()
    b = synthetic.make_batch(11, lengths=list(lengths), device="cuda", B=len(lengths), L=max(lengths), **cfg)
    flat = T.flatten_labels(b["label"], b["lengths"])

    def step():
        m.zero_grad(set_to_none=True)
        T.backward(_loss(m, b, flat))
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("linear_lds_kernel" in n for n in names) and any("gru_seq" in n for n in names), names
    assert not [n for n in names if n.startswith("Cijk_") or ("gemm" in n.lower() and "gemm_tn" not in n)], names


@pytest.mark.parametrize("padded_inputs", [True, False], ids=["staged-padded", "plain-contiguous"])
def test_odd_feature_widths_forward_and_gradients_against_oracle(padded_inputs):
    """The reference's IEMOCAP feature widths (1582 / 342: neither a multiple of 4) through the row-padded path: log-probs and
    ALL live parameter gradients against the CPU oracle, with features staged row-padded (the data pipeline's form) and as
    plain contiguous tensors (copied into a padded buffer by the module); state_dict round trip keeps the reference's shapes;
    the weight-gradient batch and plain autograd agree."""
    import mmdfn_oracle as O
    from util import rel_err
    lengths = [23, 9, 17]
    m = synthetic.build_model(dropout=0.0, **REF_DIMS)
    sd = synthetic.seeded_state_dict(m.state_dict(), 3)
    m.load_state_dict(sd)
    m = m.cuda().train()
    b = synthetic.make_batch(21, lengths=lengths, device="cuda", B=3, L=23, **REF_DIMS)
    if not padded_inputs:
        b["acouf"], b["visuf"] = b["acouf"].contiguous(), b["visuf"].contiguous()
    flat = T.flatten_labels(b["label"], b["lengths"])
    logp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
    T.backward(FocalLoss(gamma=0.5)(logp, flat))
    assert tuple(m.linear_a.weight.shape) == (200, 1582) and tuple(m.linear_a.weight.grad.shape) == (200, 1582)
    assert m.linear_a.weight.stride(0) == 1584 and m.linear_v.weight.stride(0) == 344       # the padded storage
    got = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    # oracle (CPU, reference op structure; aten engine: autograd through torch's own GRU)
    cpu = {k: v.cpu() for k, v in b.items() if torch.is_tensor(v)}
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want_logp = O.forward(P, cpu["textf"], cpu["qmask"], cpu["umask"], lengths, cpu["acouf"], cpu["visuf"],
                          O.default_cfg(nlayers=2), engine="aten")
    assert float((logp.detach().cpu() - want_logp).abs().max()) < 1e-4
    O.focal_loss(want_logp, O.flatten_labels(cpu["label"], lengths), gamma=0.5).backward()
    for n in ("linear_a.weight", "linear_v.weight", "linear_l.weight", "linear_a.bias", "lstm_l.weight_ih_l0",
              "graph_model.graph_net.convs.0.weight", "smax_fc.weight"):
        assert rel_err(got[n], P[n].grad) < 1e-4, n
    # plain autograd (no batch): same gradients
    m.zero_grad(set_to_none=True)
    FocalLoss(gamma=0.5)(m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0], flat).backward()
    for n in ("linear_a.weight", "linear_v.weight"):
        assert rel_err(dict(m.named_parameters())[n].grad.cpu(), got[n]) < 1e-5, n
    # the state_dict keeps the reference's shapes and loads back
    out = m.state_dict()
    assert tuple(out["linear_a.weight"].shape) == (200, 1582)
    m2 = synthetic.build_model(dropout=0.0, **REF_DIMS)
    m2.load_state_dict({k: v.cpu() for k, v in out.items()})
    assert torch.equal(m2.linear_a.weight.detach(), sd["linear_a.weight"])


def test_odd_feature_widths_training_trajectory_flat_adam_vs_torch_adam():
    """Three optimizer steps at the reference's feature widths: FlatAdam (row-padded slots in the flat buffers) follows
    torch.optim.Adam on the padded parameters, through the captured-step cache."""
    from mm_dfn_amd.optim import FlatAdam
    lengths = [23, 9, 17]
    sd = None
    traj = {}
    for kind in ("torch", "flat"):
        m = synthetic.build_model(dropout=0.0, **REF_DIMS)
        sd = sd or synthetic.seeded_state_dict(m.state_dict(), 3)
        m.load_state_dict(sd)
        m = m.cuda().train()
        opt = (torch.optim.Adam(m.parameters(), lr=3e-4, weight_decay=1e-4) if kind == "torch"
               else FlatAdam(m, lr=3e-4, weight_decay=1e-4))
        cache = T.StepGraphCache(m, FocalLoss(gamma=0.5))
        losses = []
        for step in range(3):
            b = synthetic.make_batch(21 + step, lengths=lengths, device="cuda", B=3, L=23, **REF_DIMS)
            loss, _, _ = cache.step((b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]), lengths, True)
            losses.append(float(loss))
            opt.step()
        traj[kind] = (losses, m.linear_a.weight.detach().cpu().clone(), m.linear_v.weight.detach().cpu().clone())
    for a, c in zip(traj["torch"][0], traj["flat"][0]):
        assert abs(a - c) < 2e-4 * max(1.0, abs(a))
    for i in (1, 2):
        assert float((traj["torch"][i] - traj["flat"][i]).abs().max()) < 5e-5
        assert float((traj["torch"][i] - sd["linear_a.weight" if i == 1 else "linear_v.weight"]).abs().max()) > 1e-5   # it moved


def test_thirty_step_trajectory_flat_adam_vs_torch_adam_at_the_reference_widths():
    """the captured step + FlatAdam + row-padded parameters against torch.optim.Adam over THIRTY steps (the
    3-step tests above cannot see a slow drift): losses step by step, and every parameter at the end, on six different
    ragged batches cycled through the captured-step cache (so every signature is replayed several times) with the party
    encoder on the valid-length launches.  Adam divides by sqrt(v) + eps: a gradient difference at fp32 rounding level moves
    a parameter whose second moment is tiny by a visible amount, so the bound on the parameters is relative to the distance
    they travelled."""
    from mm_dfn_amd.optim import FlatAdam
    from mm_dfn_amd import gru as fused
    lens = [[23, 9, 17], [19, 19, 4], [23, 2, 11], [8, 23, 15], [23, 23, 23], [5, 6, 7]]
    sd = None
    traj = {}
    prev, fused.TRUNCATE = fused.TRUNCATE, True
    try:
        for kind in ("torch", "flat"):
            m = synthetic.build_model(dropout=0.0, **REF_DIMS)
            sd = sd or synthetic.seeded_state_dict(m.state_dict(), 3)
            m.load_state_dict(sd)
            m = m.cuda().train()
            opt = (torch.optim.Adam(m.parameters(), lr=3e-4, weight_decay=1e-4) if kind == "torch"
                   else FlatAdam(m, lr=3e-4, weight_decay=1e-4))
            cache = T.StepGraphCache(m, FocalLoss(gamma=0.5))
            losses = []
            for step in range(30):
                lengths = lens[step % len(lens)]
                b = synthetic.make_batch(40 + step % len(lens), lengths=lengths, device="cuda", B=3, L=23, **REF_DIMS)
                loss, _, _ = cache.step((b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]), lengths, True)
                losses.append(float(loss))
                opt.step()
            assert cache.hits >= 20
            traj[kind] = (losses, {k: v.detach().cpu().clone() for k, v in m.named_parameters()})
    finally:
        fused.TRUNCATE = prev
    for a, c in zip(traj["torch"][0], traj["flat"][0]):
        assert abs(a - c) < 5e-4 * max(1.0, abs(a))
    assert traj["torch"][0][-1] < traj["torch"][0][0]                    # it trains
    worst = 0.0
    for k, a in traj["torch"][1].items():
        c = traj["flat"][1][k]
        moved = float((a - sd[k]).abs().max())
        if moved == 0.0:
            assert float((c - sd[k]).abs().max()) == 0.0, k              # parameters the path never reaches stay put in both
            continue
        worst = max(worst, float((a - c).abs().max()) / moved)
    assert worst < 0.05, worst


def _eager_step(m, loss_f, b):
    m.zero_grad(set_to_none=True)
    logp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
    flat = T.flatten_labels(b["label"], b["lengths"])
    loss = loss_f(logp, flat)
    T.backward(loss)
    return float(loss), logp.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("cfg_name", ["small", "refdims", "meld"])
def test_bucketed_step_cache_serves_unseen_length_tuples(cfg_name):
    """StepGraphCache(bucket_rows=g): one captured step per (B, L, utterance bucket); batches whose length TUPLES were never
    seen replay it -- padded to the bucket with a dummy dialogue, index arrays rewritten before the replay -- and give the
    eager step's log-probabilities (bit for bit: every row's arithmetic is that of its own dialogue), loss and gradients
    (to summation-order noise: the padding rows add exact zeros) on the batch alone."""
    cfgs = {"small": (CFG, 2, 8), "refdims": (REF_DIMS, 2, 8), "meld": (dict(P=9, C=7, nlayers=3, D_t=100, D_a=100, D_v=512), 9, 16)}
    cfg, P, g = cfgs[cfg_name]
    L, B = 23, 3
    m = synthetic.build_model(dropout=0.0, **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 31))
    m = m.cuda().train()
    loss_f = FocalLoss(gamma=0.5)
    cache = T.StepGraphCache(m, loss_f, bucket_rows=g, larger_bucket_fallback=False)    # (every bucket gets its own entry)
    tuples = [[23, 9, 17], [23, 17, 9], [19, 23, 4], [23, 2, 21], [8, 15, 23], [23, 23, 1], [23, 12, 12], [23, 11, 15], [5, 23, 6]]
    keys = set()
    for i, lengths in enumerate(tuples):
        b = synthetic.make_batch(60 + i, lengths=lengths, device="cuda", B=B, L=L, **cfg)
        want_loss, want_logp, want_g = _eager_step(m, loss_f, b)
        loss, logp, flat = cache.step((b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]), lengths, True)
        N = sum(lengths)
        assert tuple(logp.shape) == (N, cfg["C"]) and tuple(flat.shape) == (N,)
        assert torch.equal(flat, T.flatten_labels(b["label"], lengths))
        assert torch.equal(logp, want_logp), float((logp - want_logp).abs().max())
        assert abs(float(loss) - want_loss) < 2e-6 * max(1.0, abs(want_loss))
        got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        assert set(got) == set(want_g)
        for k in want_g:
            den = float(want_g[k].abs().max())
            assert float((got[k] - want_g[k]).abs().max()) <= 2e-5 * den + 1e-9, (k, lengths)
        assert torch.equal(cache.last_pred, torch.argmax(want_logp, 1))
        keys.add(((N + 1 + g - 1) // g) * g)
    assert cache.misses == len(keys) < len(tuples) and cache.hits == len(tuples) - len(keys)
    # eval entries are separate
    b = synthetic.make_batch(90, lengths=[23, 9, 17], device="cuda", B=B, L=L, **cfg)
    m.eval()
    with torch.no_grad():
        want = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
    loss, logp, _ = cache.step((b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]), [23, 9, 17], False)
    assert torch.equal(logp, want)
    m.train()
    # a batch padded beyond its own maximum length (the reference's collate never does that) takes the exact-signature path
    b2 = synthetic.make_batch(91, lengths=[20, 9, 17], device="cuda", B=B, L=L, **cfg)
    padL = lambda t, dim: torch.cat([t, torch.zeros_like(t.narrow(dim, 0, 3))], dim)
    inp = (padL(b2["textf"], 0), padL(b2["visuf"], 0), padL(b2["acouf"], 0), padL(b2["qmask"], 0), padL(b2["umask"], 1),
           padL(b2["label"], 1))
    before = cache.misses
    loss, logp, _ = cache.step(inp, [20, 9, 17], True)
    assert cache.misses == before + 1 and sum(1 for k in cache.entries if k[0] != "bucket") == 1
    assert tuple(logp.shape) == (46, cfg["C"])


def test_a_batch_without_its_own_bucket_is_served_by_a_larger_captured_one():
    """larger_bucket_fallback (the default): no capture in the middle of a pass when a larger bucket of the same (B, L) is within
    reach of the padding dialogue (<= L utterances); the results are still those of the batch alone."""
    cfg, g, L, B = CFG, 8, 23, 3
    m = synthetic.build_model(dropout=0.0, **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 33))
    m = m.cuda().train()
    loss_f = FocalLoss(gamma=0.5)
    cache = T.StepGraphCache(m, loss_f, bucket_rows=g)
    # N = 63 -> bucket 64 (captured); N = 49 -> bucket 56: served by 64 (15 padding utterances); N = 62 -> 64 itself;
    # N = 30 -> bucket 32: 64 is out of reach (30 + 23 < 64): captured
    plan = [([23, 20, 20], (1, 0, 0)), ([23, 17, 9], (1, 1, 1)), ([23, 19, 20], (1, 2, 1)), ([23, 3, 4], (2, 2, 1)),
            ([23, 9, 17], (2, 3, 2))]
    for i, (lengths, (misses, hits, fallbacks)) in enumerate(plan):
        b = synthetic.make_batch(160 + i, lengths=lengths, device="cuda", B=B, L=L, **cfg)
        want_loss, want_logp, want_g = _eager_step(m, loss_f, b)
        loss, logp, flat = cache.step((b["textf"], b["visuf"], b["acouf"], b["qmask"], b["umask"], b["label"]), lengths, True)
        assert (cache.misses, cache.hits, cache.fallbacks) == (misses, hits, fallbacks), (lengths, cache.misses, cache.hits, cache.fallbacks)
        assert tuple(logp.shape) == (sum(lengths), cfg["C"])
        assert torch.equal(flat, T.flatten_labels(b["label"], lengths))
        assert torch.equal(logp, want_logp), float((logp - want_logp).abs().max())
        assert abs(float(loss) - want_loss) < 2e-6 * max(1.0, abs(want_loss))
        got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        for k in want_g:
            den = float(want_g[k].abs().max())
            assert float((got[k] - want_g[k]).abs().max()) <= 2e-5 * den + 1e-9, (k, lengths)


def test_reshuffled_epochs_stay_on_replays_with_the_bucketed_cache(tmp_path):
    """A loader that reshuffles every epoch (a different seed per pass -- unlike the reference's per-pass reseed,
    run_train_erc.py:164): with exact signatures every batch of every epoch is a new capture; with bucketed entries the
    later epochs are replays.  The evaluation pass gives the eager loop's metrics on every epoch."""
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=40, n_test=4, max_len=26, min_len=9, seed=12)
    names = ['hap', 'sad', 'neu', 'ang', 'exc', 'fru']
    loss_f = FocalLoss(gamma=0.5)
    m = _model(27)
    tr, _, _ = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0)
    cache = T.StepGraphCache(m, loss_f, bucket_rows=16)
    stats = []
    for e in range(6):
        before = (cache.hits, cache.misses)
        want = T.train_or_eval_graph_model(m, loss_f, tr, e, False, None, True, 'avl', names, seed=100 + e)
        got = T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(tr), e, False, None, False, 'avl', names, seed=100 + e,
                                          graph_cache=cache)
        assert got[3] == want[3] and got[6] == want[6] and np.array_equal(got[5], want[5]) and abs(got[2] - want[2]) < 1e-4
        stats.append((cache.hits - before[0], cache.misses - before[1]))
    late_hits = sum(h for h, _ in stats[3:])
    late = sum(h + mi for h, mi in stats[3:])
    assert late_hits >= 0.8 * late, stats


def test_precaptured_buckets_leave_no_capture_to_the_passes(tmp_path):
    """StepGraphCache.precapture (round 6): the bucket set is captured from the loader's length histogram BEFORE the first pass --
    walks of the same loader under the seeds the passes will use (the reference reseeds every pass, run_train_erc.py:164, so its
    shuffles are known in advance), no optimizer, no metrics -- and the training passes that follow capture NOTHING; the model's
    .grad fields are left alone."""
    p = D.write_synthetic_pickle(str(tmp_path / "f.pkl"), n_train=40, n_test=4, max_len=26, min_len=9, seed=12)
    names = ['hap', 'sad', 'neu', 'ang', 'exc', 'fru']
    loss_f = FocalLoss(gamma=0.5)
    m = _model(27).train()
    tr, _, _ = D.get_IEMOCAP_loaders(p, batch_size=4, valid_rate=0.0)
    cache = T.StepGraphCache(m, loss_f, bucket_rows=16)
    marker = torch.ones_like(m.smax_fc.bias)
    m.smax_fc.bias.grad = marker
    made = 0
    for e in range(3):                                   # the warm-up: the passes' own shuffles (seed 100 + e, as below)
        T.seed_everything(100 + e)
        made += cache.precapture(D.DevicePrefetcher(tr), train_flag=True)
    assert made == len(cache.entries) >= 1 and m.smax_fc.bias.grad is marker      # .grad fields untouched
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    for e in range(3):
        before = cache.misses
        T.train_or_eval_graph_model(m, loss_f, D.DevicePrefetcher(tr), e, True, opt, False, 'avl', names, seed=100 + e,
                                    graph_cache=cache)
        assert cache.misses == before, (e, cache.misses - before)      # zero captures after the warm-up
