"""bench.py -- utterances/sec (fwd + loss + bwd) of the MM-DFN hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfgK] [--ragged]

Workload: BASELINE configs[1] (cfg2: 16 dialogues of 110 utterances) on one GPU; with --gpus N > 1 BASELINE configs[3]
(cfg4: 32 dialogues per GPU = 256 at 8 GPUs), weak scaling.  --ragged at N > 1 draws ONE global batch of 32 N dialogues
and shards it over the ranks by sum(L^2) (distributed.shard_dialogues); the JSON then carries the per-rank loads and step
times.

With --gpus N > 1 and no torch.distributed environment the script re-launches ITSELF under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` (one rank per GPU over RCCL);
started by such a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  It refuses to
report a world size other than the one asked for.

One "step" = zero_grad -> DialogueGNNModel forward -> FocalLoss -> backward (+ the data-parallel gradient all-reduce
when N > 1) on one synthetic IEMOCAP-shaped batch already resident in HBM.  Weak scaling: every rank processes its own
batch of the configured size (dialogues are independent; the only collective is the flat-bucket gradient all-reduce).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":      live HIP-event timing of the north-star kernel (K6 scatter-propagate) vs the HBM roofline at the
                   bench workload; "roofline_cfg5*": the same kernel family at BASELINE config 5, where one launch moves
                   282 MB (the >= 40 % target is a cfg5 property), buffers rotated so nothing is served by the MALL
  "dominant":      what actually bounds the step at this workload (per-kernel shares from the committed rocprof stats)
  "cpu_baseline":  the CPU oracle (port of the reference's op structure) and, next to it, the vectorised CPU
                   restatement, timed on this host on the same batch, N=1 only.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)   # (a box that idled runs its first replays at a lower clock)
    ap.add_argument("--config", default=None,
                    help="default: cfg2 (BASELINE configs[1], 16 dialogues) at --gpus 1; cfg4 (BASELINE configs[3]: 32 dialogues per "
                         "GPU, 256 at 8 GPUs) at --gpus N > 1")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-part-bucket", action="store_true",
                    help="data parallel: reduce the graph / head half of the gradient bucket while the encoders' backward is "
                         "still running (distributed.GradientBucket(parts=2)); off by default until measured on a multi-GPU node")
    ap.add_argument("--predict-scaling", type=int, default=0, metavar="N",
                    help="single GPU: shard one global ragged batch of 32 N dialogues by sum(L^2) as --gpus N --ragged would, "
                         "time every rank's shard on THIS GPU and print the predicted N-GPU throughput / efficiency")
    ap.add_argument("--no-extra", action="store_true", help="skip the other_workloads legs (cfg2 ragged, cfg3, cfg4 shard, cfg5, streamed)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the K6 roofline legs (profiling runs of the step alone)")
    ap.add_argument("--only-roofline", action="store_true", help="profiling aid: run only the K6 roofline legs (clean rocprof traces)")
    ap.add_argument("--roofline-legs", default="fwd,bwd,stack,d512,b64",
                    help="profiling aid: which cfg5 legs to run (tools/collect_traffic.py separates the per-call legs from the "
                         "stack leg, whose launches share kernel names and grids with them)")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work per cpu_baseline mode")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke tests)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: all ranks use GPU 0 (needs --backend gloo)")
    ap.add_argument("--no-floor", action="store_true",
                    help="do not measure the dependent-launch cost live (a chain of 6 400 tiny launches: it would sit in a "
                         "rocprof kernel summary of this run); the step floor then uses the price list's 1.45 us")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel machinery (process group, flat bucket, all-reduce) even at N = 1")
    ap.add_argument("--eager-allreduce", action="store_true",
                    help="keep the gradient all-reduce outside the captured step (default: captured with it, eager on failure)")
    return ap.parse_args()


def self_launch(a):
    """--gpus N > 1 without a launcher: become `torch.distributed.run` with N ranks on this node."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: needed by RCCL across processes here
    os.execv(sys.executable, cmd)


def profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


TRAFFIC_JSON = "r06_propagate_traffic.json"      # (this round's PMC passes)


def measured_traffic(key, kernel):
    """HBM bytes per launch from the committed PMC passes of THIS round (tools/collect_traffic.py: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --only-roofline`, FETCH x2 on gfx950; the file records the
    commit it was collected at).  The counters cannot share a pass with the timing run, so they are not re-collected
    here; a figure is attached only if the kernel it was measured on still exists, by name, in the library this process
    loaded."""
    d = profile_json(TRAFFIC_JSON)
    if not d or key not in d:
        return None, None
    ent = d[key]
    from mm_dfn_amd import build as _b
    try:
        with open(_b.LIBPATH, "rb") as fh:
            present = ent["kernel"].encode() in fh.read()
    except OSError:
        present = False
    if not present:
        return None, "profiles/%s names kernel %r, which is not in the current library: not attached" % (TRAFFIC_JSON, ent["kernel"])
    return ent["traffic_bytes"], "profiles/%s (commit %s, kernel %s)" % (TRAFFIC_JSON, d.get("commit", "?"), ent["kernel"])


def measured_traffic_bwd():
    """HBM bytes of the cfg5 backward leg = the sum over its three kernels (dH: propagate_split, dA: tile_dot_split +
    cross_dot) of the same committed PMC passes; attached only if all three kernels are still in the loaded library."""
    d = profile_json(TRAFFIC_JSON)
    if not d or "legs" not in d or "cfg5_b32" not in d:
        return None, None
    names = ("tile_dot_split_kernel", "cross_dot_kernel")
    legs = {n: [l for l in d["legs"] if l["kernel"] == n] for n in names}
    if not all(legs.values()):
        return None, None
    from mm_dfn_amd import build as _b
    try:
        blob = open(_b.LIBPATH, "rb").read()
    except OSError:
        return None, None
    if not all(n.encode() in blob for n in names + ("propagate_split_kernel",)):
        return None, "profiles/%s names kernels that are not in the current library: not attached" % TRAFFIC_JSON
    total = d["cfg5_b32"]["traffic_bytes"] + sum(max(l["traffic_bytes"] for l in legs[n]) for n in names)
    return total, "profiles/%s (commit %s): propagate_split + tile_dot_split + cross_dot" % (TRAFFIC_JSON, d.get("commit", "?"))


def time_propagate(make_set, nsets, iters, warm_replays=10, timed_replays=3):
    """Average duration (ms) of one K6 propagate launch.  ``nsets`` independent (adjacency, H, out) buffer sets are
    rotated launch by launch, sized so that together they exceed the 256 MB Infinity Cache: no launch finds its
    operands in the MALL.  `iters` back-to-back launches are captured in a hipGraph (the host is out of the picture)
    and bracketed by HIP events on the replay stream."""
    import torch
    from mm_dfn_amd import ops
    sets = [make_set(i) for i in range(nsets)]
    outs = [torch.empty_like(H) for _, H in sets]
    for (adj, H), out in zip(sets, outs):
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            adj, H = sets[i % nsets]
            ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout, out=outs[i % nsets])
    # sustained-load warm-up: the first few milliseconds after an idle period run at a lower engine clock
    for _ in range(warm_replays):
        g.replay()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(timed_replays):
        g.replay()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / (iters * timed_replays)


def dependent_launch_us(n=256, replays=20):
    """Cost of ONE MORE dependent launch inside a replayed hipGraph, measured live: a captured chain of `n` one-workgroup
    kernels (the package's smallest launch: a 4-element Adam step on its own buffers), each depending on its predecessor
    through the stream; HIP events around the replays.  This is the per-launch term of the step floor (tools/step_floor.py);
    MI355X_MICROARCH.md's price list has 1.45 us for it between trivial kernels."""
    import torch
    from mm_dfn_amd import _hip
    dev = torch.device("cuda", torch.cuda.current_device())
    p = torch.zeros(4, device=dev)
    g = torch.zeros(4, device=dev)
    m = torch.zeros(4, device=dev)
    v = torch.zeros(4, device=dev)

    def one():
        _hip.check(_hip.lib().mmdfn_adam_step(_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), 4, 1e-3, 0.9, 0.999, 1e-8,
                                              0.0, 1, _hip.stream()), "mmdfn_adam_step")
    one()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            one()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(replays):
        gr.replay()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * replays)


def attach_step_floor(dominant, config, lengths, t_launch_us):
    """`dominant.families[*].bound` for EVERY family + `step_floor_ms`: tools/step_floor.py's analytic
    inventory of the step's launch classes, priced per class at max(algorithmic bytes / 8 TB/s, flops / matrix peak, serial
    chain, the dependent-launch cost measured above); the family times they are compared with are the committed rocprof
    summary's (`dominant.source`)."""
    if not dominant or config not in ("cfg2", "cfg3", "cfg4"):
        return dominant
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import step_floor
    finally:
        sys.path.pop(0)
    r = step_floor.for_config(config, lengths=lengths, t_launch_us=t_launch_us, measured=dominant.get("families"))
    for k, ent in r["families"].items():
        fam = dominant["families"].get(k)
        if fam is None:
            continue
        prev = fam.get("bound")
        fam["bound"] = {"floor_us_per_step": ent["floor_us"], "floor_f32_mfma_us_per_step": ent["floor_f32_mfma_us"],
                        "frac": ent.get("frac"), "launches_modelled": ent["launches_modelled"], "classes": ent["classes"]}
        if prev is not None:
            fam["bound"]["detail"] = prev                      # (the GRU family's chain model of tools/step_breakdown.py)
    dominant["step_floor_ms"] = r["step_floor_us"] / 1e3
    dominant["step_floor_f32_mfma_ms"] = r["step_floor_f32_mfma_us"] / 1e3
    dominant["step_floor_method"] = r["method"]
    dominant["dependent_launch_us"] = t_launch_us
    if dominant.get("kernel_us_per_step"):
        dominant["step_floor_frac_of_kernel_time"] = round(r["step_floor_us"] / dominant["kernel_us_per_step"], 3)
    return dominant


def timed_replays(cap, steps, warmup, post=None):
    import torch
    for _ in range(warmup):
        cap.replay()
        if post:
            post()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cap.replay()
        if post:
            post()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def quick_leg(cfgname, ragged, dropout, steps=12, warmup=4):
    """One more workload through the same captured step (SURVEY 8d asks for cfg2 ragged, cfg3 and the cfg4 shard
    next to the headline): returns {utterances_per_s, ms_per_step, ...}.  Single GPU, fewer steps, not the headline."""
    import torch
    from mm_dfn_amd import FocalLoss, synthetic, train
    from mm_dfn_amd.graphs import CapturedStep
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = dict(synthetic.CONFIGS[cfgname])
    model = synthetic.build_model(dropout=dropout, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev).train()
    batch = synthetic.make_batch(2021, ragged=ragged, device=dev, **cfg)
    lengths = batch["lengths"]
    label = train.flatten_labels(batch["label"], lengths)
    loss_f = FocalLoss(gamma=0.5)

    def fwd_bwd():
        logp = model(batch["textf"], batch["qmask"], batch["umask"], lengths, batch["acouf"], batch["visuf"])[0]
        loss = loss_f(logp, label)
        train.backward(loss)
        return loss

    dt = timed_replays(CapturedStep(model, fwd_bwd, warmup=2), steps, warmup)
    n = sum(lengths)
    return {"workload": "%s%s: B=%d, L<=%d, P=%d, %d GCN layers, dims %d/%d/%d" % (
                cfgname, " ragged" if ragged else "", cfg["B"], cfg["L"], cfg["P"], cfg["nlayers"], cfg["D_t"], cfg["D_a"],
                cfg["D_v"]),
            "utterances": n, "padded_rows": max(lengths) * len(lengths), "ms_per_step": dt * 1e3,
            "utterances_per_s": n / dt, "steps": steps}


def predict_scaling(world, cfgname, dropout, steps=12, warmup=4):
    """What `--gpus N --ragged` would measure, predicted on ONE GPU: the global ragged batch of B x N dialogues (same seed as
    the real run) is sharded by sum(L^2) exactly as distributed.shard_dialogues does, every rank's shard is stepped on this
    GPU as its own captured step, and the N-GPU step time is modelled as max over ranks (the barrier) + the ring
    all-reduce of the flat gradient bucket over xGMI (2 (N-1)/N x bytes / 153 GB/s per link + 20 us).
    The equal-shard figure (every rank a fixed-length cfg4 shard) is printed next to it: their ratio is the price of the
    ragged imbalance alone.  Gives the first real 8-GPU run a number to be checked against."""
    import numpy as np
    import torch
    from mm_dfn_amd import FocalLoss, distributed, synthetic, train
    from mm_dfn_amd.graphs import CapturedStep
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = dict(synthetic.CONFIGS[cfgname])
    glens = synthetic.make_lengths(np.random.RandomState(2021), cfg["B"] * world, cfg["L"], True)
    model = synthetic.build_model(dropout=dropout, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev).train()
    loss_f = FocalLoss(gamma=0.5)

    def time_shard(lengths, seed):
        batch = synthetic.make_batch(seed, lengths=lengths, device=dev, **cfg)
        label = train.flatten_labels(batch["label"], lengths)

        def fwd_bwd():
            logp = model(batch["textf"], batch["qmask"], batch["umask"], lengths, batch["acouf"], batch["visuf"])[0]
            loss = loss_f(logp, label)
            train.backward(loss)
            return loss
        dt = timed_replays(CapturedStep(model, fwd_bwd, warmup=2), steps, warmup)
        torch.cuda.empty_cache()
        return dt

    ranks = []
    for r in range(world):
        mine = distributed.shard_dialogues(glens, world, r)
        lens = [glens[i] for i in mine]
        ranks.append({"rank": r, "dialogues": len(lens), "utterances": sum(lens), "sumL2": sum(x * x for x in lens),
                      "ms_per_step": time_shard(lens, 2021 + r) * 1e3})
    fixed_ms = time_shard([cfg["L"]] * cfg["B"], 2021) * 1e3
    live = [p for p in model.parameters() if p.grad is not None]
    bucket_bytes = 4 * sum(distributed.slot_size(p) for p in live)
    comm_ms = 0.0 if world == 1 else (2.0 * (world - 1) / world * bucket_bytes / 153e9 + 20e-6) * 1e3
    slow = max(x["ms_per_step"] for x in ranks)
    total = sum(x["utterances"] for x in ranks)
    one = ranks[0]["utterances"] / (ranks[0]["ms_per_step"] * 1e-3) if world == 1 else None
    mean_rate = sum(x["utterances"] / (x["ms_per_step"] * 1e-3) for x in ranks) / world      # what one GPU does on such a shard
    return {"mode": "predict-scaling", "n_gpus_modelled": world, "config": cfgname, "ranks": ranks,
            "gradient_bucket_bytes": bucket_bytes, "allreduce_model_ms": comm_ms,
            "predicted_ms_per_step": slow + comm_ms, "predicted_value": total / ((slow + comm_ms) * 1e-3),
            "predicted_efficiency_vs_one_gpu_on_its_shard": total / ((slow + comm_ms) * 1e-3) / (world * mean_rate),
            "imbalance_only": sum(x["ms_per_step"] for x in ranks) / world / slow,
            "fixed_length_shard": {"ms_per_step": fixed_ms, "predicted_ms_per_step": fixed_ms + comm_ms,
                                   "predicted_efficiency": fixed_ms / (fixed_ms + comm_ms)},
            "single_gpu_rate": one, "unit": "utterances/s",
            "note": "all-reduce modelled, not measured (one GPU per box); not overlapped with the backward pass (one-part bucket)"}


def cfg5_leg(name, dropout, steps=6, warmup=2):
    """BASELINE config 5 through the module stack (mm_dfn_amd.MultiStreamGraphModel: six 512-d streams, L = 512,
    8 GCN layers, d = 100): fwd + FocalLoss + bwd of the whole model as one captured step."""
    import torch
    from mm_dfn_amd import FocalLoss, synthetic, train
    from mm_dfn_amd.graphs import CapturedStep
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = dict(synthetic.STREAM_CONFIGS[name])
    model = synthetic.build_stream_model(dropout=dropout, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev).train()
    batch = synthetic.make_stream_batch(2021, device=dev, **cfg)
    lengths = batch["lengths"]
    label = train.flatten_labels(batch["label"], lengths)
    loss_f = FocalLoss(gamma=0.5)

    def fwd_bwd():
        loss = loss_f(model(batch["streams"], batch["qmask"], batch["umask"], lengths)[0], label)
        train.backward(loss)
        return loss

    dt = timed_replays(CapturedStep(model, fwd_bwd, warmup=1), steps, warmup)
    n = sum(lengths)
    return {"workload": "%s: B=%d dialogues, L=%d, M=%d streams x %d-d, %d GCN layers, d=100 (MultiStreamGraphModel: "
                        "projections + graph stack + head + loss)" % (name, cfg["B"], cfg["L"], len(cfg["D_streams"]),
                                                                      cfg["D_streams"][0], cfg["nlayers"]),
            "utterances": n, "ms_per_step": dt * 1e3, "utterances_per_s": n / dt, "steps": steps}


def streamed_leg(cfgname, dropout, nbatches=32, passes=2):
    """The drop-in pass loop (train.train_or_eval_graph_model incl. torch Adam) over `nbatches` DIFFERENT ragged
    batches streamed from pinned host memory: eager launches vs the shape-keyed captured-step cache (second pass of
    the cache = all replays, which is every epoch after the first in a real run: the reference re-seeds per pass)."""
    import torch
    from mm_dfn_amd import FocalLoss, synthetic, train
    from mm_dfn_amd import data as D
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = dict(synthetic.CONFIGS[cfgname])
    batches = []
    for i in range(nbatches):
        b = synthetic.make_batch(3000 + i, ragged=True, **cfg)
        batches.append([b["textf"].pin_memory(), b["visuf"].pin_memory(), b["acouf"].pin_memory(), b["qmask"].pin_memory(),
                        b["umask"].pin_memory(), b["label"].pin_memory(), ["b%d" % i]])
    n_utt = sum(int(b[4].sum()) for b in batches)
    loss_f = FocalLoss(gamma=0.5)
    res = {}
    from mm_dfn_amd.optim import FlatAdam

    def make(seed0, n):
        out = []
        for i in range(n):
            b = synthetic.make_batch(seed0 + i, ragged=True, **cfg)
            out.append([b["textf"].pin_memory(), b["visuf"].pin_memory(), b["acouf"].pin_memory(), b["qmask"].pin_memory(),
                        b["umask"].pin_memory(), b["label"].pin_memory(), ["u%d" % i]])
        return out
    # ---- bucketed entries (train.StepGraphCache(bucket_rows=32)): the cache is warmed with OTHER batches (four passes'
    # worth of different length tuples), then three more passes stream `nbatches` batches each whose tuples it has never
    # seen; ONE prefetcher serves all passes (its device staging ring persists, as with a loader iterated every epoch)
    try:
        model = synthetic.build_model(dropout=dropout, **cfg)
        model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
        model = model.to(dev)
        opt = FlatAdam(model, lr=3e-4, weight_decay=1e-4)
        cache = train.StepGraphCache(model, loss_f, max_entries=96, bucket_rows=32)
        pre = D.DevicePrefetcher(make(7000, nbatches), device=dev)
        # the bucket set is captured BEFORE the first pass from four passes' worth of batches drawn like the ones to come
        # (StepGraphCache.precapture: forward + loss + backward of the first batch of every new bucket, no optimizer, no metrics)
        # (two real steps first: FlatAdam lays the parameters out in its flat buffer at its first step, and a captured step bakes
        # the parameter storages -- entries captured before that would all be captured again)
        pre.loader = make(6000, 2)
        train.train_or_eval_graph_model(model, loss_f, pre, 0, True, opt, False, graph_cache=cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        made = 0
        for w in range(4):
            pre.loader = make(7000 + 100 * w, nbatches)
            made += cache.precapture(pre, train_flag=True)
        torch.cuda.synchronize()
        precapture_s = time.perf_counter() - t0
        passes_out = []
        for w in range(3):
            unseen = make(9000 + 100 * w, nbatches)
            pre.loader = unseen
            h0, m0, f0 = cache.hits, cache.misses, cache.fallbacks
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train.train_or_eval_graph_model(model, loss_f, pre, 0, True, opt, False, graph_cache=cache)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            nu = sum(int(b[4].sum()) for b in unseen)
            passes_out.append({"replayed": cache.hits - h0, "captured": cache.misses - m0,
                               "served_by_a_larger_bucket": cache.fallbacks - f0, "ms_per_step": dt / nbatches * 1e3,
                               "utterances_per_s": nu / dt})
        res["bucketed_unseen_tuples_flat_adam"] = {
            "bucket_rows": 32, "entries": len(cache.entries), "precaptured_entries": made, "precapture_s": precapture_s,
            "note": "the bucket set is captured ahead of the first pass (StepGraphCache.precapture over 4 x %d batches drawn like "
                    "the timed ones: precapture_s); every timed pass -- the FIRST one included -- streams length tuples the cache has "
                    "never seen; a batch whose own bucket was not drawn is served by a larger captured bucket of its (B, L) when one is within "
                    "reach of the padding dialogue (served_by_a_larger_bucket), and captured (55-90 ms) only otherwise; "
                    "tools/streamed_gap.py splits a replayed step: 0.96 ms device time of the real dialogues (exact-signature replays, "
                    "resident inputs), +0.02 for the bucket (padding dialogue, rounding, index retarget), +0.10-0.14 for what the pass "
                    "loop adds on the device around a replay (gradient pack, FlatAdam step, plane refresh, metrics copies, one graph "
                    "launch per step), +0.10-0.20 with the batches coming from pinned host memory (the host itself issues a step in "
                    "~0.6 ms and then waits for the device in the staging ring)" % nbatches, "passes": passes_out}
        del model, opt, cache, pre
        torch.cuda.empty_cache()
    except Exception as exc:
        res["bucketed_unseen_tuples_flat_adam"] = {"skipped": "%s: %s" % (type(exc).__name__, exc)}
    for mode in ("eager", "captured", "captured_flat_adam"):
        model = synthetic.build_model(dropout=dropout, **cfg)
        model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
        model = model.to(dev)
        # torch.optim.Adam as the reference uses it, or this package's fused flat Adam (same update, one launch)
        opt = (FlatAdam(model, lr=3e-4, weight_decay=1e-4) if mode == "captured_flat_adam" else
               torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=1e-4))
        cache = train.StepGraphCache(model, loss_f, max_entries=nbatches + 4) if mode != "eager" else None
        times = []
        for _ in range(passes + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train.train_or_eval_graph_model(model, loss_f, D.DevicePrefetcher(batches, device=dev), 0, True, opt, False,
                                            graph_cache=cache)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        res[mode] = {"first_pass_s": times[0], "steady_pass_s": min(times[1:]),
                     "utterances_per_s": n_utt / min(times[1:]), "ms_per_step": min(times[1:]) / nbatches * 1e3}
        del model, opt, cache
    return {"workload": "%s ragged, %d different batches streamed through train_or_eval_graph_model (fwd + loss + bwd + "
                        "Adam step [torch.optim.Adam; captured_flat_adam: mm_dfn_amd.optim.FlatAdam], metrics, pinned-host "
                        "prefetch)" % (cfgname, nbatches),
            "utterances_per_pass": n_utt, **res}


def cpu_baseline(cfg, batch, state, threads, dropout, budget_s):
    """Times the two CPU restatements on the host, SAME batch and dropout as the GPU step, fwd + loss + bwd:
    ``port`` = oracle/mmdfn_oracle.py (the reference's op structure: dense adjacency, per-speaker GRU passes, dense
    A.H) and ``vectorised`` = oracle/mmdfn_vectorised.py (block tiles, batched party GRU)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mmdfn_oracle as O
    import mmdfn_vectorised as V
    # torch's default (all hardware threads) oversubscribes badly on a 256-thread host for these small
    # ops; 16 intra-op threads was the fastest setting measured (8/16/32/64 tried), override with --cpu-threads
    torch.set_num_threads(threads if threads > 0 else min(16, os.cpu_count() or 1))
    used = torch.get_num_threads()
    lens = batch["lengths"]
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in state.items()}
    ocfg = O.default_cfg(cfg["nlayers"], dropout=dropout)
    label = O.flatten_labels(batch["label"].cpu(), lens)
    args = (batch["textf"].cpu(), batch["qmask"].cpu(), batch["umask"].cpu(), lens, batch["acouf"].cpu(),
            batch["visuf"].cpu())

    def run(fwd):
        def step():
            for p in params.values():
                p.grad = None
            O.focal_loss(fwd(), label, 0.5).backward()
        step()  # warm-up
        t0 = time.time()
        n = 0
        while True:
            step()
            n += 1
            if time.time() - t0 > budget_s or n >= 20:
                break
        return (time.time() - t0) / n, n

    dt_p, n_p = run(lambda: O.forward(params, *args, ocfg, training=True, engine="aten"))
    dt_v, n_v = run(lambda: V.forward(params, *args, ocfg, training=True))
    sample = "the full bench batch (%d dialogues, N=%d utt), dropout %.2f, fwd+loss+bwd" % (len(lens), sum(lens), dropout)
    port = {"value": sum(lens) / dt_p, "unit": "utterances/s", "cores": used, "kind": "port",
            "sample": "%s, %d steps of oracle/mmdfn_oracle.py (dense adjacency, per-speaker GRU passes, aten GRU), "
                      "%.2f s/step" % (sample, n_p, dt_p)}
    port["vectorised"] = {"value": sum(lens) / dt_v, "unit": "utterances/s", "cores": used, "kind": "port",
                          "sample": "%s, %d steps of oracle/mmdfn_vectorised.py (block-tile adjacency, batched party "
                                    "GRU; SURVEY 8d's conservative comparison), %.3f s/step" % (sample, n_v, dt_v)}
    ref = profile_json("r01_reference_cpu_build_container.json")
    if ref:
        port["reference_on_build_container"] = ref
    return port


def main():
    a = parse()
    if a.config is None:
        a.config = "cfg2" if (a.gpus == 1 and not a.predict_scaling) else "cfg4"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    # stdout carries ONE JSON line and nothing else: libraries that print to the C-level stdout (RCCL's version banner
    # at teardown, gloo's connection log) are pointed at stderr; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: refusing to report a world size that was not asked for" % (world, a.gpus))
    if a.share_gpu:
        local = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d visible devices" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from mm_dfn_amd import FocalLoss, synthetic, train, ops, distributed
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # eager + captured steps share parameters

    if a.predict_scaling:
        out = predict_scaling(a.predict_scaling, a.config if a.config in ("cfg2", "cfg4") else "cfg4", a.dropout)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        return
    if a.only_roofline:
        out = {"note": "K6 roofline legs only (profiling aid)"}
        cfg = dict(synthetic.CONFIGS[a.config])
        lengths = synthetic.make_lengths(None, cfg["B"], cfg["L"], False)
        roofline_legs(out, a, dev, sum(lengths), lengths)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        return
    use_dp = world > 1 or a.force_dp
    if use_dp:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        distributed.init(backend=a.backend)

    cfg = dict(synthetic.CONFIGS[a.config])
    model = synthetic.build_model(dropout=a.dropout, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev).train()
    shard_info = None
    if use_dp and world > 1 and a.ragged:
        # ONE global ragged batch of B * world dialogues (the same on every rank: seeded), sharded by sum(L^2)
        import numpy as np
        glens = synthetic.make_lengths(np.random.RandomState(2021), cfg["B"] * world, cfg["L"], True)
        mine = distributed.shard_dialogues(glens, world, rank)
        batch = synthetic.make_batch(2021 + rank, lengths=[glens[i] for i in mine], device=dev, **cfg)
        loads = [sum(glens[i] ** 2 for i in distributed.shard_dialogues(glens, world, r)) for r in range(world)]
        shard_info = {"global_dialogues": len(glens), "global_utterances": sum(glens), "sumL2_max": max(loads),
                      "sumL2_min": min(loads), "dialogues_per_rank": [len(distributed.shard_dialogues(glens, world, r))
                                                                      for r in range(world)]}
    else:
        batch = synthetic.make_batch(2021 + rank, ragged=a.ragged, device=dev, **cfg)
    lengths = batch["lengths"]
    n_utt = sum(lengths)
    label = train.flatten_labels(batch["label"], lengths)
    loss_f = FocalLoss(gamma=0.5)
    # sum-reduce with the 1/world factor folded into the loss scale below: no separate averaging kernel after the all-reduce
    dp = distributed.GradientBucket(model, average=False, parts=2 if a.two_part_bucket else 1) if use_dp else None
    total_utt = distributed.all_reduce_scalar(n_utt) if use_dp else n_utt

    scale = (n_utt / total_utt) if dp is not None else 1.0   # local mean -> this rank's share of the GLOBAL mean

    def fwd_bwd():
        logp = model(batch["textf"], batch["qmask"], batch["umask"], lengths, batch["acouf"], batch["visuf"])[0]
        loss = loss_f(logp, label)
        if dp is not None:
            loss = loss * scale        # the summed bucket is then the gradient of the mean over ALL ranks' utterances
            dp.arm()                   # (two-part bucket: its first collective starts inside this backward pass)
        train.backward(loss)
        return loss

    def eager_step():
        model.zero_grad(set_to_none=True)
        loss = fwd_bwd()
        ops.join_weight_grads()
        if dp is not None:
            dp.all_reduce()
        return loss

    step = eager_step
    launch_mode = "eager"
    allreduce_mode = None if dp is None else "eager after the step"
    if not a.no_graph:
        from mm_dfn_amd.graphs import CapturedStep
        captured = None
        if dp is not None and not a.eager_allreduce and a.backend == "nccl":   # gloo collectives are host-driven: not capturable
            # the all-reduce as a node of the captured step: no host launch between backward and the collective
            try:
                captured = CapturedStep(model, fwd_bwd, warmup=3, bucket=dp, reduce_in_graph=True)
                allreduce_mode = "captured in the step's hipGraph"
                step = captured.replay
            except Exception as exc:
                allreduce_fallback = "%s: %s" % (type(exc).__name__, exc)
                allreduce_mode = "eager after the step (capturing it failed: %s)" % allreduce_fallback
                print("[bench] capturing the all-reduce failed (%s: %s); all-reduce stays eager" % (type(exc).__name__, exc),
                      file=sys.stderr)
                torch.cuda.synchronize()
                captured = None
                dp = distributed.GradientBucket(model, average=False, parts=2 if a.two_part_bucket else 1)
        if captured is None:
            if dp is not None and a.two_part_bucket:
                # a two-part bucket armed inside a capture bakes its first collective into the graph; with the second one
                # issued eagerly after every replay nothing re-arms it and the first part would be reduced twice:
                # without the in-graph all-reduce the bucket is one part
                print("[bench] --two-part-bucket needs the all-reduce inside the captured step; using a one-part bucket",
                      file=sys.stderr)
                dp = distributed.GradientBucket(model, average=False, parts=1)
                allreduce_mode = (allreduce_mode or "") + " (two-part bucket dropped: all-reduce not captured)"
            try:
                captured = CapturedStep(model, fwd_bwd, warmup=3, bucket=dp)

                def step():
                    loss = captured.replay()
                    if dp is not None:
                        dp.reduce_flat()
                    return loss
            except Exception as exc:  # capture is an optimisation; never lose the measurement over it
                print("[bench] hipGraph capture failed (%s: %s); running eagerly" % (type(exc).__name__, exc), file=sys.stderr)
                torch.cuda.synchronize()
                captured = None
                step = eager_step
        if captured is not None:
            launch_mode = "hipGraph replay of the whole step"

    for _ in range(a.warmup):
        step()
    if use_dp:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if use_dp:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if use_dp:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, (dt / a.steps * 1e3, n_utt))
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # ---- multi-GPU self-check (the first real N-GPU run has no earlier measurement to be compared with): every rank
    # times ITS OWN shard without the collective, rank 0 adds the modelled ring all-reduce of --predict-scaling and prints
    # the prediction next to what was measured, plus the per-rank times (imbalance) and whether the collective was captured.
    self_check = None
    if use_dp and world > 1:
        try:
            from mm_dfn_amd.graphs import CapturedStep

            def fwd_bwd_local():
                logp = model(batch["textf"], batch["qmask"], batch["umask"], lengths, batch["acouf"], batch["visuf"])[0]
                loss = loss_f(logp, label) * scale
                train.backward(loss)
                return loss
            cap_local = CapturedStep(model, fwd_bwd_local, warmup=2)
            local_ms = timed_replays(cap_local, max(10, a.steps // 4), 3) * 1e3
            every = [None] * world
            torch.distributed.all_gather_object(every, local_ms)
            bucket_bytes = dp.flat.numel() * 4 if (dp is not None and dp.flat is not None) else 0
            comm_ms = (2.0 * (world - 1) / world * bucket_bytes / 153e9 + 20e-6) * 1e3
            measured = dt / a.steps * 1e3
            predicted = max(every) + comm_ms
            self_check = {"compute_only_ms_per_rank": every, "imbalance": sum(every) / world / max(every),
                          "allreduce_model_ms": comm_ms, "allreduce_model": "ring over xGMI: 2 (N-1)/N x bytes / 153 GB/s + 20 us, "
                          "not overlapped (one-part bucket)", "predicted_ms_per_step": predicted, "measured_ms_per_step": measured,
                          "measured_over_predicted": measured / predicted,
                          "allreduce_captured": bool(allreduce_mode and allreduce_mode.startswith("captured")),
                          "verdict": ("within 10 % of the model" if abs(measured / predicted - 1.0) <= 0.10 else
                                      "OFF the model by more than 10 %: look at per_rank / allreduce before trusting the value")}
        except Exception as exc:
            self_check = {"skipped": "%s: %s" % (type(exc).__name__, exc)}

    # ---- the same step followed by the fused Adam update (one extra launch), reported next to the headline.
    # Parameters and gradients are flat buffers here, so the step is re-captured against the flat storage.
    adam_ms = None
    if not use_dp and not a.no_graph and launch_mode != "eager":
        try:
            from mm_dfn_amd.graphs import CapturedStep
            from mm_dfn_amd.optim import FlatAdam
            model.zero_grad(set_to_none=True)
            fwd_bwd()
            opt = FlatAdam(model, lr=3e-4, weight_decay=1e-4)
            opt.bucket.flatten()
            opt._materialise()
            cap2 = CapturedStep(model, fwd_bwd, warmup=2, bucket=opt.bucket)
            adam_ms = timed_replays(cap2, a.steps, 3, post=lambda: opt.step(grads_already_flat=True)) * 1e3
        except Exception as exc:
            print("[bench] fused-Adam leg skipped: %s" % exc, file=sys.stderr)

    if rank == 0:
        out = {
            "metric": "utterances/sec (fwd+bwd), IEMOCAP-shaped batch", "value": total_utt * a.steps / dt,
            "unit": "utterances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "fp32 in, fp32 out, fp32 accumulation everywhere.  Launches with many rows (the party-GRU input "
                          "projection at cfg2; K6 / K6' / projections at cfg5) carry the fp32 product on bf16 MFMAs through "
                          "three exact bf16 pieces per operand (six piece products): error vs fp64 at fp32 rounding level "
                          "(<= 4x the exact-f32 MFMA kernel's, tests/test_graph_kernels_gpu.py), not a reduced-precision mode",
            "data": "synthetic", "launch": launch_mode, "allreduce": allreduce_mode,
            "config": {"workload": "%s (BASELINE %s): B=%d dialogues/GPU, L=%s, dims %d/%d/%d, %d GCN layers, P=%d, dropout %.2f"
                                   % (a.config, {"cfg1": "configs[0]", "cfg2": "configs[1]", "cfg3": "configs[2]",
                                                 "cfg4": "configs[3], %d dialogues global" % (cfg["B"] * world)}.get(a.config, "-"),
                                      cfg["B"], "ragged<=%d" % cfg["L"] if a.ragged else cfg["L"], cfg["D_t"],
                                      cfg["D_a"], cfg["D_v"], cfg["nlayers"], cfg["P"], a.dropout),
                       "utterances_per_gpu": n_utt, "parallelism": "dp%d" % world},
            "with_fused_adam_step": None if adam_ms is None else {"ms_per_step": adam_ms,
                                                                   "value": total_utt / (adam_ms * 1e-3)},
        }
        if use_dp:
            out["per_rank"] = {"ms_per_step": [x[0] for x in per_rank], "utterances": [x[1] for x in per_rank]}
            if shard_info is not None:
                out["shard"] = shard_info
            if self_check is not None:
                out["self_check"] = self_check
        if dp is not None and dp.flat is not None:
            out["gradient_bucket"] = {"floats": dp.flat.numel(), "bytes": dp.flat.numel() * 4, "backend": a.backend}
        if not a.no_roofline:
            roofline_legs(out, a, dev, n_utt, lengths)
        out["dominant"] = (profile_json("r06_step_breakdown_%s.json" % a.config) or profile_json("r05_step_breakdown_%s.json" % a.config)
                           or profile_json("r04_step_breakdown_%s.json" % a.config))
        try:
            out["dominant"] = attach_step_floor(out["dominant"], a.config, [int(x) for x in lengths],
                                                1.45 if a.no_floor else dependent_launch_us())
        except Exception as exc:
            print("[bench] step floor not attached: %s: %s" % (type(exc).__name__, exc), file=sys.stderr)
        if world == 1 and not a.no_extra and not use_dp:
            out["other_workloads"] = []
            legs = [lambda c=c, r=r: quick_leg(c, r, a.dropout) for c, r in
                    (("cfg2", True), ("cfg2_refdims", False), ("cfg3", True), ("cfg4", False), ("cfg4", True))]
            legs += [lambda: cfg5_leg("cfg5", a.dropout), lambda: cfg5_leg("cfg5_b32", a.dropout, steps=4, warmup=1),
                     lambda: streamed_leg("cfg2", a.dropout)]
            for leg in legs:
                try:
                    out["other_workloads"].append(leg())
                except Exception as exc:
                    print("[bench] extra workload skipped: %s: %s" % (type(exc).__name__, exc), file=sys.stderr)
                torch.cuda.empty_cache()
        if world == 1 and not a.no_cpu_baseline and not use_dp:
            out["cpu_baseline"] = cpu_baseline(cfg, batch, model.state_dict(), a.cpu_threads, a.dropout, a.cpu_budget)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dp:
        # Normal teardown (round 6): the captured steps that hold a collective node are released FIRST (CapturedStep.close():
        # hipGraphExec + pool gone, device idle), then the ranks meet and the group is destroyed; an exception here -- or a
        # teardown that does not return within two minutes -- ends this rank with a non-zero exit code, so a rank that dies
        # after the result line is visible to the launcher (round 5 left through os._exit(0), which hid exactly that).
        import gc
        import threading
        guard = threading.Timer(120.0, lambda: os._exit(3))
        guard.daemon = True
        guard.start()
        for name in ("captured", "cap_local"):
            cap = locals().get(name)
            if cap is not None and hasattr(cap, "close"):
                cap.close()
        cap = None
        gc.collect()
        torch.cuda.synchronize()
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        guard.cancel()


def roofline_legs(out, a, dev, n_utt, lengths):
    import torch
    from mm_dfn_amd import ops
    d = 100
    # ---- the north-star kernel (K6 scatter-propagate forward, d = 100) at the bench workload.  At cfg2 it is ~2 % of
    # the step and launch-latency bound (6.6 MB per launch); the bandwidth result is the cfg5 leg below.
    def mk(i):
        g = torch.Generator(device=dev).manual_seed(100 + i)
        adj = ops.build_adjacency(torch.randn(3, n_utt, 200, device=dev, generator=g), lengths)
        return adj, torch.randn(3 * n_utt, d, device=dev, generator=g)

    ms = time_propagate(mk, nsets=48, iters=192)
    lay = ops.DialogueLayout.get(lengths, 3, dev)
    alg_bytes = lay.propagate_bytes(d)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    traffic, src = measured_traffic("cfg2", "propagate_v2_kernel") if (a.config == "cfg2" and not a.ragged) else (None, None)
    out["roofline"] = {"bound": "hbm", "kernel": "propagate_v2_kernel<2,4,2,16,1> (K6 fwd, d=100, exact-f32 MFMA)",
                       "role": "north-star target kernel (BASELINE.json: 'GCN scatter-propagate'); NOT the dominant cost of "
                               "this workload, see 'dominant'; its HBM-roofline figure of merit is 'roofline_cfg5'",
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "algorithmic_bytes": alg_bytes, "avg_launch_us": ms * 1e3, "traffic": traffic,
                       "traffic_source": src,
                       "buffers": "48 rotating (adjacency, H, out) sets = %.0f MB > 256 MB MALL" % (48 * (alg_bytes / 1e6))}
    # the same kernel on BASELINE config 5 (L=512, M=6, d=100, 32 dialogues), where one launch moves 282 MB and
    # the launch-latency floor no longer hides the kernel (the >=40 % HBM target is a cfg5 property)
    try:
        l5 = [512] * 32

        def mk5(i):
            g = torch.Generator(device=dev).manual_seed(500 + i)
            adj = ops.build_adjacency(torch.randn(6, sum(l5), 200, device=dev, generator=g), l5)
            return adj, torch.randn(6 * sum(l5), d, device=dev, generator=g)

        ms5 = time_propagate(mk5, nsets=3, iters=21, warm_replays=15, timed_replays=5)
        lay5 = ops.DialogueLayout.get(l5, 6, dev)
        b5 = lay5.propagate_bytes(d)
        t5, src5 = measured_traffic("cfg5_b32", "propagate_split_kernel")
        out["roofline_cfg5"] = {"workload": "cfg5: B=32, L=512, M=6, d=100", "bound": "hbm",
                                "kernel": "propagate_split_kernel<0, true> (K6 fwd, bf16-piece MFMA, three 32-column tiles + a 16-column tail tile, fp32-level error)",
                                "traffic": t5, "traffic_source": src5,
                                "achieved": b5 / (ms5 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": b5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": b5,
                                "avg_launch_us": ms5 * 1e3,
                                "buffers": "3 rotating (adjacency, H, out) sets = %.0f MB > 256 MB MALL" % (3 * b5 / 1e6),
                                "useful_tflops": lay5.propagate_flops(d) / (ms5 * 1e-3) / 1e12}
        # the second figure of this leg: at 36 flop per algorithmic byte the launch sits ABOVE the exact-fp32
        # matrix ridge (157.3 TFLOP/s / 8 TB/s = 19.7 flop/B), so the matrix side is its binding roof; priced against the
        # dense fp32 matrix peak (the fp32 product rides on six bf16-piece MFMAs per K = 16)
        fl5 = lay5.propagate_flops(d)
        out["roofline_cfg5"]["second_bound"] = {
            "bound": "mfma_f32", "achieved": fl5 / (ms5 * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
            "frac": fl5 / (ms5 * 1e-3) / 1e12 / 157.3, "flop_per_algorithmic_byte": fl5 / b5,
            "ridge_flop_per_byte": 157.3e12 / (HBM_PEAK_GBS * 1e9),
            "note": "useful fp32 flop (2 d nnz) per launch / launch time; the kernel's issue-side accounting (MFMA ~20 us + "
                    "piece cutting and other issue 25 us + epilogue per workgroup pair, no overlap between them on a SIMD) "
                    "is in DESIGN.md 4g / 4k, profiles/r03_k6_memory_path.md, profiles/r05_k6_levers_upper_bounds.md"}
        # the same launch with 64 dialogues: 1 536 workgroups = three FULL rounds of the chip's 512 workgroup slots (32 dialogues:
        # 768 = one and a half) -- what the round quantisation of the figure above is worth; not the figure of merit
        if "b64" in set(a.roofline_legs.split(",")) and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            try:
                l6 = [512] * 64

                def mk6(i):
                    g = torch.Generator(device=dev).manual_seed(600 + i)
                    adj = ops.build_adjacency(torch.randn(6, sum(l6), 200, device=dev, generator=g), l6)
                    return adj, torch.randn(6 * sum(l6), d, device=dev, generator=g)

                ms6 = time_propagate(mk6, nsets=2, iters=10, warm_replays=10, timed_replays=5)
                b6 = ops.DialogueLayout.get(l6, 6, dev).propagate_bytes(d)
                out["roofline_cfg5_b64"] = {"workload": "cfg5 shapes, B=64: L=512, M=6, d=100 (three full rounds of workgroups)",
                                            "bound": "hbm", "kernel": out["roofline_cfg5"]["kernel"],
                                            "achieved": b6 / (ms6 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": b6 / (ms6 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": b6,
                                            "avg_launch_us": ms6 * 1e3, "traffic": None,
                                            "buffers": "2 rotating (adjacency, H, out) sets = %.0f MB > 256 MB MALL" % (2 * b6 / 1e6)}
                del mk6
                torch.cuda.empty_cache()
            except Exception as exc:
                out["roofline_cfg5_b64"] = {"skipped": "%s: %s" % (type(exc).__name__, exc)}
        # K6 backward at the same workload, reported separately: dH = A^T dO (the forward kernel, A is
        # symmetric) + dA = dO . H^T on the tile pattern (tile_dot + cross_dot); bytes_bwd = 8 nnz + 16 M N d
        legs5 = set(a.roofline_legs.split(","))
        adj5, H5 = mk5(0)
        dO5 = torch.randn_like(H5)
        gb = None
        if "bwd" in legs5:
            for _ in range(3):
                ops.propagate_raw(adj5.tiles, adj5.cross, dO5, adj5.layout)
                ops.tile_outer_raw(dO5, H5, adj5.layout)
            torch.cuda.synchronize()
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb):
                for _ in range(10):
                    ops.propagate_raw(adj5.tiles, adj5.cross, dO5, adj5.layout)
                    ops.tile_outer_raw(dO5, H5, adj5.layout)
            for _ in range(10):
                gb.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gb.replay()
            e1.record()
            e1.synchronize()
            msb = e0.elapsed_time(e1) / 50
            bb = 8 * lay5.nnz + 16 * 6 * sum(l5) * d
            out["roofline_cfg5_bwd"] = {"workload": "cfg5 backward of one K6 call: dH (propagate) + dA (tile_dot + cross_dot)",
                                        "bound": "hbm", "achieved": bb / (msb * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": bb / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "algorithmic_bytes": bb, "avg_us": msb * 1e3}
            tb, srcb = measured_traffic_bwd()
            out["roofline_cfg5_bwd"]["traffic"] = tb
            if srcb:
                out["roofline_cfg5_bwd"]["traffic_source"] = srcb
        # the same leg as the GCN stack runs it since round 5 ("if several layers are fused, count A-hat once per
        # fused group"): nl = 8 layers share the adjacency, so the backward is 8 x dH = A^T dO_l and ONE dA = [dhi_1 | .. |
        # dhi_8] [zin_1 | .. | zin_8]^T over the tile pattern (width nl d): bytes = nl (4 nnz + 8 M N d) + 8 M N d nl + 4 nnz
        if "stack" in legs5:
            nl5 = 8
            X8 = torch.randn(6 * sum(l5), nl5 * d, device=dev)
            Y8 = torch.randn(6 * sum(l5), nl5 * d, device=dev)
            dz8 = torch.empty(6 * sum(l5), d, device=dev)
            for _ in range(2):
                for l in range(nl5):
                    ops.propagate_raw(adj5.tiles, adj5.cross, X8[:, l * d:(l + 1) * d], adj5.layout, out=dz8)
                ops.tile_outer_raw(X8, Y8, adj5.layout)
            torch.cuda.synchronize()
            gs = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gs):
                for _ in range(3):
                    for l in range(nl5):      # (dhi_l: a column block of the stack's (MN, nl d) buffer, as gcn_stack.py hands it over)
                        ops.propagate_raw(adj5.tiles, adj5.cross, X8[:, l * d:(l + 1) * d], adj5.layout, out=dz8)
                    ops.tile_outer_raw(X8, Y8, adj5.layout)
            for _ in range(5):
                gs.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gs.replay()
            e1.record()
            e1.synchronize()
            mss = e0.elapsed_time(e1) / 15
            bs = nl5 * (4 * lay5.nnz + 8 * 6 * sum(l5) * d) + 8 * 6 * sum(l5) * d * nl5 + 4 * lay5.nnz
            out["roofline_cfg5_bwd_stack"] = {
                "workload": "cfg5 backward of the K6 calls of one 8-layer stack: 8 x dH (propagate) + ONE dA over all layers "
                            "(tile_dot_split d = 800 + cross_dot pieces)", "bound": "hbm", "achieved": bs / (mss * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bs / (mss * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": bs,
                "avg_us": mss * 1e3, "per_layer_equivalent_us": mss * 1e3 / nl5,
                "note": "A-hat's gradient counted once per stack (one write of the tile array); the per-call leg above is what "
                        "rounds 1-4 ran once per layer (8 read-modify-writes of the tile array)"}
            ts_, srcs_ = measured_traffic("cfg5_b32_bwd_stack", "tile_dot_split_kernel")
            out["roofline_cfg5_bwd_stack"]["traffic"] = ts_
            if srcs_:
                out["roofline_cfg5_bwd_stack"]["traffic_source"] = srcs_
            del gs, X8, Y8, dz8
        del adj5, H5, dO5, gb
        torch.cuda.empty_cache()
        # the d = 512 stress variant SURVEY 8d asks for next to the reference-faithful d = 100: 18.94 MB and 1.63 GFLOP per
        # dialogue-layer = 86 flop/B, above the fp32 ridge (157 TFLOP/s / 8 TB/s = 20 flop/B): MFMA-bound, priced against
        # the dense fp32 matrix peak (the kernel carries the fp32 product on bf16 pieces, 6 MFMA products per fp32 product)
        if "d512" in legs5:
            l8 = [512] * 8

            def mk512(i):
                g = torch.Generator(device=dev).manual_seed(900 + i)
                adj = ops.build_adjacency(torch.randn(6, sum(l8), 200, device=dev, generator=g), l8)
                return adj, torch.randn(6 * sum(l8), 512, device=dev, generator=g)

            ms512 = time_propagate(mk512, nsets=3, iters=12, warm_replays=8, timed_replays=4)
            lay8 = ops.DialogueLayout.get(l8, 6, dev)
            fl = lay8.propagate_flops(512)
            out["roofline_cfg5_d512"] = {"workload": "cfg5 stress variant: B=8, L=512, M=6, d=512 (K6 fwd)", "bound": "mfma",
                                         "note": "MFMA-bound, 86 flop/B", "achieved": fl / (ms512 * 1e-3) / 1e12,
                                         "peak": 157.3, "unit": "TFLOP/s", "frac": fl / (ms512 * 1e-3) / 1e12 / 157.3,
                                         "algorithmic_bytes": lay8.propagate_bytes(512), "avg_launch_us": ms512 * 1e3,
                                         "hbm_frac": lay8.propagate_bytes(512) / (ms512 * 1e-3) / 1e9 / HBM_PEAK_GBS}
    except Exception as exc:
        print("[bench] cfg5 roofline leg skipped: %s" % exc, file=sys.stderr)


if __name__ == "__main__":
    main()
