"""bench.py -- utterances/sec (fwd + loss + bwd) of the MM-DFN hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--ragged]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = zero_grad -> DialogueGNNModel forward -> FocalLoss -> backward (+ the data-parallel
gradient all-reduce when N > 1) on one synthetic IEMOCAP-shaped batch already resident in HBM.
Weak scaling: every rank processes its own batch of the configured size (dialogues are independent;
the only collective is the flat-bucket gradient all-reduce over RCCL).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     live HIP-event timing of the dominant graph kernel (K6 propagate) vs the HBM roofline
  "cpu_baseline": the CPU oracle (port of the reference algorithm) timed on this host, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other_workloads legs (cfg2 ragged, cfg3, cfg4 shard)")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke tests)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: all ranks use GPU 0")
    ap.add_argument("--deferred-wgrad", action="store_true",
                    help="queue weight-gradient kernels and issue them on a side stream in batches (overlapping the GRU backward)")
    ap.add_argument("--async-wgrad", action="store_true",
                    help="enqueue weight-gradient kernels on a side stream (measured slower on MI355X: 2.16 vs 1.99 ms)")
    return ap.parse_args()


def measured_traffic(key):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_propagate_traffic.json): the counters need
    their own rocprofv3 --pmc runs (FETCH_SIZE / WRITE_SIZE do not fit one pass), so they are not re-collected here."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_propagate_traffic.json")) as f:
            return json.load(f)[key]["traffic_bytes"]
    except Exception:
        return None


def time_propagate(adj_builder, lay_d, iters=200, warm_replays=10, timed_replays=3):
    """Average duration (ms) of one K6 propagate launch: `iters` back-to-back launches captured in a hipGraph
    (so the host is out of the picture) and bracketed by HIP events on the replay stream."""
    from mm_dfn_amd import ops
    adj, H = adj_builder()
    for _ in range(5):
        ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            out = ops.propagate_raw(adj.tiles, adj.cross, H, adj.layout)
    # sustained-load warm-up: the first few milliseconds after an idle period run at a lower engine clock
    # (measured: the same launch takes 103 us in the first 100 launches and 91 us afterwards)
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


def quick_leg(cfgname, ragged, dropout, steps=12, warmup=4):
    """One more workload through the same captured step (SURVEY 8d asks for cfg2 ragged, cfg3 and the cfg4 shard
    next to the headline): returns {utterances_per_s, ms_per_step, ...}.  Single GPU, fewer steps, not the headline."""
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
        loss.backward()
        return loss

    cap = CapturedStep(model, fwd_bwd, warmup=2)
    for _ in range(warmup):
        cap.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cap.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = sum(lengths)
    return {"workload": "%s%s: B=%d, L<=%d, P=%d, %d GCN layers, dims %d/%d/%d" % (
                cfgname, " ragged" if ragged else "", cfg["B"], cfg["L"], cfg["P"], cfg["nlayers"], cfg["D_t"], cfg["D_a"],
                cfg["D_v"]),
            "utterances": n, "padded_rows": max(lengths) * len(lengths), "ms_per_step": dt * 1e3,
            "utterances_per_s": n / dt, "steps": steps}


def cpu_baseline(cfg, batch, state, threads, budget_s=20.0):
    """Times the oracle (reference op structure: dense adjacency, looped party GRUs, aten GRU) on the host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mmdfn_oracle as O
    # torch's default (all hardware threads) oversubscribes badly on a 256-thread host for these small
    # ops; 16 intra-op threads was the fastest setting measured (8/16/32/64 tried), override with --cpu-threads
    torch.set_num_threads(threads if threads > 0 else min(16, os.cpu_count() or 1))
    used = torch.get_num_threads()
    # bounded sample: the first `nb` dialogues of the batch (reference CPU throughput peaks near B=16)
    nb = min(len(batch["lengths"]), 8)
    lens = batch["lengths"][:nb]
    L = max(lens)
    sl = lambda t: t[:L, :nb].contiguous()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in state.items()}
    ocfg = O.default_cfg(cfg["nlayers"], dropout=0.0)
    label = O.flatten_labels(batch["label"][:nb, :L].cpu(), lens)
    args = (sl(batch["textf"].cpu()), sl(batch["qmask"].cpu()), batch["umask"][:nb, :L].cpu(), lens,
            sl(batch["acouf"].cpu()), sl(batch["visuf"].cpu()))

    def step():
        for p in params.values():
            p.grad = None
        logp = O.forward(params, *args, ocfg, training=True, engine="aten")
        O.focal_loss(logp, label, 0.5).backward()

    step()  # warm-up
    t0 = time.time()
    n = 0
    while True:
        step()
        n += 1
        if time.time() - t0 > budget_s or n >= 20:
            break
    dt = (time.time() - t0) / n
    return {"value": sum(lens) / dt, "unit": "utterances/s", "cores": used, "kind": "port",
            "sample": "%d dialogues (L<=%d, N=%d utt) of the bench batch, %d fwd+bwd steps, oracle/mmdfn_oracle.py "
                      "(dense adjacency, per-speaker GRU passes, aten GRU), %.2f s/step" % (nb, L, sum(lens), n, dt)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from mm_dfn_amd import FocalLoss, synthetic, train, ops, distributed
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # eager + captured steps share parameters

    if world > 1:
        distributed.init(backend=a.backend)

    cfg = dict(synthetic.CONFIGS[a.config])
    model = synthetic.build_model(dropout=a.dropout, **cfg)
    model.load_state_dict(synthetic.seeded_state_dict(model.state_dict(), 2021))
    model = model.to(dev).train()
    batch = synthetic.make_batch(2021 + rank, ragged=a.ragged, device=dev, **cfg)
    lengths = batch["lengths"]
    n_utt = sum(lengths)
    label = train.flatten_labels(batch["label"], lengths)
    loss_f = FocalLoss(gamma=0.5)
    # sum-reduce with the 1/world factor folded into the loss scale below: no separate averaging kernel after the all-reduce
    dp = distributed.GradientBucket(model, average=False) if world > 1 else None
    total_utt = distributed.all_reduce_scalar(n_utt) if world > 1 else n_utt

    scale = (n_utt / total_utt) if dp is not None else 1.0   # local mean -> this rank's share of the GLOBAL mean

    def fwd_bwd():
        logp = model(batch["textf"], batch["qmask"], batch["umask"], lengths, batch["acouf"], batch["visuf"])[0]
        loss = loss_f(logp, label)
        if dp is not None:
            loss = loss * scale        # the summed bucket is then the gradient of the mean over ALL ranks' utterances
        loss.backward()
        return loss

    ops.set_async_weight_grads("deferred" if a.deferred_wgrad else a.async_wgrad)

    def eager_step():
        model.zero_grad(set_to_none=True)
        loss = fwd_bwd()
        ops.join_weight_grads()
        if dp is not None:
            dp.all_reduce()
        return loss

    step = eager_step
    launch_mode = "eager"
    if not a.no_graph:
        from mm_dfn_amd.graphs import CapturedStep
        try:
            captured = CapturedStep(model, fwd_bwd, warmup=3, bucket=dp)

            def step():
                loss = captured.replay()
                if dp is not None:
                    dp.reduce_flat()
                return loss
            launch_mode = "hipGraph replay of the whole step"
        except Exception as exc:  # capture is an optimisation; never lose the measurement over it
            print("[bench] hipGraph capture failed (%s: %s); running eagerly" % (type(exc).__name__, exc), file=sys.stderr)
            torch.cuda.synchronize()
            step = eager_step

    for _ in range(a.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # ---- the same step followed by the fused Adam update (one extra launch), reported next to the headline.
    # Parameters and gradients are flat buffers here, so the step is re-captured against the flat storage.
    adam_ms = None
    if world == 1 and not a.no_graph and launch_mode != "eager":
        try:
            from mm_dfn_amd.graphs import CapturedStep
            from mm_dfn_amd.optim import FlatAdam
            model.zero_grad(set_to_none=True)
            fwd_bwd()
            opt = FlatAdam(model, lr=3e-4, weight_decay=1e-4)
            opt.bucket.flatten()
            opt._materialise()
            cap2 = CapturedStep(model, fwd_bwd, warmup=2, bucket=opt.bucket)
            for _ in range(3):
                cap2.replay()
                opt.step(grads_already_flat=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                cap2.replay()
                opt.step(grads_already_flat=True)
            torch.cuda.synchronize()
            adam_ms = (time.perf_counter() - t1) / a.steps * 1e3
        except Exception as exc:
            print("[bench] fused-Adam leg skipped: %s" % exc, file=sys.stderr)

    if rank == 0:
        # ---- roofline of the dominant graph kernel at this workload (K6 propagate forward, d = 100)
        feats = torch.randn(3, n_utt, 200, device=dev)
        d = 100

        def mk():
            adj = ops.build_adjacency(feats, lengths)
            return adj, torch.randn(3 * n_utt, d, device=dev)

        ms = time_propagate(mk, d)
        lay = ops.DialogueLayout.get(lengths, 3, dev)
        alg_bytes = lay.propagate_bytes(d)
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        out = {
            "metric": "utterances/sec (fwd+bwd), IEMOCAP-shaped batch", "value": total_utt * a.steps / dt,
            "unit": "utterances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "launch": launch_mode,
            "config": {"workload": "%s: B=%d dialogues/GPU, L=%s, dims %d/%d/%d, %d GCN layers, P=%d, dropout %.2f"
                                   % (a.config, cfg["B"], "ragged<=%d" % cfg["L"] if a.ragged else cfg["L"], cfg["D_t"],
                                      cfg["D_a"], cfg["D_v"], cfg["nlayers"], cfg["P"], a.dropout),
                       "utterances_per_gpu": n_utt, "parallelism": "dp%d" % world},
            "with_fused_adam_step": None if adam_ms is None else {"ms_per_step": adam_ms,
                                                                   "value": total_utt / (adam_ms * 1e-3)},
            "roofline": {"bound": "hbm", "kernel": "propagate_v2_kernel<2,4,2,16,1> (K6 fwd, d=100, exact-f32 MFMA)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes": alg_bytes, "avg_launch_us": ms * 1e3,
                         "traffic": measured_traffic("cfg2") if (a.config == "cfg2" and not a.ragged) else None,
                         "traffic_source": "profiles/r01_propagate_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                           "separate passes)"},
        }
        # the same kernel on BASELINE config 5 (L=512, M=6, d=100, 32 dialogues), where one launch moves 282 MB and
        # the launch-latency floor no longer hides the kernel (the >=40 % HBM target is a cfg5 property)
        try:
            l5 = [512] * 32
            f5 = torch.randn(6, sum(l5), 200, device=dev)

            def mk5():
                adj = ops.build_adjacency(f5, l5)
                return adj, torch.randn(6 * sum(l5), d, device=dev)

            ms5 = time_propagate(mk5, d, iters=20, warm_replays=15, timed_replays=5)
            lay5 = ops.DialogueLayout.get(l5, 6, dev)
            b5 = lay5.propagate_bytes(d)
            out["roofline_cfg5"] = {"workload": "cfg5: B=32, L=512, M=6, d=100", "bound": "hbm",
                                    "kernel": "propagate_split_kernel (K6 fwd, bf16-piece MFMA, fp32-level error)",
                                    "traffic": measured_traffic("cfg5_b32"),
                                    "achieved": b5 / (ms5 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": b5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": b5,
                                    "avg_launch_us": ms5 * 1e3,
                                    "useful_tflops": lay5.propagate_flops(d) / (ms5 * 1e-3) / 1e12}
            # K6 backward at the same workload, reported separately: dH = A^T dO (the forward kernel, A is
            # symmetric) + dA = dO . H^T on the tile pattern (tile_dot + cross_dot); bytes_bwd = 8 nnz + 16 M N d
            adj5, H5 = mk5()
            dO5 = torch.randn_like(H5)
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
            del f5, adj5, H5, dO5, gb
        except Exception as exc:
            print("[bench] cfg5 roofline leg skipped: %s" % exc, file=sys.stderr)
        if world == 1 and not a.no_extra:
            out["other_workloads"] = []
            for cname, rag in (("cfg2", True), ("cfg3", True), ("cfg4", False), ("cfg4", True)):
                try:
                    out["other_workloads"].append(quick_leg(cname, rag, a.dropout))
                except Exception as exc:
                    print("[bench] extra workload %s skipped: %s" % (cname, exc), file=sys.stderr)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, batch, model.state_dict(), a.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
