"""Root-cause hunt for the RCCL teardown abort: world size 1, backend "nccl", in ONE process,
N times: init_process_group -> model -> CapturedStep(reduce_in_graph=True) -> replays -> teardown variant -> destroy_process_group.

  --variant keep     the graph objects are still alive when the group is destroyed (what aborted in round 5)
  --variant drop     `del` the captured step, gc, synchronize, then destroy (the order tests/bench used)
  --variant close    CapturedStep.close() (graph.reset() + pool release), synchronize, destroy
  --variant eager    no collective inside the graph (eager all-reduce after each replay), drop, destroy
The driver mode (no --variant) runs every variant in a child process and reports how far each one got.
"""
import argparse
import gc
import os
import socket
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one_round(variant, it):
    import torch
    import torch.distributed as dist
    from mm_dfn_amd import FocalLoss, distributed, synthetic
    from mm_dfn_amd import train as T
    from mm_dfn_amd.graphs import CapturedStep
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    distributed.init(backend="nccl")
    cfg = dict(P=2, C=6, nlayers=2, D_t=100, D_a=100, D_v=512)
    b = synthetic.make_batch(31 + it, lengths=[12, 5, 9], device="cuda", B=3, L=12, **cfg)
    label = T.flatten_labels(b["label"], b["lengths"])
    loss_f = FocalLoss(gamma=0.5)
    m = synthetic.build_model(dropout=0.0, **cfg)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 33))
    m = m.cuda().train()

    def fb():
        lp = m(b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])[0]
        loss = loss_f(lp, label)
        loss.backward()
        return loss

    bucket = distributed.GradientBucket(m, average=True)
    m.zero_grad(set_to_none=True)
    fb()
    bucket.flatten()
    in_graph = variant != "eager"
    cap = CapturedStep(m, fb, warmup=1, bucket=bucket, reduce_in_graph=in_graph)
    for _ in range(3):
        loss = cap.replay()
        if not in_graph:
            bucket.reduce_flat()
    val = float(loss)
    torch.cuda.synchronize()
    if variant == "keep":
        dist.destroy_process_group()
        del cap
    elif variant in ("drop", "eager"):
        del cap
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    elif variant == "close":
        cap.close()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        del cap
    else:
        raise SystemExit("unknown variant " + variant)
    del bucket, m
    gc.collect()
    torch.cuda.empty_cache()
    return val


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default=None)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    if a.variant:
        for it in range(a.iters):
            v = one_round(a.variant, it)
            print("ROUND-OK %s %d loss %.6f" % (a.variant, it, v), flush=True)
        print("VARIANT-DONE %s" % a.variant, flush=True)
        sys.exit(0)
    for variant in ("keep", "drop", "close", "eager"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--variant", variant, "--iters", str(a.iters)],
                           capture_output=True, text=True, timeout=1500)
        ok = p.stdout.count("ROUND-OK")
        print("variant %-6s rc %4d  rounds completed %2d / %d  done=%s" % (variant, p.returncode, ok, a.iters,
                                                                          "VARIANT-DONE" in p.stdout), flush=True)
        if p.returncode != 0 or ok < a.iters:
            print("  --- stderr tail ---\n" + "\n".join(p.stderr.strip().splitlines()[-25:]), flush=True)
