"""Times the REAL reference (imported from /root/reference through ref_shim) on this container's CPU.

TEST/MEASUREMENT INFRASTRUCTURE ONLY - runs only where /root/reference exists (the build container); the
result is committed as profiles/r01_reference_cpu_build_container.json so the bench's `cpu_baseline`
(kind "port", timed on the GPU box's host) can be read next to the true reference's own number.

    python oracle/time_reference_cpu.py [--threads 8] [--steps 5]

One step = zero_grad -> forward -> FocalLoss(gamma=.5) -> backward (train mode, dropout .5), the bench's
definition; batches come from mm_dfn_amd/synthetic.py exactly as in bench.py.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from mm_dfn_amd import synthetic  # noqa: E402


def time_case(name, cfg, lengths, steps):
    _, _, _, ref_loss = ref_shim.modules()
    m = ref_shim.build_reference_model(cfg["D_t"], cfg["D_a"], cfg["D_v"], cfg["P"], cfg["C"], cfg["nlayers"],
                                       dropout=0.5)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 7))
    m.train()
    b = synthetic.make_batch(2021, lengths=lengths, **cfg)
    loss_f = ref_loss.FocalLoss(gamma=0.5)
    n_utt = int(sum(b["lengths"]))
    label = torch.cat([b["label"][j, :n] for j, n in enumerate(b["lengths"])])
    args = (b["textf"], b["qmask"], b["umask"], b["lengths"], b["acouf"], b["visuf"])

    def step():
        m.zero_grad()
        logp = m(*args)[0]
        loss_f(logp, label).backward()

    step()
    ts = []
    for _ in range(steps):
        t0 = time.time()
        step()
        ts.append(time.time() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"case": name, "B": len(b["lengths"]), "N": n_utt, "nlayers": cfg["nlayers"], "P": cfg["P"],
            "s_per_step_median": round(med, 4), "utterances_per_s": round(n_utt / med, 1), "steps": steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_reference_cpu_build_container.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    import numpy as np
    rs = np.random.RandomState(5)
    cases = [
        ("cfg1_b1_L110", dict(synthetic.CONFIGS["cfg1"]), None),
        ("cfg2_b16_L110_fixed", dict(synthetic.CONFIGS["cfg2"]), None),
        ("cfg2_b16_ragged_27_110", dict(synthetic.CONFIGS["cfg2"]), [110] + [int(x) for x in rs.randint(27, 111, 15)]),
        ("cfg3_b32_meld_like", dict(synthetic.CONFIGS["cfg3"]), [33] + [int(x) for x in rs.randint(3, 34, 31)]),
    ]
    res = []
    for name, cfg, lengths in cases:
        if lengths is None:
            lengths = [cfg["L"]] * cfg["B"]
        r = time_case(name, cfg, lengths, a.steps)
        print(r, flush=True)
        res.append(r)
    out = {"what": "real reference (zerohd4869/MM-DFN code/, imported unmodified via oracle/ref_shim.py) fwd+FocalLoss+bwd, "
                   "train mode, synthetic batches of bench.py",
           "host": "build container, %d vCPU" % (os.cpu_count() or 0), "threads": a.threads,
           "torch": torch.__version__, "cases": res}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
