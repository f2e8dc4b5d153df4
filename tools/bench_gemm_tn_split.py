"""The weight-gradient batch of one training step (mmdfn_gemm_tn_batch) on the bf16-piece form (gemm_tn_split.hip) against the
exact-f32 forms of gemm_tn.hip: parity with fp64 and time per launch pair, on the segment lists a step of each BASELINE config
queues (tools/dump_wgrad_segments.py).

    python tools/bench_gemm_tn_split.py [cfg2,cfg3,cfg4,cfg5] [check]
Environment (tuning build): MMDFN_TN_SPLIT=0|1, MMDFN_TNS_WGS=<target workgroups>, MMDFN_TNS_ABL=<ablation bits>."""
import os
import sys

os.environ["MMDFN_TUNING_LIB"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_dfn_amd import ops  # noqa: E402

# (count, rows, M, N, shift, lda, ldb, biases)
SEGS = {
    "cfg2": [(2, 5280, 100, 100, 0, 100, 100, 0), (2, 5280, 400, 100, 0, 400, 100, 2), (1, 5280, 400, 100, 0, 400, 100, 0),
             (2, 5280, 100, 100, 0, 100, 100, 0), (1, 5280, 100, 200, 0, 100, 300, 1),
             (1, 1760, 300, 100, -16, 600, 200, 1), (1, 1760, 300, 100, 16, 600, 200, 1),
             (1, 7040, 300, 100, -64, 600, 200, 1), (1, 7040, 300, 100, 64, 600, 200, 1), (2, 7040, 300, 200, 0, 600, 200, 1),
             (2, 1760, 300, 200, 0, 600, 200, 1), (1, 1760, 300, 100, -16, 600, 200, 1), (1, 1760, 300, 100, 16, 600, 200, 1),
             (1, 7040, 300, 100, -64, 600, 200, 1), (1, 7040, 300, 100, 64, 600, 200, 1), (2, 1760, 300, 200, 0, 600, 200, 1),
             (4, 1760, 300, 200, 0, 600, 200, 0), (1, 1760, 200, 100, 0, 200, 100, 1), (1, 1760, 200, 512, 0, 200, 512, 1),
             (1, 1760, 200, 100, 0, 200, 100, 1)],
    "cfg3": [(2, 3168, 100, 100, 0, 100, 100, 0), (4, 3168, 400, 100, 0, 400, 100, 2), (3, 3168, 400, 100, 0, 400, 100, 0),
             (6, 3168, 100, 100, 0, 100, 100, 0), (1, 3168, 100, 200, 0, 100, 300, 1),
             (1, 1056, 300, 100, -32, 600, 200, 1), (1, 1056, 300, 100, 32, 600, 200, 1),
             (1, 19008, 300, 100, -576, 600, 200, 1), (1, 19008, 300, 100, 576, 600, 200, 1),
             (2, 19008, 300, 200, 0, 600, 200, 1), (2, 1056, 300, 200, 0, 600, 200, 1),
             (1, 1056, 300, 100, -32, 600, 200, 1), (1, 1056, 300, 100, 32, 600, 200, 1),
             (1, 19008, 300, 100, -576, 600, 200, 1), (1, 19008, 300, 100, 576, 600, 200, 1),
             (2, 1056, 300, 200, 0, 600, 200, 1), (4, 1056, 300, 200, 0, 600, 200, 0), (1, 1056, 200, 300, 0, 200, 300, 1),
             (1, 1056, 200, 344, 0, 200, 344, 1), (1, 1056, 200, 600, 0, 200, 600, 1)],
    "cfg4": [(2, 10560, 100, 100, 0, 100, 100, 0), (2, 10560, 400, 100, 0, 400, 100, 2), (1, 10560, 400, 100, 0, 400, 100, 0),
             (2, 10560, 100, 100, 0, 100, 100, 0), (1, 10560, 100, 200, 0, 100, 300, 1),
             (1, 3520, 300, 100, -32, 600, 200, 1), (1, 3520, 300, 100, 32, 600, 200, 1),
             (1, 14080, 300, 100, -128, 600, 200, 1), (1, 14080, 300, 100, 128, 600, 200, 1),
             (2, 14080, 300, 200, 0, 600, 200, 1), (2, 3520, 300, 200, 0, 600, 200, 1),
             (1, 3520, 300, 100, -32, 600, 200, 1), (1, 3520, 300, 100, 32, 600, 200, 1),
             (1, 14080, 300, 100, -128, 600, 200, 1), (1, 14080, 300, 100, 128, 600, 200, 1),
             (2, 3520, 300, 200, 0, 600, 200, 1), (4, 3520, 300, 200, 0, 600, 200, 0), (1, 3520, 200, 100, 0, 200, 100, 1),
             (1, 3520, 200, 512, 0, 200, 512, 1), (1, 3520, 200, 100, 0, 200, 100, 1)],
    "cfg5": [(2, 24576, 100, 100, 0, 100, 100, 0), (8, 24576, 400, 100, 0, 400, 100, 2), (7, 24576, 400, 100, 0, 400, 100, 0),
             (14, 24576, 100, 100, 0, 100, 100, 0), (1, 24576, 100, 200, 0, 100, 300, 1), (6, 4096, 200, 512, 0, 200, 512, 1)],
    # edge cases: ragged rows (not a multiple of 32), shifts larger than a chunk, narrow outputs, two segments into one output
    "edge": [(1, 1000, 100, 100, 0, 100, 100, 1), (1, 333, 300, 200, -7, 600, 200, 1), (1, 333, 300, 200, 45, 600, 200, 1),
             (1, 97, 8, 4, 0, 8, 4, 1), (1, 4097, 132, 116, 1, 132, 116, 0), (2, 777, 400, 100, 0, 400, 100, 2),
             (1, 64, 128, 112, 0, 128, 112, 1), (1, 31, 20, 228, 0, 24, 232, 1)],
}


def make_batch(name, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    batch, flops = [], 0.0
    for (cnt, R, M, N, sh, lda, ldb, nb) in SEGS[name]:
        for _ in range(cnt):
            A = torch.randn(R, lda, device="cuda", generator=g)[:, :M]
            B = torch.randn(R, ldb, device="cuda", generator=g)[:, :N]
            C = torch.empty(M, N, device="cuda")
            cs = [torch.empty(M, device="cuda") for _ in range(nb)]
            batch.append((dict(M=M, N=N), C, cs, 0, [(A, B, sh)]))
            flops += 2.0 * R * M * N
    if name == "edge":       # the two 777-row segments contribute to ONE output
        (o1, C1, cs1, _, s1), (o2, _, _, _, s2) = batch[5], batch[6]
        batch[5] = (o1, C1, cs1, 0, s1 + s2)
        del batch[6]
    return batch, flops


def reference(item):
    o, C, cs, _, segs = item
    ref = torch.zeros(o["M"], o["N"], dtype=torch.float64, device="cuda")
    col = torch.zeros(o["M"], dtype=torch.float64, device="cuda")
    for (A, B, sh) in segs:
        A64, B64 = A.double(), B.double()
        R = A.shape[0]
        Bs = torch.zeros_like(B64)
        if sh >= 0:
            Bs[:R - sh] = B64[sh:]
        else:
            Bs[-sh:] = B64[:R + sh]
        ref += A64.t() @ Bs
        col += A64.sum(0)
    return ref, col


def run(batch):
    for i in range(0, len(batch), 40):
        ops._prepare_wgrad_batch(batch[i:i + 40])(ops._hip.stream())


def check(name):
    batch, _ = make_batch(name)
    run(batch)
    torch.cuda.synchronize()
    worst = 0.0
    for item in batch:
        ref, col = reference(item)
        o, C, cs, _, segs = item
        scale = float(ref.abs().max()) + 1e-30
        e = float((C.double() - ref).abs().max()) / scale
        worst = max(worst, e)
        for b in cs:
            e = float((b.double() - col).abs().max()) / (float(col.abs().max()) + 1e-30)
            worst = max(worst, e)
    return worst


def gtime(fn, iters=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def stamps(name):
    """Mean cycle stamps of waves 0 (cuts first) and 4 (multiplies first) over the workgroups of one launch."""
    batch, _ = make_batch(name)
    trace = torch.zeros(4096 * 32, device="cuda")
    os.environ["MMDFN_TRACE_PTR"] = hex(trace.data_ptr())
    os.environ["MMDFN_TNS_ABL"] = "16"
    os.environ["MMDFN_TN_SPLIT"] = "1"
    run(batch[:40])
    torch.cuda.synchronize()
    os.environ.pop("MMDFN_TNS_ABL")
    t = trace.view(-1, 2, 16).double().cpu()
    live = t[:, 0, 6] > 0
    t = t[live]
    n = t[:, :, 6:7]
    for wv, nm in ((0, "wave 0 (group 0)"), (1, "wave 4 (group 1)")):
        per = (t[:, wv, :5] / n[:, wv]).mean(0)
        print("   %s %-18s per step (cycles): load wait %.0f issue %.0f cut+write %.0f products %.0f barrier %.0f | loop %.0f cycles, "
              "%.1f steps, %d workgroups" % (name, nm, per[0], per[1], per[2], per[3], per[4], t[:, wv, 7].mean(),
                                             n[:, wv].mean(), t.shape[0]))


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "stamps":
        for name in sys.argv[1].split(","):
            stamps(name)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "abl":
        # round 6: what the launch is made of -- timing-only ablations of gemm_tn_split_kernel (MMDFN_TNS_ABL:
        # 1 no cutting, 2 no MFMAs, 4 no fragment reads, 8 no global loads; 7 = loads + LDS staging only, 15 = skeleton)
        os.environ["MMDFN_TN_SPLIT"] = "1"
        for name in sys.argv[1].split(","):
            batch, flops = make_batch(name)
            for rep in range(2):
                row = []
                for abl, what in ((0, "full"), (1, "no cut"), (2, "no MFMA"), (4, "no fragment reads"), (8, "no global loads"),
                                  (7, "loads + staging only"), (15, "skeleton")):
                    if abl:
                        os.environ["MMDFN_TNS_ABL"] = str(abl)
                    else:
                        os.environ.pop("MMDFN_TNS_ABL", None)
                    row.append("%s %.1f" % (what, gtime(lambda: run(batch))))
                os.environ.pop("MMDFN_TNS_ABL", None)
                print("%-5s (%.2f GFLOP) us per launch pair: %s" % (name, flops / 1e9, " | ".join(row)), flush=True)
        return
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2", "cfg3", "cfg4", "cfg5"]
    do_check = len(sys.argv) > 2
    for name in names:
        line = "%-5s" % name
        for form in ("0", "1"):
            os.environ["MMDFN_TN_SPLIT"] = form
            if do_check:
                line += "  split=%s max rel err %.2e" % (form, check(name))
            if name != "edge":
                batch, flops = make_batch(name)
                t = gtime(lambda: run(batch))
                line += "  split=%s %7.1f us %6.1f TFLOP/s" % (form, t, flops / t / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
