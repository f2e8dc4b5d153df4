"""Where the time of the K5s strip kernels goes: the tuning build's early-return switch MMDFN_ADJ_STOP=k (timing only; results are
incomplete) on the cfg2 shape.  Forward k: 1 unit rows in LDS, 2 + cross-modal part, 3 + Gram / similarity / partial sums, 0 all.
k = 9: nothing but the launch and the block decode.  Backward k: 1 Z sums + strip capture, 2 + d(degree), 3 + E strip, 4 + MFMAs, 0 all."""
import os, sys
os.environ["MMDFN_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("bas", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_adj_small.py"))
src = open(spec.origin).read().split("for name, lengths, M, D in SHAPES:")[0]
exec(compile(src, spec.origin, "exec"))
B = int(os.environ.get("ADJ_B", "16"))
lengths, M, D = [110] * B, 3, 200
N = sum(lengths)
lay = DialogueLayout.get(lengths, M, torch.device(dev))
gen = torch.Generator(device=dev).manual_seed(5)
bufs = [buffers(lay, M, N, D, gen) for _ in range(NSET)]
os.environ["MMDFN_ADJ_SMALL"] = "1"
for sr in os.environ.get("ADJ_SRS", "32,64").split(","):
    os.environ["MMDFN_ADJ_SR"] = sr
    for rep in range(2):
        out = []
        for stop in (9, 1, 2, 3, 0):
            os.environ["MMDFN_ADJ_STOP"] = str(stop)
            out.append("%d: %5.1f" % (stop, timed(lambda k=0: run_fwd(bufs[k], lay, M, N, D))))
        print("SR %s forward  (incl. finish launch) stop-> us  " % sr + "   ".join(out), flush=True)
        os.environ["MMDFN_ADJ_STOP"] = "0"
        for k in range(NSET):
            run_fwd(bufs[k], lay, M, N, D)
        out = []
        for stop in (9, 1, 2, 3, 4, 0):
            os.environ["MMDFN_ADJ_STOP"] = str(stop)
            out.append("%d: %5.1f" % (stop, timed(lambda k=0: run_bwd(bufs[k], lay, M, N, D))))
        print("SR %s backward stop-> us  " % sr + "   ".join(out), flush=True)
