"""Same-box A/B of one environment switch on the cfg5 module-stack legs of bench.py (MultiStreamGraphModel, B = 8 and 32):
    python tools/ab_cfg5.py VAR A B        (tuning-build switches need MMDFN_TUNING_LIB=1 in the environment)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, a, b = sys.argv[1:4]
code = ("import sys; sys.path.insert(0, %r); import bench; "
        "r8 = bench.cfg5_leg('cfg5', 0.5, steps=6, warmup=2); r32 = bench.cfg5_leg('cfg5_b32', 0.5, steps=4, warmup=2); "
        "print('B=8 %%.3f ms   B=32 %%.3f ms' %% (r8['ms_per_step'], r32['ms_per_step']))" % ROOT)
for rep in range(2):
    for v in (a, b):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
        print("%s=%s rep%d  %s" % (var, v, rep, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
