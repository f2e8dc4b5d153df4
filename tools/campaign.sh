set -e
python tools/stress_split.py 400 101 | tail -1
python tools/stress_split.py 400 102 | tail -1
python - <<'PY'
import subprocess, sys
# random ragged model parity over more seeds than the committed test
import os
os.environ["PYTHONPATH"] = os.getcwd() + "/tests:" + os.getcwd() + "/oracle"
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import test_edge_cases_gpu as T
bad = 0
for seed in range(6, 46):
    try:
        T.test_random_ragged_batches_against_oracle(seed)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAILED", str(e)[:200])
print("random model parity: 40 seeds,", bad, "failures")
'''
print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout[-1500:])
PY
