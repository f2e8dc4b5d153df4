"""The streamed leg of bench.py alone (eager / exact-signature cache / FlatAdam / bucketed cache on unseen length tuples):
    python tools/run_streamed_leg.py [nbatches]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
print(json.dumps(bench.streamed_leg("cfg2", 0.5, int(sys.argv[1]) if len(sys.argv) > 1 else 32), indent=1))
