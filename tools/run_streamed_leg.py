"""The streamed pass-loop leg of bench.py alone (eager / captured-step cache / cache + FlatAdam)."""
import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.streamed_leg("cfg2", 0.5), indent=1))
