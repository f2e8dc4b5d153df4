"""One workload = the cfg5 module stack (MultiStreamGraphModel) stepped as a captured graph: for rocprofv3 --kernel-trace --stats.
    python tools/prof_cfg5.py [cfg5|cfg5_b32] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_b32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print(bench.cfg5_leg(name, 0.5, steps, 5))
