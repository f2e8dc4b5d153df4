"""HBM traffic of the K6 roofline legs: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE passes over
`bench.py --only-roofline` (kernel-trace only), FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64,
MI355X_MICROARCH.md), written to profiles/r05_propagate_traffic.json together with the commit the library was built from.

    COMMIT=$(git rev-parse --short HEAD) gpurun -- 'python tools/collect_traffic.py <commit>'
"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1] if len(sys.argv) > 1 else "unknown"
out_dir = os.path.join(ROOT, "gpurun_out", "traffic")
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out_dir, ctr)
    cmd = ["timeout", "240", "rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--only-roofline"]
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            for name in ("propagate_split_kernel", "propagate_v2_kernel", "tile_dot_split_kernel", "cross_dot_kernel"):
                if name in k:
                    # grid size separates the cfg2 / cfg5 / cfg5-d512 legs of the same kernel family
                    vals.setdefault((name, row.get("Grid_Size", "?"), ctr), []).append(float(row["Counter_Value"]))
res = {"commit": commit, "protocol": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --only-roofline` "
       "(rotating buffer sets); FETCH_SIZE x 2 (gfx950), KB -> bytes", "raw": {}}
per = {}
for (name, grid, ctr), v in sorted(vals.items()):
    res["raw"]["%s grid=%s %s" % (name, grid, ctr)] = {"launches": len(v), "avg_kb": sum(v) / len(v)}
    per.setdefault((name, grid), {})[ctr] = sum(v) / len(v)
legs = []
for (name, grid), c in per.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        legs.append((name, grid, 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024, c))
for name, grid, tb, c in legs:
    res.setdefault("legs", []).append({"kernel": name, "grid": grid, "traffic_bytes": tb, "fetch_kb_raw": c["FETCH_SIZE"],
                                       "write_kb": c["WRITE_SIZE"]})
# the two figures bench.py attaches: cfg2 (propagate_v2, 6.59 MB algorithmic) and cfg5 B=32 (propagate_split, 281.9 MB)
def pick(name, alg):
    c = [l for l in res.get("legs", []) if l["kernel"] == name]
    return min(c, key=lambda l: abs(l["traffic_bytes"] / alg - 1.0)) if c else None
for key, name, alg in (("cfg2", "propagate_v2_kernel", 6589440), ("cfg5_b32", "propagate_split_kernel", 281935872)):
    # (the stack-backward leg launches the same kernel with the same grid on column blocks of a wide buffer: its launches are
    # averaged into the same (kernel, grid) bucket; the figure is then an upper bound for the forward leg's contiguous operands)
    l = pick(name, alg)
    if l:
        res[key] = {"kernel": name, "traffic_bytes": l["traffic_bytes"], "algorithmic_bytes": alg, "ratio": l["traffic_bytes"] / alg}
with open(os.path.join(ROOT, "gpurun_out", "r05_propagate_traffic.json"), "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps({k: res[k] for k in res if k in ("cfg2", "cfg5_b32", "commit")}, indent=1))
