"""HBM traffic of the K6 roofline legs: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE passes over
`bench.py --only-roofline` (kernel-trace only), FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64,
MI355X_MICROARCH.md), written to profiles/r06_propagate_traffic.json together with the commit the library was built from.

    COMMIT=$(git rev-parse --short HEAD) gpurun -- 'python tools/collect_traffic.py <commit>'
"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1] if len(sys.argv) > 1 else "unknown"
out_dir = os.path.join(ROOT, "gpurun_out", "traffic")
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
vals = {}
stack = {}      # the stack-backward leg (its launches share kernel names and grids with the per-call legs: own passes)
NAMES = ("propagate_split_kernel", "propagate_v2_kernel", "tile_dot_split_kernel", "cross_dot_kernel")
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for legs in ("fwd,bwd,d512", "stack"):
        d = os.path.join(out_dir, ctr + "_" + legs.replace(",", "_"))
        cmd = ["timeout", "240", "rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--only-roofline", "--roofline-legs", legs]
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                for name in NAMES:
                    if name in k:
                        if legs == "stack":
                            stack.setdefault((name, row.get("Grid_Size", "?"), ctr), []).append(float(row["Counter_Value"]))
                        else:
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
# stack leg: one iteration = 8 dH launches (propagate_split on column blocks) + ONE dA over all layers (tile_dot_split of
# width 800 + its cross_dot pieces).  The "stack"-only run also times the forward leg (24+ contiguous propagate launches of
# the same grid, needed for the set-up): the column-block launches are the LAST 8 x iterations of that (kernel, grid) list,
# iterations = the launches of the widest tile_dot_split grid / 1.
def stack_traffic():
    td = {}
    for (name, grid, ctr), v in stack.items():
        if name == "tile_dot_split_kernel":
            td.setdefault(grid, {})[ctr] = v
    if not td:
        return None
    # the d = 800 contraction is the tile_dot_split launch with the largest traffic per launch
    grid = max(td, key=lambda g: sum(td[g].get("FETCH_SIZE", [0])) / max(1, len(td[g].get("FETCH_SIZE", [1]))))
    iters = len(td[grid].get("FETCH_SIZE", []))
    if not iters or len(td[grid].get("WRITE_SIZE", [])) != iters:
        return None
    tot = 0.0
    detail = {}
    for (name, g, ctr), v in stack.items():
        mult = 2048.0 if ctr == "FETCH_SIZE" else 1024.0
        if name == "tile_dot_split_kernel" and g == grid:
            use = v
        elif name == "cross_dot_kernel":
            use = v[-(len(v) // iters) * iters:] if len(v) >= iters else []
            # (cross_dot launches of the set-up adjacency build, if any, come first)
        elif name == "propagate_split_kernel":
            use = v[-8 * iters:]
        else:
            continue
        tot += mult * sum(use)
        detail["%s grid=%s %s" % (name, g, ctr)] = {"launches_counted": len(use), "avg_kb": sum(use) / max(1, len(use))}
    return {"iterations": iters, "traffic_bytes": tot / iters, "detail": detail}


st = stack_traffic()
if st:
    res["cfg5_b32_bwd_stack"] = {"kernel": "tile_dot_split_kernel", "traffic_bytes": st["traffic_bytes"], "iterations": st["iterations"],
                                 "detail": st["detail"]}
with open(os.path.join(ROOT, "gpurun_out", "r06_propagate_traffic.json"), "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps({k: res[k] for k in res if k in ("cfg2", "cfg5_b32", "cfg5_b32_bwd_stack", "commit")}, indent=1))
