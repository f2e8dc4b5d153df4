# Round evidence in one go (run on the GPU box):  bash tools/collect_round_profiles.sh r06
#   gpurun_out/<r>_bench_cfg2.json                      the unmodified `python bench.py` line
#   gpurun_out/<r>_bench_{cfg2,cfg3,cfg4}_kernel_stats.csv + <r>_step_breakdown_*.json   rocprofv3 --kernel-trace --stats of one workload each
#   gpurun_out/<r>_cfg5_stream_{b8,b32}_kernel_stats.csv
#   gpurun_out/<r>_k6_roofline_legs_kernel_stats.csv
r=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp; export TMPDIR=/tmp
python $root/bench.py > $root/gpurun_out/${r}_bench_cfg2.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
prof() {   # tag, command...
  tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats -d /tmp/p_$tag -o p --output-format csv -- "$@" > /tmp/p_$tag.log 2>&1
  f=$(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ]; then echo "no stats for $tag"; tail -5 /tmp/p_$tag.log; return; fi
  cp $f $root/gpurun_out/${r}_${tag}_kernel_stats.csv
}
prof bench_cfg2 python $root/bench.py --config cfg2 --no-extra --no-roofline --no-cpu-baseline --no-floor --steps 200 --warmup 30
prof bench_cfg3 python $root/bench.py --config cfg3 --ragged --no-extra --no-roofline --no-cpu-baseline --no-floor --steps 60 --warmup 15
prof bench_cfg4 python $root/bench.py --config cfg4 --no-extra --no-roofline --no-cpu-baseline --no-floor --steps 60 --warmup 15
prof cfg5_stream_b8 python $root/tools/prof_cfg5.py cfg5 20
prof cfg5_stream_b32 python $root/tools/prof_cfg5.py cfg5_b32 8
prof k6_roofline_legs python $root/bench.py --only-roofline
cd $root
python tools/step_breakdown.py gpurun_out/${r}_bench_cfg2_kernel_stats.csv gpurun_out/${r}_step_breakdown_cfg2.json cfg2 110
python tools/step_breakdown.py gpurun_out/${r}_bench_cfg3_kernel_stats.csv gpurun_out/${r}_step_breakdown_cfg3.json cfg3 33
python tools/step_breakdown.py gpurun_out/${r}_bench_cfg4_kernel_stats.csv gpurun_out/${r}_step_breakdown_cfg4.json cfg4 110
python - <<PY
import json
d = json.load(open("gpurun_out/${r}_bench_cfg2.json"))
print("headline", d["value"], d["ms_per_step"], "roofline_cfg5", d.get("roofline_cfg5", {}).get("frac"), d.get("roofline_cfg5", {}).get("avg_launch_us"))
for w in d.get("other_workloads", []):
    print(" ", w.get("workload", "")[:70], w.get("ms_per_step"))
for c in ("cfg2", "cfg3", "cfg4"):
    b = json.load(open("gpurun_out/${r}_step_breakdown_%s.json" % c))
    print(c, b["kernel_us_per_step"], b["launches_per_step"], {k: v["us_per_step"] for k, v in list(b["families"].items())[:4]})
PY
