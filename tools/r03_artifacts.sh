#!/bin/bash
# Round-3 measurement campaign on one MI355X box: bench line, rocprof kernel stats of the cfg2 / cfg3 / cfg4 / cfg5 steps and
# of the K6 roofline legs, the few-row projection table.  Outputs under gpurun_out/r03/ (copied to profiles/ by hand).
out=gpurun_out/r03; mkdir -p $out
if [ ! -s $out/bench_cfg2.json ]; then python bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err; fi
TAILN=2 TOPN=0 tools/prof_stats.sh r03_bench_cfg2 python bench.py --no-extra --no-roofline --no-cpu-baseline
TAILN=2 TOPN=0 tools/prof_stats.sh r03_bench_cfg3 python bench.py --config cfg3 --ragged --no-extra --no-roofline --no-cpu-baseline --steps 30 --warmup 10
TAILN=2 TOPN=0 tools/prof_stats.sh r03_bench_cfg4 python bench.py --config cfg4 --no-extra --no-roofline --no-cpu-baseline --steps 30 --warmup 10
TAILN=2 TOPN=0 tools/prof_stats.sh r03_cfg5_stream_b8 python tools/run_stream_step.py cfg5 10
TAILN=2 TOPN=0 tools/prof_stats.sh r03_k6_roofline_legs python bench.py --only-roofline
python tools/bench_linear_group.py 2>&1 | grep -v amdgpu > $out/linear_group_vs_hipblaslt.txt
mv gpurun_out/r03_*_kernel_stats.csv $out/ 2>/dev/null
for c in cfg2 cfg3 cfg4; do python tools/step_breakdown.py $out/r03_bench_${c}_kernel_stats.csv $out/r03_step_breakdown_$c.json $c > /dev/null; done
tail -c 400 $out/bench_cfg2.json
