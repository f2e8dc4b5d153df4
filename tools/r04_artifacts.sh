#!/bin/bash
# Round-4 measurement campaign on one MI355X box: bench line, rocprof kernel stats of the cfg2 / cfg3 / cfg4 / cfg2_refdims / cfg5
# steps and of the K6 roofline legs, step breakdowns, PMC traffic of the roofline legs.  Outputs under gpurun_out/r04/ (copied to
# profiles/ by hand).    usage: tools/r04_artifacts.sh <commit>
out=gpurun_out/r04; mkdir -p $out
python bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err
TAILN=2 TOPN=0 tools/prof_stats.sh r04_bench_cfg2 python bench.py --no-extra --no-roofline --no-cpu-baseline
TAILN=2 TOPN=0 tools/prof_stats.sh r04_bench_cfg3 python bench.py --config cfg3 --ragged --no-extra --no-roofline --no-cpu-baseline --steps 30 --warmup 10
TAILN=2 TOPN=0 tools/prof_stats.sh r04_bench_cfg4 python bench.py --config cfg4 --no-extra --no-roofline --no-cpu-baseline --steps 30 --warmup 10
TAILN=2 TOPN=0 tools/prof_stats.sh r04_bench_cfg2_refdims python bench.py --config cfg2_refdims --no-extra --no-roofline --no-cpu-baseline --steps 30 --warmup 10
TAILN=2 TOPN=0 tools/prof_stats.sh r04_cfg5_stream_b8 python tools/run_stream_step.py cfg5 10
TAILN=2 TOPN=0 tools/prof_stats.sh r04_cfg5_stream_b32 python tools/run_stream_step.py cfg5_b32 6
TAILN=2 TOPN=0 tools/prof_stats.sh r04_k6_roofline_legs python bench.py --only-roofline
mv gpurun_out/r04_*_kernel_stats.csv $out/ 2>/dev/null
python tools/step_breakdown.py $out/r04_bench_cfg2_kernel_stats.csv $out/r04_step_breakdown_cfg2.json cfg2 110 > /dev/null
python tools/step_breakdown.py $out/r04_bench_cfg3_kernel_stats.csv $out/r04_step_breakdown_cfg3.json cfg3 33 > /dev/null
python tools/step_breakdown.py $out/r04_bench_cfg4_kernel_stats.csv $out/r04_step_breakdown_cfg4.json cfg4 110 > /dev/null
python tools/collect_traffic.py ${1:-unknown} > $out/traffic.log 2>&1
cp gpurun_out/r04_propagate_traffic.json $out/ 2>/dev/null
tail -c 600 $out/bench_cfg2.json
