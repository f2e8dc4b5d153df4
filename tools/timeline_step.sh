# kernel timeline of one captured step:  bash tools/timeline_step.sh "<bench args>" <tag>   (env MMDFN_* passed through)
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp; export TMPDIR=/tmp
out=/tmp/tl_$2
rm -rf $out
rocprofv3 --kernel-trace -d $out -o p --output-format csv -- python $root/bench.py $1 --no-extra --no-roofline --no-cpu-baseline --steps 30 --warmup 20 > $out.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
if [ -z "$f" ]; then tail -20 $out.log; exit 1; fi
python $root/tools/step_timeline.py $f > $root/gpurun_out/timeline_$2.txt
tail -1 $root/gpurun_out/timeline_$2.txt
