# rocprofv3 kernel stats of one workload with / without the valid-length GRU launches
# usage: bash tools/prof_gru_truncate.sh "<bench args>" <tag>      (writes gpurun_out/<tag>_<mode>_kernel_stats.csv)
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp; export TMPDIR=/tmp
for tr in 0 auto; do
  out=/tmp/prof_$2_$tr
  rm -rf $out
  MMDFN_GRU_TRUNCATE=$tr rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python $root/bench.py $1 --no-extra --no-roofline --no-cpu-baseline --steps 100 --warmup 30 > $out.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ]; then tail -20 $out.log; continue; fi
  cp $f $root/gpurun_out/$2_${tr}_kernel_stats.csv
  echo "== $2 truncate=$tr"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-90s %6s %10.1f %6.2f%%"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
done
