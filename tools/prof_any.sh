# rocprofv3 kernel stats of an arbitrary command: bash tools/prof_any.sh <tag> <command...>   -> gpurun_out/<tag>_kernel_stats.csv
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
mkdir -p $root/gpurun_out
cd /tmp; export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- "$@" > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then tail -20 $out.log; exit 1; fi
cp $f $root/gpurun_out/${tag}_kernel_stats.csv
tail -3 $out.log
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%-100s %6s %10.1f %6.2f%%"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
