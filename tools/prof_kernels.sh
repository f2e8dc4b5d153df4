# per-kernel averages of one bench.py workload (run on the GPU box):  bash tools/prof_kernels.sh <pattern> [bench args...]
pat=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p_k
rocprofv3 --kernel-trace --stats -d /tmp/p_k -o p --output-format csv -- python $root/bench.py --no-extra --no-roofline --no-cpu-baseline --no-floor "$@" > /tmp/p_k.log 2>&1
f=$(find /tmp/p_k -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then echo "no stats"; tail -5 /tmp/p_k.log; exit 1; fi
python - "$f" "$pat" <<PY
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print("%-90s calls %6s  avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
