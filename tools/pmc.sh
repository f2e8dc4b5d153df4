#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <command...>  -> prints per-kernel counter sums (kernel-trace + pmc only)
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
mkdir -p gpurun_out
python - <<PY
import csv, collections, glob
f = glob.glob('/tmp/pmc_$tag/*counter_collection.csv')
if not f:
    print(open('/tmp/pmc_$tag.log').read()[-2000:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    cnt[(k, r['Counter_Name'])] += 1
with open('gpurun_out/${tag}_pmc.txt', 'w') as out:
    for k, d in agg.items():
        if '${FILTER:-}' and '${FILTER:-}' not in k: continue
        line = k + ' | ' + ' '.join('%s=%.4g' % (c, v / cnt[(k, c)]) for c, v in sorted(d.items()))
        print(line); out.write(line + '\n')
PY
