#!/bin/bash
# usage: tools/prof_stats.sh <tag> <command...>   -> gpurun_out/<tag>_kernel_stats.csv (+ printed top rows)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" > /tmp/prof_$tag.log 2>&1
mkdir -p gpurun_out
cp /tmp/prof_$tag/p_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
grep -v amdgpu.ids /tmp/prof_$tag.log | grep -v "^W2026\|^E2026\|^I2026" | tail -${TAILN:-15}
python - <<PY
import csv
for r in list(csv.DictReader(open('gpurun_out/${tag}_kernel_stats.csv')))[:${TOPN:-14}]:
    print("%-100s calls %5s avg %9.2f us  %5s%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
PY
