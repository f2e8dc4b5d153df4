#!/bin/bash
# usage: tools/pmc.sh <outdir under gpurun_out> <kernel-name regex> -- <command...>
# One rocprofv3 --pmc pass per counter group (separate passes, kernel-trace only), then the per-kernel averages.
out=$1; shift; kre=$1; shift; shift
root=$(pwd)
mkdir -p $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
# groups come from $PMC_GROUPS (';'-separated) or the default list; every pass runs under its own `timeout` (a pass that
# names a counter the tool cannot schedule aborts and then hangs)
if [ -n "$PMC_GROUPS" ]; then IFS=';' read -ra groups <<< "$PMC_GROUPS"; else
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
fi
i=0
for g in "${groups[@]}"; do
  timeout ${PMC_PASS_TIMEOUT:-120} rocprofv3 --pmc $g --kernel-trace -d $root/gpurun_out/$out/p$i -o p --output-format csv -- "$@" > $root/gpurun_out/$out/p$i.log 2>&1
  i=$((i+1))
done
cd $root
python - "$out" "$kre" <<'PY'
import csv, glob, re, sys, collections
out, kre = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/%s/p*/**/*counter_collection.csv" % out, recursive=True):
    for row in csv.DictReader(open(f)):
        if kre.search(row.get("Kernel_Name", "")):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/%s/summary.txt" % out, "w") as fh:
    for k in sorted(acc):
        v = acc[k]
        line = "%-40s n=%3d avg=%.6g" % (k, len(v), sum(v) / len(v))
        print(line); fh.write(line + "\n")
PY
