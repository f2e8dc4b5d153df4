"""Kernel timeline of ONE captured step from a rocprofv3 --kernel-trace CSV: start offset, duration and stream overlap of
every launch of the last replay.  usage: step_timeline.py <kernel_trace.csv> [n_last_kernels_per_step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay = the kernels after the last long gap (> 50 us) ... simpler: take the last K kernels where K is the
# number of kernels between the last two occurrences of the focal-loss forward kernel
idx = [i for i, r in enumerate(rows) if "focal_loss_fwd" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
# a step runs from its first kernel to the kernel before the next step's first kernel: align on the first kernel after focal bwd... use
# the span [a, b): rotated, but complete
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
end_prev = 0
def short(n):
    for p in ("void ", "(anonymous namespace)::", "at::native::"):
        n = n.replace(p, "")
    return n.split("(")[0][:58]
tot = 0
for r in step:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    ov = "||" if s < end_prev else "  "
    print("%8.1f %7.1f %s %s" % (s / 1e3, (e - s) / 1e3, ov, short(r["Kernel_Name"])))
    end_prev = max(end_prev, e)
    tot += e - s
print("span %.1f us, kernel time %.1f us, %d launches" % ((end_prev) / 1e3, tot / 1e3, len(step)))
