"""Per-step comparison of rocprofv3 kernel-stats CSVs (steps = gru_seq_fwd launches / 2):  python tools/cmp_stats.py a.csv b.csv [substr ...]"""
import csv, sys
files = [a for a in sys.argv[1:] if a.endswith(".csv")]
keys = [a for a in sys.argv[1:] if not a.endswith(".csv")]
for f in files:
    rows = list(csv.DictReader(open(f)))
    steps = max(1, sum(int(r["Calls"]) for r in rows if "gru_seq_fwd" in r["Name"]) // 2)
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
    n = sum(int(r["Calls"]) for r in rows) / steps
    print("%s: steps %d, %.1f us/step in %.1f launches" % (f, steps, tot, n))
    for r in rows:
        if not keys or any(k in r["Name"] for k in keys):
            print("   %-70s %5.2f/step avg %7.1f us  %7.1f us/step" % (r["Name"][:70], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                                     float(r["TotalDurationNs"]) / steps / 1e3))
