"""Aggregate the LAST `window_ms` of a rocprofv3 kernel_trace.csv per kernel name (steady state of a
run whose start-up -- MIOpen find / compilation -- would otherwise swamp --stats).
usage: trace_tail_stats.py <kernel_trace.csv> <window_ms> <out.csv>"""
import csv
import sys
from collections import defaultdict

src, window_ms, dst = sys.argv[1], float(sys.argv[2]), sys.argv[3]
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
end = max(r[1] for r in rows)
lo = end - int(window_ms * 1e6)
agg = defaultdict(lambda: [0, 0])
first = None
for s, e, n in rows:
    if s >= lo:
        first = s if first is None else min(first, s)
        agg[n][0] += 1
        agg[n][1] += e - s
span = (end - first) / 1e6
tot = sum(v[1] for v in agg.values())
with open(dst, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "window_ms=%.1f busy_ms=%.1f" % (span, tot / 1e6)])
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n[:300], c, t, "%.1f" % (t / c), "%.2f" % (100.0 * t / tot)])
