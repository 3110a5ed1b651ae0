"""Idle time of the GPU in the steady state of a rocprofv3 --kernel-trace run.
usage: trace_gaps.py <kernel_trace.csv> <window_ms> -- prints the share of the last <window_ms> spent idle,
a histogram of the gaps between consecutive kernels, and the largest gaps with the kernels around them."""
import csv
import sys

src, window_ms = sys.argv[1], float(sys.argv[2])
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(src))]
rows.sort()
end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= end - int(window_ms * 1e6)]
busy_until = rows[0][1]
gaps = []
for i in range(1, len(rows)):
    s, e, n = rows[i]
    if s > busy_until:
        gaps.append((s - busy_until, rows[i - 1][2], n, s))
    busy_until = max(busy_until, e)
tot = sum(g[0] for g in gaps)
span = rows[-1][1] - rows[0][0]
print("window %.1f ms, kernels %d, idle %.2f ms (%.1f %%) in %d gaps" % (span / 1e6, len(rows), tot / 1e6, 100.0 * tot / span, len(gaps)))
edges = [1, 2, 3, 5, 10, 20, 50, 100, 1000, 100000]
lo = 0
for hi in edges:
    sel = [g[0] for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
    print("  gaps %6g-%-6g us: %5d  total %8.3f ms" % (lo, hi, len(sel), sum(sel) / 1e6))
    lo = hi
print("largest gaps:")
for g in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f us  after %-60s before %-60s" % (g[0] / 1e3, g[1][:60], g[2][:60]))
