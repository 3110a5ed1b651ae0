"""Two-stream balance of the encoder backward pass from a rocprofv3 --kernel-trace CSV: per hardware queue, busy time
and first / last kernel inside the last train step's backward window (from the first bn_bwd_reduce to the clamp_adam).
usage: trace_streams.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    rd = csv.DictReader(fh)
    cols = rd.fieldnames
    qcol = "Queue_Id" if "Queue_Id" in cols else ("Stream_Id" if "Stream_Id" in cols else None)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qcol, "0") if qcol else "0"))
rows.sort()
adam = [i for i, r in enumerate(rows) if "clamp_adam" in r[2]]
end_i = adam[-2] if len(adam) >= 2 else adam[-1]          # first of the last step's two optimizer launches
# walk back to the first bn_bwd_reduce after the previous optimizer launch
prev = [i for i in adam if i < end_i - 50]
lo = prev[-1] if prev else 0
first = next(i for i in range(lo, end_i) if "bn_bwd_reduce" in rows[i][2])
win = rows[first:end_i]
t0, t1 = win[0][0], max(r[1] for r in win)
print("encoder backward window: %.3f ms, %d kernels, queue column %s" % ((t1 - t0) / 1e6, len(win), qcol))
perq = defaultdict(list)
for s, e, n, q in win:
    perq[q].append((s, e, n))
for q, ks in sorted(perq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    print("queue %-6s kernels %5d  busy %7.3f ms  first +%.3f ms  last end +%.3f ms   last kernel %s" %
          (q, len(ks), busy / 1e6, (ks[0][0] - t0) / 1e6, (max(e for _, e, _ in ks) - t0) / 1e6, ks[-1][2][:60]))
# time with both / one / no queue busy (two busiest queues)
qs = sorted(perq, key=lambda q: -len(perq[q]))[:2]
ev = []
for qi, q in enumerate(qs):
    for s, e, _ in perq[q]:
        ev.append((s, 1, qi)); ev.append((e, -1, qi))
ev.sort()
cnt = [0, 0]; last = t0; acc = defaultdict(int)
for t, d, qi in ev:
    key = (cnt[0] > 0, cnt[1] > 0)
    acc[key] += t - last
    last = t
    cnt[qi] += d
for k, v in sorted(acc.items()):
    print("  main busy %-5s side busy %-5s : %7.3f ms" % (k[0], k[1], v / 1e6))
