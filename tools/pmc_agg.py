"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: mean counter value per dispatch.
usage: pmc_agg.py <dir> <out.csv>   (walks <dir> for *counter_collection*.csv)"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True) if "counter" in os.path.basename(f)]
agg = defaultdict(lambda: [0, 0.0])
cols = None
for f in files:
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        for r in rd:
            name = r.get("Kernel_Name") or r.get("Name") or "?"
            cname = r.get("Counter_Name") or "?"
            val = float(r.get("Counter_Value") or 0.0)
            a = agg[(name, cname)]
            a[0] += 1
            a[1] += val
with open(dst, "w") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "MeanValue", "columns=%s files=%d" % ("|".join(cols or []), len(files))])
    for (n, c), (k, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n[:200], c, k, "%.3f" % (v / k)])
