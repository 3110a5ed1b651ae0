import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sgemm_kernel" in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]]
end = max(int(r["End_Timestamp"]) for r in rows)
agg = defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) < end - int(float(sys.argv[2]) * 1e6):
        continue
    name = r["Kernel_Name"].split("sgemm_kernel")[-1][:22] if "sgemm" in r["Kernel_Name"] else "splitk_reduce"
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1; agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-24s grid %6s %5s %4s  calls %3d  avg %8.1f us  total %8.1f us" % (k[0], k[1], k[2], k[3], c, t / c / 1e3, t / 1e3))
