"""Decode-step anatomy from a rocprofv3 --kernel-trace CSV: for every kernel of the recurrence (forward and backward
loops) the mean duration and the mean gap between the end of its predecessor in the same loop and its own start,
over the last <n_steps> decoder calls.  usage: trace_decode_step.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()


def short(n):
    n = n.replace("scn::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


FWD = ("skinny_kernel", "attn_scores", "attn_context", "scn_mix_fwd", "lstm_fwd")
BWD = ("lstm_bwd", "skinny_kernel", "scn_mix_bwd", "gate_bwd", "attn_dalpha", "attn_softmax_bwd")
# a loop = a maximal run of kernels whose names are all in the set and that contains the loop's marker kernel
for label, names, marker in (("forward", FWD, "lstm_fwd"), ("backward", BWD, "lstm_bwd")):
    runs, cur = [], []
    for r in rows:
        if any(k in r[2] for k in names):
            cur.append(r)
        else:
            if sum(marker in c[2] for c in cur) >= 10:
                runs.append(cur)
            cur = []
    if sum(marker in c[2] for c in cur) >= 10:
        runs.append(cur)
    runs = runs[-3:]
    if not runs:
        print(label, ": no loop found")
        continue
    dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
    steps = 0
    span = 0
    for run in runs:
        steps += sum(marker in c[2] for c in run)
        span += run[-1][1] - run[0][0]
        for i, (s, e, n, q) in enumerate(run):
            key = short(n)
            if "skinny" in key:      # tell the step's products apart by their position after the previous marker-type kernel
                j, pos = i, 0
                while j > 0 and marker not in run[j - 1][2]:
                    j -= 1
                    pos += "skinny" in run[j][2]
                key = "skinny #%d %s" % (pos, key[14:])
            dur[key] += e - s
            cnt[key] += 1
            if i:
                gap[key] += max(0, s - run[i - 1][1])
    print("%s loop: %d steps in %d runs, %.2f us per step (first start -> last end)" % (label, steps, len(runs), span / 1e3 / steps))
    tot_d = tot_g = 0.0
    for k in sorted(cnt, key=lambda k: -dur[k]):
        per = cnt[k] / steps
        d, g = dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3
        tot_d += d * per
        tot_g += g * per
        print("  %-52s x%.2f/step  duration %6.2f us  gap before %5.2f us" % (k, per, d, g))
    print("  sum per step: kernels %.2f us + gaps %.2f us" % (tot_d, tot_g))
