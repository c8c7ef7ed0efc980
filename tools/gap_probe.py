"""Developer probe (GPU box): kernel durations and the gaps between consecutive kernels of back-to-back sorts, from a
rocprofv3 --kernel-trace CSV. usage: python tools/gap_probe.py <dir with *_kernel_trace.csv> [kernels per sort]"""
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)   # the newest run
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = [r for r in rows if "clo_" in r[2]]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = rows[len(rows) // 2 // per * per:]            # the second half: warm
dur, gap, names = {}, {}, []
for i, (s, e, n) in enumerate(rows):
    k = n.split("<")[0].replace("void (anonymous namespace)::", "")
    pos = i % per
    names.append(k) if len(names) < per else None
    dur.setdefault(pos, []).append(e - s)
    if i + 1 < len(rows):
        gap.setdefault(pos, []).append(rows[i + 1][0] - e)
tot_d = tot_g = 0.0
for pos in range(per):
    d = sum(dur[pos]) / len(dur[pos]) / 1e3
    g = sum(gap.get(pos, [0])) / max(1, len(gap.get(pos, [0]))) / 1e3
    tot_d += d
    tot_g += g
    print("%2d %-36s %8.2f us   then idle %6.2f us" % (pos, names[pos], d, g))
print("sum of kernels %.1f us, sum of gaps %.1f us (one sort)" % (tot_d, tot_g))
