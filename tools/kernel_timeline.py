"""Prints the kernels of the LAST call found in a rocprofv3 kernel trace (…_kernel_trace.csv), in start order, with the
idle time before each: where a step's time goes beyond the sum of its kernels.
usage: python tools/kernel_timeline.py <dir or csv> <kernels per call>"""
import csv
import glob
import os
import sys

path = sys.argv[1]
per_call = int(sys.argv[2])
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
rows = [r for r in rows if r[2].startswith("clo_") or "clo_" in r[2]][-per_call:]
t0 = rows[0][0]
prev_end = t0
busy = 0
print("#   start   duration   idle before   kernel   (microseconds)")
for s, e, name, qid in rows:
    print("%9.1f %9.1f %9.1f   %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name.replace("(anonymous namespace)::", "")))
    busy += e - s
    prev_end = max(prev_end, e)
print("# call: %.1f us from the first kernel's start to the last one's end; kernels %.1f us, idle %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))
