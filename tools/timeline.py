"""Kernel timeline of a rocprofv3 --kernel-trace run: for a window in the middle of the trace, every
kernel with its duration, the gap since the end of the previous kernel on the device (negative =
overlap with a kernel of another stream) and its stream / queue.  usage: timeline.py trace.csv [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"))
             for r in rows), key=lambda t: t[0])
mid = len(ks) * 3 // 5
prev_end = ks[mid - 1][1]
t0 = ks[mid][0]
for s, e, name, q in ks[mid:mid + n]:
    short = name.split("(")[0].split("::")[-1][:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{q}  {short}")
    prev_end = max(prev_end, e)
