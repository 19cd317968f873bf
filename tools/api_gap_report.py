"""Reads rocprofv3's kernel trace of dropin_bench: the timed CG solve's last 100 iterations - kernels per
iteration, busy time, idle time between consecutive kernels (largest gaps first), and the HIP API stats."""
import csv
import glob
import os
import sys

d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not kt:
    print("no kernel trace"); sys.exit(0)
rows = list(csv.DictReader(open(kt[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "").replace("gkoc::", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0]
# the timed solve = the last 200 occurrences of the SpMV kernel family after the apply loop
spmv = [i for i, r in enumerate(rows) if "csr_spmv" in r["Kernel_Name"]]
print(f"{len(rows)} kernels, {len(spmv)} csr_spmv launches")
first = spmv[-100]
seg = rows[first:]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"last 100 CG iterations: {len(seg)} kernels = {len(seg) / 100:.1f} per iteration, wall {(t1 - t0) / 1e5:.2f} us/iter, "
      f"busy {busy / 1e5:.2f} us/iter, idle {(t1 - t0 - busy) / 1e5:.2f} us/iter")
per = {}
for r in seg:
    k = name(r)
    a = per.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("\nkernel                                      launches/iter   us/launch   us/iter")
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:44]:44s} {c / 100:10.2f} {t / c / 1e3:12.2f} {t / 1e5:10.2f}")
gaps = {}
for a, b in zip(seg, seg[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    key = f"{name(a)[:30]} -> {name(b)[:30]}"
    e = gaps.setdefault(key, [0, 0])
    e[0] += 1
    e[1] += g
print("\nidle between consecutive kernels                                    count   us each   us/iter")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k:66s} {c:6d} {t / c / 1e3:9.2f} {t / 1e5:9.2f}")
# the Csr::apply loop: 50 + 2 launches in a row before the solver is generated
ap = rows[spmv[2]:spmv[51] + 1]
if len(ap) == 50:
    w = (int(ap[-1]["End_Timestamp"]) - int(ap[0]["Start_Timestamp"])) / 50e3
    b_ = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ap) / 50e3
    print(f"\nCsr::apply loop: {w:.2f} us per apply on the device's clock, kernel {b_:.2f} us")
else:
    print(f"\nCsr::apply loop: {len(ap)} kernels between the 3rd and the 52nd SpMV (other kernels in between):")
    seen = {}
    for r in ap:
        seen[name(r)] = seen.get(name(r), 0) + 1
    print(seen)
for f in glob.glob(os.path.join(d, "**", "*hip_api_stats.csv"), recursive=True):
    print("\n== HIP API stats")
    print("".join(open(f).readlines()[:16]))
# round 6: one iteration of the timed solve seen from BOTH sides - the HIP calls of the host thread (start,
# duration) between the kernels' start / end on the device, on one time axis (rocprofv3 reports both in ns of
# the same clock)
ha = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
if ha:
    api = list(csv.DictReader(open(ha[0])))
    k0 = spmv[-50]                         # an SpMV in the middle of the timed solve
    k1 = spmv[-48]
    w0, w1 = int(rows[k0]["Start_Timestamp"]) - 5000, int(rows[k1]["Start_Timestamp"]) + 5000
    ev = []
    for r in rows[k0:k1 + 1]:
        ev.append((int(r["Start_Timestamp"]), f"    device  START {name(r)}"))
        ev.append((int(r["End_Timestamp"]), f"    device  END   {name(r)}  ({(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f} us)"))
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if w0 <= s <= w1 and not r["Function"].startswith("__hip"):
            ev.append((s, f"host  {r['Function']}  ({(e - s) / 1e3:.1f} us)"))
    ev.sort()
    print("\n== two iterations, host and device on one axis (us from the first SpMV's start)")
    base = int(rows[k0]["Start_Timestamp"])
    for t, what in ev:
        print(f"{(t - base) / 1e3:10.1f}  {what}")
