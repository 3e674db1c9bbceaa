#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of tools/batch_timeline.py: copies by direction, the copy kernels
the runtime launched on its own (__amd_rocclr_*), gaps in the kernel stream.   python tools/batch_timeline_report.py <kernel_trace.csv> <memory_copy_trace.csv>"""
import csv
import sys
from collections import defaultdict

k = list(csv.DictReader(open(sys.argv[1])))
m = list(csv.DictReader(open(sys.argv[2])))
t0 = min(int(r["Start_Timestamp"]) for r in k + m)
by = defaultdict(lambda: [0, 0.0, 1e18, 0.0])
for r in m:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    e = by[r["Direction"]]
    e[0] += 1
    e[1] += d
    e[2] = min(e[2], (int(r["Start_Timestamp"]) - t0) / 1e6)
    e[3] = max(e[3], d)
for d, e in by.items():
    print("copies %-32s n %4d total %8.2f ms longest %7.2f ms first at %.1f ms" % (d, e[0], e[1], e[3], e[2]))
rk = [r for r in k if "rocclr" in r["Kernel_Name"]]
tot = defaultdict(lambda: [0, 0.0, 0.0])
for r in rk:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    e = tot[r["Kernel_Name"][:40]]
    e[0] += 1
    e[1] += d
    e[2] = max(e[2], d)
for n, e in tot.items():
    print("runtime kernel %-40s n %4d total %8.2f ms longest %7.2f ms" % (n, e[0], e[1], e[2]))
# the long copies and what the compute queue did meanwhile
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r["Queue_Id"]) for r in k)
for r in m:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < 2e6:
        continue
    busy = sum(min(e, ke) - max(s, ks_) for ks_, ke, _, _ in ks if ke > s and ks_ < e)
    print("%-28s %9.1f - %9.1f ms (%6.2f ms): kernels ran %.2f ms of it" % (r["Direction"], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, busy / 1e6))
long_k = [x for x in ks if x[1] - x[0] > 5e6 and "rocclr" in x[2]]
for s, e, n, q in long_k:
    others = [(max(s, a), min(e, b), nm) for a, b, nm, qq in ks if qq != q and b > s and a < e]
    busy = sum(b - a for a, b, _ in others)
    print("long runtime kernel %s on queue %s: %.1f - %.1f ms; kernels of other queues ran %.2f ms of it: %s" % (
        n, q, (s - t0) / 1e6, (e - t0) / 1e6, busy / 1e6, ", ".join(sorted({nm[:24] for _, _, nm in others}))[:160]))
qs = defaultdict(int)
for _, _, _, q in ks:
    qs[q] += 1
print("kernels per queue:", dict(qs))
