#!/usr/bin/env python3
"""Per-launch durations out of a rocprofv3 --kernel-trace directory, in dispatch order.

    python tools/kernel_trace_list.py <dir> [name substring]"""
import csv
import os
import sys

root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
rows = []
for d, _, files in os.walk(root):
    for f in files:
        if f.endswith("kernel_trace.csv"):
            with open(os.path.join(d, f)) as fh:
                for r in csv.DictReader(fh):
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for s, e, n in rows:
    if pat in n:
        short = n.split("(")[0].split("::")[-1]
        print("%-40s %9.1f us" % (short[:40], (e - s) / 1000.0))
