#!/usr/bin/env python3
"""print the per-kernel table of one or more bench.py JSON lines (gpurun_out/*.log)"""
import json
import sys

for path in sys.argv[1:]:
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    c = d["config"]
    print("%s: %.1f MPix/s, %.3f ms/step, pipe %.1f%% of HBM roofline, %s" % (
        path, d["value"], d["ms_per_step"], 100 * c["pipe_hbm_frac"], c.get("executor", "")))
    for k, v in sorted(c["kernels_ms"].items(), key=lambda kv: -kv[1]):
        print("    %-22s %8.4f ms" % (k, v))
