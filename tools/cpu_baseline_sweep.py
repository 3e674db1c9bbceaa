#!/usr/bin/env python3
"""Which OpenMP binding serves the reference's CPU path best on this host?  Runs bench.py's cpu_baseline() in a fresh
process per setting (the OpenMP runtime reads its environment once) and prints one line each.

    python tools/cpu_baseline_sweep.py [sample size]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [("unbound", {}), ("spread/cores", {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}),
            ("close/cores", {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}),
            ("spread/threads", {"OMP_PROC_BIND": "spread", "OMP_PLACES": "threads"}),
            ("true", {"OMP_PROC_BIND": "true"})]
code = ("import sys, json; sys.path.insert(0, %r); import bench; "
        "print(json.dumps(bench.cpu_baseline(%r, True, 'light')))" % (ROOT, sys.argv[1] if len(sys.argv) > 1 else "24MP"))
for name, env in SETTINGS:
    e = dict(os.environ)
    for k in ("OMP_PROC_BIND", "OMP_PLACES"):
        e.pop(k, None)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("%-16s best %8.2f MPix/s at %3d threads; sweep %s" % (name, d["value"], d["cores"], d["thread_sweep_mpix_s"]), flush=True)
    except Exception:
        print(name, "failed:", out.stderr[-300:], flush=True)
