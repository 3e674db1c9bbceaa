#!/usr/bin/env python3
"""Merge the two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) of one bench.py command
into per-kernel HBM bytes per launch.

    python tools/pmc_hbm_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [note]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for gfx950: both
counters are in KB (x 1024); FETCH_SIZE tallies 64 B per 128-B request on wide coalesced reads, so
read bytes = 2 x FETCH_SIZE; WRITE_SIZE is taken as reported."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("ansel::", "")
    m = re.match(r"(?:void\s+)?([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name[:40]


def per_kernel(path, counter):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


def main():
    fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": (sys.argv[4] if len(sys.argv) > 4 else "") + " rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate "
           "passes; per launch; read bytes = 2 x FETCH_SIZE x 1024 (gfx950: 64 B tallied per 128-B request), write "
           "bytes = WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM)", "kernels": {}}
    for k in sorted(fetch):
        rd = 2.0 * fetch[k] * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"launches_sampled": n[k], "FETCH_SIZE_KB": round(fetch[k], 1),
                             "WRITE_SIZE_KB": round(write.get(k, 0.0), 1), "read_bytes": int(rd), "write_bytes": int(wr),
                             "hbm_bytes": int(rd + wr)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-28s read %8.1f MB  write %8.1f MB" % (k, v["read_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
