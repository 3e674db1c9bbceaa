#!/usr/bin/env python3
"""Effective shader clock per kernel of a bench.py run: GRBM_GUI_ACTIVE (busy cycles of the graphics engine during the
dispatch) over the dispatch's duration -- MI355X_MICROARCH.md, "DVFS give-back": the chip clocks to its power budget,
effective clock ~ GRBM_GUI_ACTIVE / kernel wall time.

    python tools/pmc_sclk_json.py <counter_collection.csv> <out.json> <note>

rocprofv3 reports the counter summed over the device's XCCs (eight on an MI355X): a value per nanosecond above 5 is
divided by eight (stated in the output).  Round 4's review, item 4: separate "cycles per instruction" from "clock"."""
import collections
import csv
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_sq_json import short  # noqa: E402


def main():
    path, out_path, note = sys.argv[1], sys.argv[2], sys.argv[3]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if ns > 0:
            per[short(r["Kernel_Name"])].append((float(r["Counter_Value"]), ns))
    out = {"note": note + " rocprofv3 --pmc GRBM_GUI_ACTIVE; sclk_mhz = cycles / duration, the counter divided by the number of XCCs it is "
           "summed over where the raw ratio exceeds 5 GHz; under the profiler (MI355X_MICROARCH.md: profiled passes clock 3 - 5 % lower)",
           "kernels": {}}
    for k, v in sorted(per.items(), key=lambda kv: -sum(ns for _, ns in kv[1])):
        cyc = sum(c for c, _ in v)
        ns = sum(n for _, n in v)
        raw = cyc / ns  # GHz
        xcc = 8 if raw > 5.0 else 1
        out["kernels"][k] = {"launches": len(v), "ms_avg": round(ns / len(v) / 1e6, 4), "gui_active_cycles_avg": round(cyc / len(v), 1),
                             "xcc_divisor": xcc, "sclk_mhz": round(raw / xcc * 1e3, 1)}
    json.dump(out, open(out_path, "w"), indent=1)
    for k, e in out["kernels"].items():
        print("%-40s %9.4f ms  %7.1f MHz" % (k, e["ms_avg"], e["sclk_mhz"]))


if __name__ == "__main__":
    main()
