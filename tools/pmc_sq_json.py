#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (one counter_collection.csv per pass, SQ counters) of one bench.py command into a
per-kernel table: mean counter value per launch, mean duration, and the derived figures DESIGN.md quotes.

    python tools/pmc_sq_json.py <out.json> <note> <pass1.csv> [<pass2.csv> ...]

Kernel keys keep their template arguments (rgb_chain<1, 0> is the light pipe's fused RGBA group, the full pipe's
two groups are other instantiations).  Derived, per launch (MI355X_MICROARCH.md: SQ_WAVE_CYCLES, SQ_WAIT_* and
SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES counts cycles per SE-level SQ):
  valu_per_wave        SQ_INSTS_VALU / waves                       (dynamic VALU instructions per wave)
  valu_issue_cycles    SQ_INST_CYCLES_VALU / SQ_INSTS_VALU         (average issue cycles per VALU instruction, as counted)
  valu_active_frac     SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / mean resident waves) -- reported raw, see DESIGN.md
  lds_conflict_frac    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("ansel::", "")
    m = re.match(r"(?:void\s+)?([A-Za-z_0-9]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else name[:48]


def main():
    out_path, note, passes = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    meta = {}
    for path in passes:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][(path, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            if k not in meta:
                meta[k] = {"grid": int(r.get("Grid_Size", 0) or 0), "workgroup": int(r.get("Workgroup_Size", 0) or 0),
                           "vgpr": int(r.get("VGPR_Count", 0) or 0), "sgpr": int(r.get("SGPR_Count", 0) or 0),
                           "lds": int(r.get("LDS_Block_Size", 0) or 0)}
    out = {"note": note + " rocprofv3 --pmc, %d passes of <= 8 SQ counters; values are means per launch under the "
           "profiler (clock 3-5 %% lower than unprofiled, MI355X_MICROARCH.md DVFS)" % len(passes), "kernels": {}}
    for k in sorted(agg, key=lambda k: -sum(dur[k].values())):
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        e = dict(meta[k])
        e["launches_sampled"] = len(dur[k]) // max(len(passes), 1)
        e["ms"] = round(sum(dur[k].values()) / len(dur[k]), 4)
        e["counters"] = {n: round(v, 1) for n, v in sorted(c.items())}
        waves = e["grid"] / 64.0 if e["grid"] else 0.0
        d = {}
        if waves and "SQ_INSTS_VALU" in c:
            d["waves"] = waves
            d["valu_per_wave"] = round(c["SQ_INSTS_VALU"] / waves, 1)
            for n in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
                if n in c:
                    d[n[9:].lower() + "_per_wave"] = round(c[n] / waves, 1)
        if c.get("SQ_INSTS_VALU") and "SQ_INST_CYCLES_VALU" in c:
            d["valu_issue_cycles_per_instr"] = round(c["SQ_INST_CYCLES_VALU"] / c["SQ_INSTS_VALU"], 3)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_WAVE_CYCLES"):
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if n in c:
                    d[n[3:].lower() + "_over_wave_cycles"] = round(c[n] / c["SQ_WAVE_CYCLES"], 4)
        cls = {n[14:]: c[n] for n in c if n.startswith("SQ_INSTS_VALU_") and "MFMA" not in n and "FLOPS" not in n}
        if cls and c.get("SQ_INSTS_VALU"):
            d["valu_class_share"] = {n: round(v / c["SQ_INSTS_VALU"], 4) for n, v in sorted(cls.items())}
        e["derived"] = d
        out["kernels"][k] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for k, e in out["kernels"].items():
        print("%-40s %9.4f ms  %s" % (k, e["ms"], json.dumps(e["derived"])))


if __name__ == "__main__":
    main()
