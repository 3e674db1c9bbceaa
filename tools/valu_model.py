#!/usr/bin/env python3
"""The VALU issue floor of each profiled kernel: its dynamic instruction mix (rocprofv3 SQ counters, tools/pmc_sq_json.py)
priced with the issue cost of each instruction class measured on this chip (tools/valu_microbench).

    python tools/valu_model.py <pmc_sq.json> <valu_microbench.json | arch> <frame pixels of the profiled run> <out.json>

`arch` (round 3, what bench.py's valu_frac uses): the ARCHITECTURAL issue rate of a SIMD-32 instead of the microbenchmark's
wall cycles -- 2 cycles per wave64 instruction for binary32 / integer / everything unclassified, 4 for binary64 and 64-bit
integer, 8 for the quarter-rate transcendental unit, 16 for binary64 transcendentals (MI355X_MICROARCH.md, "Per-instruction
cycle constants") at the 2.4 GHz peak clock: a true lower bound (floor / measured <= 1 whatever the sustained clock),
where the wall-cycle prices of round 2 folded the clock deficit into the "floor" and overshot (1.01 on rgb_chain).

For every kernel: valu_per_wave (SQ_INSTS_VALU / waves), the share of each counted class, and
issue_cycles_per_wave = sum over classes of count x cycles, where `cycles` is the wall-clock cost of one wave64
instruction per SIMD at 8 waves per SIMD (the throughput figure of the microbenchmark, 2.4 GHz-equivalent):
  ADD/MUL/FMA_F32 -> v_add/v_mul/v_fma_f32, ADD/MUL/FMA_F64 -> v_add/v_mul/v_fma_f64, TRANS_F32 -> v_rcp_f32,
  TRANS_F64 -> v_rcp_f64, CVT -> v_cvt_f64_f32, INT32 -> min(v_add_u32, v_and_b32),
  INT64 -> 2 x INT32, everything the counters do not classify (compares, selects, moves, min/max, division fix-ups,
  ldexp/frexp, bit operations) -> the CHEAPEST full-rate instruction measured (v_and_b32): the floor is a lower bound --
  priced at the mean of the measured compare / max / move / division fix-up costs (3.97 cycles) it came out ABOVE the
  measured time of rgb_chain (4.75 against 3.93 ms), i.e. those instructions are cheaper in a real mix than in a loop of
  their own.
issue_floor_ms_per_mpix = issue_cycles_per_wave x waves / (1024 SIMDs x 2.4 GHz) / (frame megapixels): what bench.py
multiplies by its frame to state `valu_issue_floor_ms`; the measured time over that floor is the kernel's distance
from being purely issue-bound (LDS, memory latency, barriers, dependency stalls)."""
import json
import sys


def main():
    sq = json.load(open(sys.argv[1]))
    mpix = float(sys.argv[3]) / 1e6
    if sys.argv[2] == "arch":
        cyc = {"v_add_f32": 2.0, "v_mul_f32": 2.0, "v_fma_f32": 2.0, "v_add_f64": 4.0, "v_mul_f64": 4.0, "v_fma_f64": 4.0,
               "v_rcp_f32": 8.0, "v_rcp_f64": 16.0, "v_cvt_f64_f32": 2.0, "v_add_u32": 2.0, "v_and_b32": 2.0}
    else:
        mb = json.load(open(sys.argv[2]))["classes"]
        cyc = {k: v["W8"]["wall"] for k, v in mb.items()}
    f32 = {"ADD_F32": cyc["v_add_f32"], "MUL_F32": cyc["v_mul_f32"], "FMA_F32": cyc["v_fma_f32"]}
    f64 = {"ADD_F64": cyc["v_add_f64"], "MUL_F64": cyc["v_mul_f64"], "FMA_F64": cyc["v_fma_f64"]}
    int32 = min(cyc["v_add_u32"], cyc["v_and_b32"])  # lower bound: most INT32 work is address adds
    other = min(cyc["v_and_b32"], cyc["v_add_u32"], cyc["v_add_f32"])
    price = dict(f32)
    price.update(f64)
    price.update({"TRANS_F32": cyc["v_rcp_f32"], "TRANS_F64": cyc["v_rcp_f64"], "CVT": cyc["v_cvt_f64_f32"], "INT32": int32,
                  "INT64": 2 * int32})
    out = {"note": __doc__.split("\n\n")[0], "cycles_per_class": {k: round(v, 3) for k, v in price.items()},
           "cycles_unclassified": round(other, 3), "profiled_frame_mpix": mpix, "kernels": {}}
    for name, k in sq["kernels"].items():
        d = k.get("derived", {})
        share = d.get("valu_class_share")
        if not share or "valu_per_wave" not in d:
            continue
        n = d["valu_per_wave"]
        classified = sum(share.get(c, 0.0) for c in price)
        cycles = sum(share.get(c, 0.0) * n * price[c] for c in price) + max(1.0 - classified, 0.0) * n * other
        floor_ms = cycles * d["waves"] / (1024 * 2.4e9) * 1e3
        tag = name.split("<")[0]
        e = {"kernel": name, "valu_per_wave": n, "waves": d["waves"], "class_share": share,
             "unclassified_share": round(1.0 - classified, 4), "issue_cycles_per_wave": round(cycles, 1),
             "issue_floor_ms": round(floor_ms, 4), "measured_ms_profiled": k["ms"],
             "floor_over_measured": round(floor_ms / k["ms"], 4), "issue_floor_ms_per_mpix": round(floor_ms / mpix, 6)}
        out["kernels"][name] = e
        out["kernels"].setdefault(tag, e)  # first (= longest-running) instantiation under the bare name
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for name, e in out["kernels"].items():
        if "<" in name or name == e["kernel"]:
            print("%-36s valu/wave %8.1f  floor %7.3f ms  measured %7.3f ms  floor/measured %.2f"
                  % (name, e["valu_per_wave"], e["issue_floor_ms"], e["measured_ms_profiled"], e["floor_over_measured"]))


if __name__ == "__main__":
    main()
