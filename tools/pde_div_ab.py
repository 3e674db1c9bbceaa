#!/usr/bin/env python3
"""What would the north star's 1 ULP buy the diffusion PDE?  (round 4's review, item 1 b)

    ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so python tools/pde_div_ab.py --size 100MP [--preset lens_deblur_soft]

Runs diffuse-or-sharpen on the measuring build twice on the same synthetic frame resident in HBM:
  exact    every division / square root correctly rounded (the product's arithmetic; the in-range forms of round 5)
  approx   ANSEL_HIP_PDE_APPROX_DIV=1: every division as v_rcp + one product + ONE residual correction (4 instructions
           instead of 8 - 11), the square root as the bare v_sqrt_f32 -- each <= 1 ulp from the correctly rounded result
and prints, as one JSON object: per-kernel milliseconds of both arms and the histogram of the ULP distance between the two
outputs, per colour channel (what the approximate forms do to the MODULE's output: errors of one ulp in a ratio are
squared, summed over nine samples, pass a division and four convolutions, and are iterated).  Nothing of the approximate
arm ships: PDE_APPROX() is the constant false in the product build.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="100MP")
    ap.add_argument("--preset", default="lens_deblur_soft")
    ap.add_argument("--iterations", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    if "measuring" not in os.environ.get("ANSEL_HIP_LIB", ""):
        raise SystemExit("set ANSEL_HIP_LIB to the measuring build (python -m ansel_amd.build --measuring)")
    import numpy as np
    import torch
    from ansel_amd import abi, lib, params, synth
    l = lib.init()
    dev = torch.device("cuda", 0)
    lib.check(l.dt_hip_set_stream(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "set_stream")
    w, h = synth.SIZES[args.size] if args.size in synth.SIZES else map(int, args.size.split("x"))
    tile = synth.rgba_image(1024, 1024, seed=3, lo=0.0, hi=1.3)
    img = torch.from_numpy(tile).to(dev).repeat(-(-h // 1024), -(-w // 1024), 1)[:h, :w].contiguous()
    piece = abi.Piece.make(w, h)
    d = params.diffuse(args.preset, iterations=args.iterations)

    def arm(approx):
        os.environ["ANSEL_HIP_PDE_APPROX_DIV"] = "1" if approx else "0"
        out = torch.empty_like(img)

        def run():
            lib.check(l.dt_hip_iop_diffuse_process(0, C.byref(piece), C.byref(d), img.data_ptr(), out.data_ptr()), "diffuse")
        run()
        torch.cuda.synchronize()
        l.dt_hip_events_reset(0)
        l.dt_hip_events_enable(0, 1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        l.dt_hip_events_enable(0, 0)
        tags, tms, cnt = (C.c_char_p * 64)(), (C.c_float * 64)(), (C.c_int * 64)()
        nk = l.dt_hip_events_profiling(0, tags, tms, cnt, 64)
        ker = {tags[i].decode(): round(tms[i] / args.steps, 4) for i in range(min(nk, 64))}
        return out, {"ms_per_call": round(ms, 3), "kernel_ms_per_call": ker}

    out0, t0 = arm(False)
    out1, t1 = arm(True)
    out0b, _ = arm(False)
    assert torch.equal(out0.view(torch.int32), out0b.view(torch.int32)), "the exact arm is not reproducible"

    def ordered(x):  # binary32 bit patterns as integers that sort like the floats
        i = x.view(torch.int32).to(torch.int64)
        return torch.where(i < 0, -(i & 0x7fffffff), i)

    hist = {}
    for c, name in enumerate("RGB"):
        dist = (ordered(out0[..., c].contiguous()) - ordered(out1[..., c].contiguous())).abs()
        bins = torch.bincount(dist.clamp(max=17).flatten(), minlength=18).cpu().tolist()
        hist[name] = {"0": bins[0], "1": bins[1], "2": bins[2], "3-4": bins[3] + bins[4], "5-8": sum(bins[5:9]), "9-16": sum(bins[9:17]),
                      ">16": bins[17], "max_ulp": int(dist.max().item()), "values": int(dist.numel())}
        hist[name]["share_within_1_ulp"] = round((bins[0] + bins[1]) / dist.numel(), 6)
    res = {"what": "diffuse or sharpen, preset %s, %d iterations, %d x %d: exact divisions / square roots against v_rcp + one correction "
                   "(<= 1 ulp each); measuring build, same process, same frame" % (args.preset, args.iterations, w, h),
           "exact": t0, "approx_1ulp_per_operation": t1, "ulp_distance_of_the_module_output": hist}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
