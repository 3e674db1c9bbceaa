#!/usr/bin/env python3
"""What does each part of diffuse_pde_strip cost?  (measuring build; round 5)

    ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so python tools/pde_off_ab.py --size 100MP

Times diffuse-or-sharpen (lens_deblur_soft, 2 iterations: 10 PDE launches) with parts of the strip kernel switched off through
ANSEL_HIP_PDE_OFF (diffuse.hip PDE_OFF: 1 barrier, 2 fetches, 4 the squared-ratio ring, 8 the store, 16 the update arithmetic).  The
results of the switched-off runs are wrong by construction; only `diffuse_pde` ms per call is read."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="100MP")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    if "measuring" not in os.environ.get("ANSEL_HIP_LIB", ""):
        raise SystemExit("set ANSEL_HIP_LIB to the measuring build (python -m ansel_amd.build --measuring)")
    import torch
    from ansel_amd import abi, lib, params, synth
    l = lib.init()
    dev = torch.device("cuda", 0)
    lib.check(l.dt_hip_set_stream(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "set_stream")
    w, h = synth.SIZES[args.size]
    tile = synth.rgba_image(1024, 1024, seed=3, lo=0.0, hi=1.3)
    img = torch.from_numpy(tile).to(dev).repeat(-(-h // 1024), -(-w // 1024), 1)[:h, :w].contiguous()
    out = torch.empty_like(img)
    piece = abi.Piece.make(w, h)
    d = params.diffuse("lens_deblur_soft", iterations=2)

    def timed(off):
        os.environ["ANSEL_HIP_PDE_OFF"] = str(off)
        lib.check(l.dt_hip_iop_diffuse_process(0, C.byref(piece), C.byref(d), img.data_ptr(), out.data_ptr()), "diffuse")
        torch.cuda.synchronize()
        l.dt_hip_events_reset(0)
        l.dt_hip_events_enable(0, 1)
        for _ in range(args.steps):
            lib.check(l.dt_hip_iop_diffuse_process(0, C.byref(piece), C.byref(d), img.data_ptr(), out.data_ptr()), "diffuse")
        torch.cuda.synchronize()
        l.dt_hip_events_enable(0, 0)
        tags, tms, cnt = (C.c_char_p * 64)(), (C.c_float * 64)(), (C.c_int * 64)()
        nk = l.dt_hip_events_profiling(0, tags, tms, cnt, 64)
        return {tags[i].decode(): round(tms[i] / args.steps, 3) for i in range(min(nk, 64))}.get("diffuse_pde")

    names = {0: "everything on", 1: "no barrier", 2: "no fetches behind the first rows", 4: "no squared-ratio ring (LDS)", 8: "no store",
             16: "no update arithmetic", 2 + 8: "no fetches, no store", 1 + 4: "no barrier, no ring", 1 + 2 + 4 + 8: "arithmetic only (update + ratios' divisions off with the ring)",
             16 + 4: "no arithmetic at all (fetch, barrier, store)", 31: "the loop skeleton"}
    res = {"what": "diffuse_pde_strip<true, 253>, %d x %d, 10 launches per call: ms per call with parts switched off (ANSEL_HIP_PDE_OFF)" % (w, h), "runs": {}}
    for off in (0, 1, 2, 4, 8, 16, 10, 5, 15, 20, 31, 0):
        res["runs"]["%d: %s%s" % (off, names[off], " (again)" if off == 0 and res["runs"] else "")] = timed(off)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
