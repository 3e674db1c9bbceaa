#!/usr/bin/env python3
"""Time the blend stage with mask feathering (the guided filter), with a mask blur and with a details threshold on a 24 MP
frame (run on the GPU box).

    python tools/bench_feather.py [radius]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ansel_amd import abi, lib, params, synth  # noqa: E402


def main():
    radius = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    w, h = 6000, 4000
    l = lib.init()
    img = synth.rgba_image(1024, 1024, seed=3, lo=0.0, hi=1.0)
    a = np.tile(img, (4, 6, 1))[:h, :w].copy()
    b = np.ascontiguousarray(a[::-1])
    da, db = lib.DeviceBuffer.from_numpy(0, a), lib.DeviceBuffer.from_numpy(0, b)
    piece = abi.Piece.make(w, h)
    d = abi.BlendData.uniform(params.WORK_IN, 80.0).channel(abi.BLENDIF_GRAY_in, 0.05, 0.2, 0.6, 0.9, boost=1.0)
    rm = lib.DeviceBuffer(0, w * h * 4)
    scratch = lib.DeviceBuffer(0, w * h * 16)
    lib.check(l.dt_hip_iop_detailmask_process(0, C.byref(piece), C.byref(abi.DetailmaskData.make((2.0, 1.0, 1.5), rm.ptr)),
                                              da.ptr, scratch.ptr), "detailmask")
    for label, r, blur, details in (("parametric mask only", 0.0, 0.0, 0.0), ("+ feathering radius %g" % radius, radius, 0.0, 0.0),
                                    ("+ mask blur radius %g" % radius, 0.0, radius, 0.0), ("+ details threshold 0.3", 0.0, 0.0, 0.3)):
        d.feathering_radius, d.feathering_guide, d.blur_radius = r, abi.MASK_GUIDE_OUT_AFTER_BLUR, blur
        d.details, d.detail_mask = details, (rm.ptr if details else None)
        l.dt_hip_events_reset(0)
        l.dt_hip_events_enable(0, 1)
        ts = []
        for _ in range(3):
            l.dt_hip_finish(0)
            t0 = time.perf_counter()
            lib.check(l.dt_hip_develop_blend_process(0, C.byref(piece), C.byref(d), da.ptr, db.ptr), "blend")
            l.dt_hip_finish(0)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%-32s %8.2f ms" % (label, min(ts)), flush=True)
    import bench
    for k, v in sorted(bench.read_kernel_events(l, 0).items()):
        print("   %-24s %8.3f ms avg x %d" % (k, v["ms_avg"], v["launches"]))


if __name__ == "__main__":
    main()
