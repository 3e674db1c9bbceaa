#!/usr/bin/env python3
"""A stream of frames through dt_hip_batch_* (pinned host -> device -> pipe -> pinned host, three slots): wall time per frame, for
running under `rocprofv3 --kernel-trace --memory-copy-trace` to see where the transfers sit against the kernels.

    python tools/batch_timeline.py [--size 100MP] [--pipe full] [--frames 8]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="100MP")
    ap.add_argument("--pipe", default="full")
    ap.add_argument("--frames", type=int, default=8)
    args = ap.parse_args()
    import numpy as np
    import bench
    from ansel_amd import abi, lib, params, pipe, synth
    l = lib.init()
    w, h = bench.frame_size(args.size)
    lut = params.srgb_encode_lut()
    d_lut = lib.DeviceBuffer.from_numpy(0, lut)
    nodes = bench.build_pipe(w, h, d_lut.ptr, lut, bench.have_filmic(), args.pipe)
    ex = pipe.DevicePipe(0, nodes, fusion=True)
    raw = synth.bayer_mosaic_tiled(w, h, seed=1)
    nb_in, nb_out = raw.nbytes, w * h * 8
    pin_in = l.dt_hip_alloc_host_pinned(nb_in)
    pin_outs = [l.dt_hip_alloc_host_pinned(nb_out) for _ in range(3)]
    C.memmove(pin_in, raw.ctypes.data, nb_in)
    b = l.dt_hip_batch_new(ex.handle, 3, nb_in, nb_out)
    stamps = []
    for k in range(args.frames + 3):
        if k == 3:
            l.dt_hip_batch_drain(b)
            t0 = time.perf_counter()
        rc = l.dt_hip_batch_submit(b, pin_in, pin_outs[k % 3])
        assert rc >= 0, l.dt_hip_last_error()
        if k >= 3:
            stamps.append(time.perf_counter() - t0)
    l.dt_hip_batch_drain(b)
    total = time.perf_counter() - t0
    print("frames %d: %.2f ms per frame; submit() returned at (ms): %s" % (args.frames, total / args.frames * 1e3,
                                                                          " ".join("%.1f" % (s * 1e3) for s in stamps)))
    l.dt_hip_batch_free(b)
    ex.close()


if __name__ == "__main__":
    main()
