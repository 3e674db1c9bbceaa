#!/usr/bin/env python3
"""Where a workgroup of nlm_chunks_v2 spends its cycles (run on the GPU box): the measuring build of the kernel
(ANSEL_NLM2_TIMED: a clock read around every step of the four-table schedule, nlm2_body.h) leaves, per wave, the cycles
of A1, C, A2, the first table row, B and the wait at the barrier, summed over the 225 offsets.

    python tools/nlm_phase_clocks.py [WxH] > gpurun_out/nlm_phase_clocks.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the A/B switches exist in the measuring build only (python -m ansel_amd.build --measuring, before gpurun)
os.environ.setdefault("ANSEL_HIP_LIB", os.path.join(ROOT, "ansel_amd", "libansel_hip_measuring.so"))
import numpy as np  # noqa: E402

from ansel_amd import abi, lib  # noqa: E402


def slice_height(height, base=60):
    """compute_slice_height(), src/pixel/nlmeans_core.c:264-295 (as in nlmeans.hip)"""
    if height % base == 0:
        return base
    best, best_incr = height % base, 0
    for incr in range(1, 10):
        plus = height % (base + incr)
        if plus == 0:
            return base + incr
        if plus > best:
            best_incr, best = incr, plus
        minus = height % (base - incr)
        if minus == 0:
            return base - incr
        if minus > best:
            best_incr, best = -incr, minus
    return base + best_incr


def main():
    w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "11648x2184").split("x"))
    l = lib.init()
    rng = np.random.default_rng(1)
    img = rng.random((h, w, 4), dtype=np.float32) * np.float32(100.0)
    din = lib.DeviceBuffer.from_numpy(0, img)
    dout = lib.DeviceBuffer(0, img.nbytes)
    piece = abi.Piece.make(w, h)
    d = abi.NlmeansData(2.0, 50.0, 0.5, 1.0)
    os.environ["ANSEL_NLM2_TIMED"] = "1"
    for _ in range(2):
        lib.check(l.dt_hip_iop_nlmeans_process(0, C.byref(piece), C.byref(d), din.ptr, dout.ptr), "nlmeans")
        l.dt_hip_finish(0)
    out = dout.to_numpy((h, w, 4), np.float32)
    chk_w = 72
    chk_h = slice_height(h)
    res = {"frame": [w, h], "chunk": [chk_w, chk_h], "offsets": 225, "chunks": {}}
    names = ["A1", "C", "A2", "first_row", "B", "barrier_wait"]
    for cy, cx in ((3, 5), (10, 40), (20, 100)):
        top, left = cy * chk_h, cx * chk_w
        if top + chk_h > h - 16 or left + chk_w > w - 16:
            continue
        waves = []
        for wv in range(16):
            a, b = out[top, left + 2 * wv], out[top, left + 2 * wv + 1]
            v = [float(a[0]), float(a[1]), float(a[2]), float(a[3]), float(b[0]), float(b[1])]
            waves.append({n: round(x / 228.0, 1) for n, x in zip(names, v)})  # per stage (225 offsets + 3)
        res["chunks"]["%d,%d" % (cy, cx)] = waves
        print("chunk (%d, %d): cycles per stage" % (cy, cx), file=sys.stderr)
        for wv, e in enumerate(waves):
            print("  wave %2d  " % wv + "  ".join("%s %7.1f" % (n, e[n]) for n in names) + "   sum %7.1f" % sum(e.values()),
                  file=sys.stderr)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
