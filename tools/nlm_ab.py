#!/usr/bin/env python3
"""A/B timing of the non-local-means chunk kernels on the GPU box (a MEASURING build of the library: the switches below are
read from the environment only there): the third version, its fused variant (three tables, the row recurrence inside
the weights' waves) and the second version, on the 100 MP frame's chunk grid (72 x 56) and on the 60 MP frame's (68 x 64),
each frame a slice of the real one's height.  Roles switched off compute garbage: that only shows where the time goes.

    python tools/nlm_ab.py > gpurun_out/nlm_ab.json
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the A/B switches exist in the measuring build only (python -m ansel_amd.build --measuring, before gpurun)
os.environ.setdefault("ANSEL_HIP_LIB", os.path.join(ROOT, "ansel_amd", "libansel_hip_measuring.so"))
import numpy as np  # noqa: E402

from ansel_amd import abi, lib  # noqa: E402

KEYS = ("ANSEL_HIP_NLM_V1", "ANSEL_HIP_NLM_V2", "ANSEL_HIP_NLM_FUSED", "ANSEL_NLM2_VARIANT", "ANSEL_NLM2_DEEP", "ANSEL_NLM2_LAYOUT")


def run(l, w, h, variants):
    rng = np.random.default_rng(1)
    img = rng.random((h, w, 4), dtype=np.float32) * np.float32(100.0)
    din = lib.DeviceBuffer.from_numpy(0, img)
    dout = lib.DeviceBuffer(0, img.nbytes)
    piece = abi.Piece.make(w, h)
    d = abi.NlmeansData(2.0, 50.0, 0.5, 1.0)
    out = {}
    for name, env in variants.items():
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        times = []
        for rep in range(4):
            l.dt_hip_finish(0)
            t0 = time.perf_counter()
            lib.check(l.dt_hip_iop_nlmeans_process(0, C.byref(piece), C.byref(d), din.ptr, dout.ptr), "nlmeans")
            l.dt_hip_finish(0)
            times.append((time.perf_counter() - t0) * 1e3)
        out[name] = {"env": env, "ms": round(min(times[1:]), 3), "ms_per_mpix": round(min(times[1:]) / (w * h / 1e6), 4)}
        print("%5d x %4d  %-34s %8.3f ms  %.4f ms/MPix" % (w, h, name, min(times[1:]), min(times[1:]) / (w * h / 1e6)), file=sys.stderr, flush=True)
    return out


def main():
    l = lib.init()
    fused = {"ANSEL_HIP_NLM_FUSED": "1"}
    res = {}
    res["11648x2184 (72 x 56 chunks)"] = run(l, 11648, 2184, {
        "v3": {}, "v3 roles dealt by VALU load (layout 1)": {"ANSEL_NLM2_VARIANT": "4096"}, "v3 again": {}, "v4 fused": fused,
        "v4 no A1": dict(fused, ANSEL_NLM2_VARIANT="16"), "v4 no A2": dict(fused, ANSEL_NLM2_VARIANT="32"),
        "v4 no row chain": dict(fused, ANSEL_NLM2_VARIANT="64"), "v4 no C": dict(fused, ANSEL_NLM2_VARIANT="128"),
        "v4 only barriers": dict(fused, ANSEL_NLM2_VARIANT=str(16 + 32 + 64 + 128)),
        "v2": {"ANSEL_HIP_NLM_V2": "1"}})
    res["9504x2112 (68 x 64 chunks)"] = run(l, 9504, 2112, {
        "v4 fused (default here)": {}, "v2": {"ANSEL_HIP_NLM_V2": "1"},
        "v4 no A1": {"ANSEL_NLM2_VARIANT": "16"}, "v4 no A2": {"ANSEL_NLM2_VARIANT": "32"},
        "v4 no row chain": {"ANSEL_NLM2_VARIANT": "64"}, "v4 no C": {"ANSEL_NLM2_VARIANT": "128"},
        "v4 only A1": {"ANSEL_NLM2_VARIANT": str(32 + 64 + 128)}, "v4 only A2": {"ANSEL_NLM2_VARIANT": str(16 + 64 + 128)},
        "v4 only C": {"ANSEL_NLM2_VARIANT": str(16 + 32)}, "v4 only barriers": {"ANSEL_NLM2_VARIANT": str(16 + 32 + 64 + 128)}})
    res["8256x2048 (72 x 64 chunks)"] = run(l, 8256, 2048, {"v4 fused (default here)": {}, "v2": {"ANSEL_HIP_NLM_V2": "1"}})
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
