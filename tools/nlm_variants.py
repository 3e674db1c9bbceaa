#!/usr/bin/env python3
"""Timing experiment for nlm_chunks_v2 (run on the GPU box): denoise (non-local means) on one Lab frame under each
A/B switch of ansel_amd/csrc/nlm2_body.h (ANSEL_NLM2_VARIANT: 4-byte forms of the recurrences, steps switched off).
Variants with a step switched off compute garbage: this only measures where the time goes.

    python tools/nlm_variants.py [WxH] > gpurun_out/nlm_variants.json

Default frame 11648 x 2184: the 100 MP frame's width and chunk grid (72 x 56 chunks, which fit the four-table schedule)
at a quarter of its height."""
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


def main():
    w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "11648x2184").split("x"))
    l = lib.init()
    rng = np.random.default_rng(1)
    img = rng.random((h, w, 4), dtype=np.float32) * np.float32(100.0)
    din = lib.DeviceBuffer.from_numpy(0, img)
    dout = lib.DeviceBuffer(0, img.nbytes)
    piece = abi.Piece.make(w, h)
    d = abi.NlmeansData(2.0, 50.0, 0.5, 1.0)
    # (label, environment) -- ANSEL_NLM2_VARIANT bits: 16 no A1, 32 no A2, 64 no B, 128 no C, 256 no first row
    names = {"v1 (nlm_chunks_pipelined)": {"ANSEL_HIP_NLM_V1": "1"},
             "v3 shipped (nlm3_body.h)": {},
             "v3 no priority": {"ANSEL_NLM2_VARIANT": "512"}, "v3 B stores at the end": {"ANSEL_NLM2_VARIANT": "1024"},
             "v3 only A2": {"ANSEL_NLM2_VARIANT": "208"}, "v3 only B": {"ANSEL_NLM2_VARIANT": "176"},
             "v3 no A1": {"ANSEL_NLM2_VARIANT": "16"}, "v3 no A2": {"ANSEL_NLM2_VARIANT": "32"},
             "v3 no B": {"ANSEL_NLM2_VARIANT": "64"}, "v3 no C": {"ANSEL_NLM2_VARIANT": "128"},
             "v3 no A2, no B": {"ANSEL_NLM2_VARIANT": "96"}, "v3 no A1, no C": {"ANSEL_NLM2_VARIANT": "144"},
             "v3 only A1": {"ANSEL_NLM2_VARIANT": "224"}, "v3 only C": {"ANSEL_NLM2_VARIANT": "112"},
             "v3 only barriers": {"ANSEL_NLM2_VARIANT": "240"},
             "v2 (nlm2_body.h)": {"ANSEL_HIP_NLM_V2": "1"},
             "v2 two tables": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_DEEP": "0"},
             "v2 loose layout": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_LAYOUT": "loose"},
             "no A1": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_VARIANT": "16"}, "no A2": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_VARIANT": "32"},
             "no B": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_VARIANT": "64"}, "no C": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_VARIANT": "128"},
             "no A2, no B": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_VARIANT": "96"}, "no A1, no C": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_VARIANT": "144"},
             "only barriers": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_VARIANT": str(16 + 32 + 64 + 128 + 256)},
             "two tables, no A2, no B": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_DEEP": "0", "ANSEL_NLM2_VARIANT": "96"},
             "two tables, no A1, no C": {"ANSEL_HIP_NLM_V2": "1", "ANSEL_NLM2_DEEP": "0", "ANSEL_NLM2_VARIANT": "144"}}
    out = {"frame": [w, h], "variants": {}}
    for name, env in names.items():
        for k in ("ANSEL_HIP_NLM_V1", "ANSEL_HIP_NLM_V2", "ANSEL_NLM2_VARIANT", "ANSEL_NLM2_DEEP", "ANSEL_NLM2_LAYOUT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        times = []
        for rep in range(4):
            l.dt_hip_finish(0)
            t0 = time.perf_counter()
            lib.check(l.dt_hip_iop_nlmeans_process(0, C.byref(piece), C.byref(d), din.ptr, dout.ptr), "nlmeans")
            l.dt_hip_finish(0)
            times.append((time.perf_counter() - t0) * 1e3)
        out["variants"][name] = {"env": env, "ms": round(min(times[1:]), 3)}
        print("%-28s %8.3f ms" % (name, min(times[1:])), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
