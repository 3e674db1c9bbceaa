#!/usr/bin/env python3
"""Where the AMaZE workgroups spend their cycles (run on the GPU box): the measuring builds (ANSEL_HIP_AMAZE_TIMED) print the
cycles between the barriers of a step of amaze_stream (the full tiles, on chip: `[amaze_stream_timed] phase k`) and between
the stage stamps of amaze_tiles (the tiles the frame cuts: `[amaze_timed] stamp k`), averaged over the tiles; then the plain
launch is timed.

    python tools/amaze_stage_clocks.py [WxH]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the A/B switches exist in the measuring build only (python -m ansel_amd.build --measuring, before gpurun)
os.environ.setdefault("ANSEL_HIP_LIB", os.path.join(ROOT, "ansel_amd", "libansel_hip_measuring.so"))
import numpy as np  # noqa: E402

from ansel_amd import abi, lib, synth  # noqa: E402


def main():
    w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "6000x4000").split("x"))
    l = lib.init()
    tile = synth.bayer_mosaic(1024, 1024, seed=5).astype(np.float32)
    cfa = np.tile(tile, ((h + 1023) // 1024, (w + 1023) // 1024))[:h, :w]
    img = ((cfa - 512.0) / np.float32(synth.WHITE - 512)).astype(np.float32)
    din = lib.DeviceBuffer.from_numpy(0, img)
    dout = lib.DeviceBuffer(0, w * h * 16)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0, 0.0)

    def run():
        lib.check(l.dt_hip_iop_demosaic_process(0, C.byref(piece), C.byref(d), din.ptr, dout.ptr), "amaze")
        l.dt_hip_finish(0)
    run()
    if not os.environ.get("AMAZE_SKIP_TIMED"):
        os.environ["ANSEL_HIP_AMAZE_TIMED"] = "1"
        run()
        del os.environ["ANSEL_HIP_AMAZE_TIMED"]
    for blocks in (os.environ.get("AMAZE_BLOCKS", "512,768").split(",")):
        os.environ["ANSEL_HIP_AMAZE_BLOCKS"] = blocks
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            run()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("blocks %s: %.2f ms (%.0f MPix/s)" % (blocks, min(ts), w * h / 1e3 / min(ts)), flush=True)


if __name__ == "__main__":
    main()
