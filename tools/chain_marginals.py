#!/usr/bin/env python3
"""Marginal cost of the stages of the fused RGBA chain (rgb_chain, pipe_fused.hip) on a 100 MP frame: the light
pipe with one stage changed at a time, timed over the executor.  A measurement aid for DESIGN.md section 4.1."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from ansel_amd import filmic, lib, params, pipe, synth
    lib.init()
    w, h = bench.frame_size(sys.argv[1] if len(sys.argv) > 1 else "100MP")
    lut_host = params.srgb_encode_lut()
    lut = torch.from_numpy(lut_host).to("cuda:0")
    coeffs = params.unbounded_coeffs(lut_host)
    raw = torch.from_numpy(synth.bayer_mosaic_tiled(w, h, seed=1).view(np.int16)).to("cuda:0")
    out = torch.empty((h, w, 4), dtype=torch.int16, device="cuda:0")

    def variant(name):
        nodes = pipe.light_pipe_nodes(w, h, lut.data_ptr(), float(lut_host[0]), coeffs, with_filmic=name != "no_filmic",
                                      filmic=filmic.default_data())
        if name == "linear_colorout":
            for n in nodes:
                if n.op == "colorout":
                    n.data = params.conversion(params.SRGB_OUT @ params.WORK_IN)
        if name == "no_calibration":
            nodes = [n for n in nodes if n.op != "channelmixerrgb"]
        return nodes

    for name in ("default", "linear_colorout", "no_filmic", "no_calibration"):
        p = pipe.DevicePipe(0, variant(name))
        for _ in range(2):
            p.process(raw.data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            p.process(raw.data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        print(json.dumps({"variant": name, "ms_per_frame": round((time.perf_counter() - t0) / 5 * 1e3, 3)}), flush=True)
        p.close()


if __name__ == "__main__":
    main()
