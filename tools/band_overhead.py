#!/usr/bin/env python3
"""What row bands cost on the device: ONE frame cut into N bands, all bands run in lockstep on the one GPU of
the box through the entry points a multi-GPU job uses (ansel_amd.tiled.process_bands_locally: the collectives
become device copies), timed against the unsplit executor run of the same frame.

    t_bands / t_whole  = the redundant work of the halos (a band runs its stencil modules on [halo][rows][halo])
    t_bands / N        = what one rank of an N-GPU job spends in kernels per frame (no communication in it)

Prints one JSON line per (pipe, N).  This is a measurement aid for DESIGN.md section 6, not the bench."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="100MP")
    ap.add_argument("--pipe", default="denoise", choices=("light", "denoise"))
    ap.add_argument("--bands", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    from ansel_amd import lib, params, pipe, synth, tiled
    lib.init()
    w, h = bench.frame_size(args.size)
    lut_host = params.srgb_encode_lut()
    lut = torch.from_numpy(lut_host).to("cuda:0")
    nodes = bench.build_pipe(w, h, lut.data_ptr(), lut_host, bench.have_filmic(), args.pipe)
    raw_host = synth.bayer_mosaic_tiled(w, h, seed=1)
    p = pipe.DevicePipe(0, nodes)
    engine = tiled.HipBandEngine(p, "cuda:0")

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    raw = torch.from_numpy(raw_host.view(np.int16)).to("cuda:0")
    out = torch.empty((h, w, 4), dtype=torch.int16, device="cuda:0")
    t_whole = timed(lambda: p.process(raw.data_ptr(), out.data_ptr()))
    whole = out.cpu()
    del out
    for n in [int(x) for x in args.bands.split(",")]:
        bands = tiled.plan_bands(w, h, n, tiled.pipe_demosaic_method(nodes))
        ins = [raw[b.row0:b.row0 + b.rows] for b in bands]  # contiguous row slices of the resident frame
        outs = [torch.empty((b.rows, w, 4), dtype=torch.int16, device="cuda:0") for b in bands]
        t = timed(lambda: tiled.process_bands_locally(engine, bands, [x.data_ptr() for x in ins],
                                                      [x.data_ptr() for x in outs], w))
        same = bool(torch.equal(torch.cat([x.cpu() for x in outs], dim=0), whole))
        print(json.dumps({"pipe": args.pipe, "size": args.size, "bands": n, "whole_ms": round(t_whole, 2),
                          "all_bands_ms": round(t, 2), "per_band_ms": round(t / n, 2),
                          "redundancy": round(t / t_whole, 3), "identical_to_unsplit": same}), flush=True)
        del outs
    p.close()


if __name__ == "__main__":
    main()
