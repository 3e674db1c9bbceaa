#!/usr/bin/env python3
"""Which of the plane sharings of the reference's AMaZE buffer (amaze.cc:300-327) does the RESULT depend on?  (CPU only.)

The reference carves its tile planes out of one allocation and lets planes with disjoint lifetimes share memory; a few
stencils read words their logical plane did not write in this tile and see what the sharing partner left there
(DESIGN.md section 3).  The device kernel reproduces that by laying its slab out like the reference's buffer; a design that
keeps the planes on chip has to know WHICH reads those are.  The oracle can give one shared plane at a time storage of its
own (oracle_amaze_unshare); a frame that changes then shows where the sharing is part of the result.

    python tools/amaze_alias_probe.py > profiles/r02_amaze_alias_probe.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import checkers as ck  # noqa: E402
from ansel_amd import abi, synth  # noqa: E402

PAIRS = [(1, "dgrb0 / dgrb1 in vcdalt"), (2, "delp / delm / rbint in cddiffsq"), (4, "pmwt in delhvsqsum"),
         (8, "rbm / rbp in vcd"), (16, "second Nyquist flag plane in cddiffsq's bytes"), (32, "dgrb2 in dgintv"),
         (64, "(not a sharing) Nyquist refinement over the whole tile, not the flags' bounding box")]


def frame(w, h, seed):
    raw = synth.bayer_mosaic(w, h, seed=seed).astype(np.float32)
    cfa = ((raw - 512) / np.float32(synth.WHITE - 512) * np.float32(1.7)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    # fine checkerboards and stripes switch the Nyquist branches on
    cfa[h // 8: h // 2, w // 6: 2 * w // 3] *= (0.55 + 0.45 * ((xx[h // 8: h // 2, w // 6: 2 * w // 3] + yy[h // 8: h // 2, w // 6: 2 * w // 3]) & 1)).astype(np.float32)
    cfa[h // 2: 7 * h // 8, w // 4: 7 * w // 8] *= (0.6 + 0.4 * ((xx[h // 2: 7 * h // 8, w // 4: 7 * w // 8] >> 1) & 1)).astype(np.float32)
    return cfa


def run(o, piece, d, cfa, bits):
    o.oracle_amaze_unshare(C.c_uint(bits))
    out = np.zeros(cfa.shape + (4,), np.float32)
    assert ck.call(o, "oracle_demosaic", piece, d, cfa, out) == 0
    o.oracle_amaze_unshare(C.c_uint(0))
    return out


def main():
    o = ck.oracle()
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0)
    print("AMaZE: the frame under one plane at a time taken out of its sharing (oracle_amaze_unshare), against the reference's layout")
    for (w, h) in ((1504, 1000), (517, 389), (300, 200)):
        for filters in (0x94949494, 0x49494949, 0x61616161, 0x16161616):
            cfa = frame(w, h, seed=w + h)
            piece = abi.Piece.make(w, h, filters=filters, channels=1, processed_maximum=(1.5, 1.0, 1.2, 1.0))
            base = run(o, piece, d, cfa, 0)
            print("\nframe %d x %d, filters 0x%08x" % (w, h, filters))
            for bit, name in PAIRS:
                got = run(o, piece, d, cfa, bit)
                diff = got.view(np.uint32) != base.view(np.uint32)
                n = int(diff.sum())
                line = "  %-48s %8d values differ" % (name, n)
                if n:
                    ys, xs, cs = np.nonzero(diff)
                    rows = sorted(set((ys % 128).tolist()))
                    cols = sorted(set((xs % 128).tolist()))
                    last_row = int((ys == h - 1).sum()), int((xs == w - 1).sum())
                    line += "; channels %s; frame rows mod 128 in %s; columns mod 128 in %s; on the last frame row / column: %d / %d" % (
                        sorted(set(cs.tolist())), _ranges(rows), _ranges(cols), last_row[0], last_row[1])
                print(line)


def _ranges(v):
    out, start, prev = [], v[0], v[0]
    for x in v[1:]:
        if x != prev + 1:
            out.append((start, prev))
            start = x
        prev = x
    out.append((start, prev))
    return ", ".join("%d" % a if a == b else "%d-%d" % (a, b) for a, b in out)


if __name__ == "__main__":
    main()
