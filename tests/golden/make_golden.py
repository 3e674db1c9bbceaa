#!/usr/bin/env python3
"""Generate tests/golden/golden.npz: the outputs of THE REFERENCE'S OWN CODE
(oracle/_ref/libansel_ref.so, built from /root/reference by `make -C oracle ref`) on the seeded
cases of tests/golden/cases.py.  Run in the build container, where /root/reference exists:

    python tests/golden/make_golden.py

The reference ships no golden image vectors for this path (its tests/integration directory holds
the runner only), so these are "outputs of the reference itself run here".  Only outputs are
stored; inputs are regenerated from their seeds."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import checkers as ck  # noqa: E402
import cases  # noqa: E402


def main():
    r = ck.ref()
    if r is None:
        raise SystemExit("oracle/_ref/libansel_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    out = {}
    threads = r.ref_get_num_threads()
    for name, op, piece, data, inp, shape in cases.cases():
        res = np.zeros(shape, np.float32)
        # modules with a thread-count dependent reduction are recorded single-threaded
        r.ref_set_num_threads(1 if (op in cases.SINGLE_THREAD_OPS or name in cases.SINGLE_THREAD_NAMES) else threads)
        if op == "develop_blend":  # in place: (module input, module output)
            res[...] = inp[1]
            inp = inp[0]
        assert ck.call(r, "ref_" + op, piece, data, np.ascontiguousarray(inp), res) == 0, name
        out[name] = res
    r.ref_set_num_threads(threads)
    img = [c for c in cases.cases() if c[0] == "exposure"][0][4]
    w, h = img.shape[1], img.shape[0]
    u16 = np.zeros(img.shape, np.uint16)
    u8 = np.zeros(img.shape, np.uint8)
    r.ref_export_convert_u16(w, h, ck.ptr(img), ck.ptr(u16))
    r.ref_export_convert_u8(w, h, ck.ptr(img), ck.ptr(u8))
    out["export_u16"] = u16
    out["export_u8"] = u8
    path = os.path.join(HERE, "golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d cases, %.1f KiB" % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
