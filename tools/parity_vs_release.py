#!/usr/bin/env python3
"""How far is the device from the binary users run today?  (Run on the GPU box.)

The parity bar of this repository is the reference's sources compiled STRICT (-O2 -fno-fast-math -ffp-contract=off): the device
equals that build bit for bit (tests/).  The reference SHIPS -O3 -ffast-math -ffp-contract=fast with OpenMP (CMakeLists.txt:239-272);
oracle/_ref/libansel_ref_fast.so is that build of the same sources.  This tool measures the distance between the two worlds,
per module and for the light / full export pipe on a 24 MP frame:

  * per module: the module's input is the ORACLE chain's intermediate at that stage (so the figures are per module, not
    accumulated; the oracle is the strict build's arithmetic with the canonical orders of DESIGN.md section 3 where the
    reference's own result depends on its thread count -- the RCD scratch columns, the wavelets' sum of detail^2, the
    bilateral grid's slices); device output against the release build's output on the same input -- ULP histogram over the
    three colour channels (alpha apart), the largest absolute difference -- , whether the device equals the oracle (it must),
    and the same histogram for the strict reference build (multi-threaded, as it comes) against the release build;
  * per pipe: the device's exported RGBA u16 against the release build's module-by-module chain -- histogram of |difference|
    in LSB of the 16-bit output.

TEST INFRASTRUCTURE uses the checkers (tests/checkers.py); nothing here is on a product path.

    python tools/parity_vs_release.py [--size 24MP] > gpurun_out/parity_vs_release_flags.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

CFA_OPS = ("rawprepare", "temperature", "highlights")
ULP_BINS = [0, 1, 2, 4, 8, 16, 64, 1024, 1 << 20]


def ulp_hist(a, b):
    import checkers as ck
    out = {}
    d = ck.ulp_diff(a, b)
    n = d.size
    edges = ULP_BINS
    for k, lo in enumerate(edges):
        hi = edges[k + 1] if k + 1 < len(edges) else None
        name = "0" if lo == 0 else ("%d" % lo if hi is not None and hi == lo + 1 else ("%d..%d" % (lo, hi - 1) if hi is not None else ">=%d" % lo))
        cnt = int(((d >= lo) & (d < hi)).sum()) if hi is not None else int((d >= lo).sum())
        if lo == 0:
            cnt = int((d == 0).sum())
            name = "0"
        out[name] = round(cnt / float(n), 6)
    return out, int(d.max())


def stats(got, want):
    """colour channels and alpha apart; float planes"""
    if got.ndim == 3:
        h, m = ulp_hist(got[..., :3], want[..., :3])
        e = {"ulp_hist_rgb": h, "max_ulp_rgb": m,
             "max_abs_rgb": float(np.nanmax(np.abs(got[..., :3].astype(np.float64) - want[..., :3].astype(np.float64))))}
        ha, ma = ulp_hist(got[..., 3], want[..., 3])
        e["alpha_identical"] = ma == 0
        return e
    h, m = ulp_hist(got, want)
    return {"ulp_hist": h, "max_ulp": m, "max_abs": float(np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64))))}


def lsb_hist(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))[..., :3]
    n = float(d.size)
    return {"0": round(float((d == 0).sum()) / n, 6), "1": round(float((d == 1).sum()) / n, 6), "2": round(float((d == 2).sum()) / n, 6),
            "3..7": round(float(((d >= 3) & (d < 8)).sum()) / n, 6), ">=8": round(float((d >= 8).sum()) / n, 6), "max": int(d.max())}


def cpu_module(lib, prefix, n, src, w, h):
    import checkers as ck
    if n.op == "export_u16":
        out = ck.aligned_empty((h, w, 4), np.uint16)
        getattr(lib, prefix + "export_convert_u16")(w, h, ck.ptr(src), ck.ptr(out))
        return out
    dst = ck.aligned_empty((h, w) if n.op in CFA_OPS else (h, w, 4), np.float32)
    dst[...] = 0
    rc = ck.call(lib, prefix + n.op, n.piece, n.data, src, dst)
    assert rc == 0, n.op
    return dst


def device_module(pipe_mod, lib_mod, n, src, w, h):
    """one module on the device through the C-ABI (pipe.run_nodes: the entry a process_cl() stub calls)"""
    din = lib_mod.DeviceBuffer.from_numpy(0, np.ascontiguousarray(src))
    if n.op == "export_u16":
        dout = lib_mod.DeviceBuffer(0, w * h * 8)
        shape, dt = (h, w, 4), np.uint16
    elif n.op in CFA_OPS:
        dout = lib_mod.DeviceBuffer(0, w * h * 4)
        shape, dt = (h, w), np.float32
    else:
        dout = lib_mod.DeviceBuffer.from_numpy(0, np.zeros((h, w, 4), np.float32))
        shape, dt = (h, w, 4), np.float32
    pipe_mod.run_nodes(0, [n], [din.ptr, dout.ptr])
    assert lib_mod.load().dt_hip_finish(0) == 1
    out = dout.to_numpy(shape, dt)
    din.release()
    dout.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="24MP")
    ap.add_argument("--no-device", action="store_true", help="the strict reference build stands in for the device (a CPU-only dry run)")
    args = ap.parse_args()
    import checkers as ck
    from ansel_amd import filmic, params, pipe, synth
    strict, fast, canon = ck.ref(), ck.ref(fast=True), ck.oracle()
    if strict is None or fast is None or canon is None:
        raise SystemExit("oracle/_ref/libansel_ref.so / libansel_ref_fast.so missing: make -f oracle/Makefile ref (needs /root/reference)")
    dev = not args.no_device
    lib_mod = None
    if dev:
        from ansel_amd import lib as lib_mod
        lib_mod.init()
    w, h = synth.SIZES[args.size] if args.size in synth.SIZES else map(int, args.size.split("x"))
    lut = params.srgb_encode_lut()
    coeffs = params.unbounded_coeffs(lut)
    d_lut = lib_mod.DeviceBuffer.from_numpy(0, lut) if dev else None

    def nodes_of(which, lut_ptr):
        if which == "light":
            return pipe.light_pipe_nodes(w, h, lut_ptr, float(lut[0]), coeffs, with_filmic=True, filmic=filmic.default_data())
        return pipe.denoise_pipe_nodes(w, h, lut_ptr, float(lut[0]), coeffs, filmic=filmic.default_data(), diffuse_iterations=2,
                                       with_nlmeans=True, with_bilat=True)
    raw = synth.bayer_mosaic_tiled(w, h, seed=2)
    res = {"frame": [w, h], "device_in_the_loop": dev,
           "oracle": "oracle/liboracle.so: the strict build's arithmetic, canonical orders where the reference depends on its thread count (the parity bar)",
           "strict_build": "oracle/_ref/libansel_ref.so: -O2 -fno-fast-math -ffp-contract=off, OpenMP as it comes",
           "release_build": "oracle/_ref/libansel_ref_fast.so: -O3 -ffast-math -ffp-contract=fast, OpenMP on %d threads (what the reference ships, "
                            "CMakeLists.txt:239-272)" % (os.cpu_count() or 1),
           "modules": {}, "pipes": {}}
    host_nodes = nodes_of("full", lut.ctypes.data)
    dev_nodes = nodes_of("full", d_lut.ptr) if dev else host_nodes
    # ---- per module, on the strict chain's intermediates
    src = raw
    t0 = time.time()
    for hn, dn in zip(host_nodes, dev_nodes):
        want = cpu_module(canon, "oracle_", hn, src, w, h)
        want_strict = cpu_module(strict, "ref_", hn, src, w, h)
        got_fast = cpu_module(fast, "ref_", hn, src, w, h)
        got_dev = device_module(pipe, lib_mod, dn, src, w, h) if dev else want
        if hn.op == "export_u16":
            e = {"device_vs_release_lsb": lsb_hist(got_dev, got_fast), "strict_vs_release_lsb": lsb_hist(want_strict, got_fast),
                 "device_equals_oracle": bool(np.array_equal(got_dev, want)), "strict_build_equals_oracle": bool(np.array_equal(want_strict, want))}
        else:
            e = {"device_vs_release": stats(got_dev, got_fast), "strict_vs_release": stats(want_strict, got_fast),
                 "device_equals_oracle": bool(np.array_equal(got_dev.view(np.uint32), want.view(np.uint32))),
                 # False where the reference is not a function of its input (its threads): RCD, the wavelets, the bilateral grid
                 "strict_build_equals_oracle": bool(np.array_equal(want_strict.view(np.uint32), want.view(np.uint32)))}
        res["modules"][hn.op] = e
        print("%-16s device==oracle %s  %s" % (hn.op, e["device_equals_oracle"], json.dumps(e.get("device_vs_release", e.get("device_vs_release_lsb")))[:200]),
              file=sys.stderr, flush=True)
        src = want
    # ---- per pipe: exported words, device pipe against the release build's chain
    for which in ("light", "full"):
        hn = nodes_of(which, lut.ctypes.data)
        src = raw
        for n in hn:
            src = cpu_module(fast, "ref_", n, src, w, h)
        fast_out = src
        src = raw
        for n in hn:
            src = cpu_module(strict, "ref_", n, src, w, h)
        strict_out = src
        src = raw
        for n in hn:
            src = cpu_module(canon, "oracle_", n, src, w, h)
        canon_out = src
        if dev:
            import torch
            dn = nodes_of(which, d_lut.ptr)
            p = pipe.DevicePipe(0, dn, fusion=True)
            d_in = lib_mod.DeviceBuffer.from_numpy(0, raw)
            d_out = lib_mod.DeviceBuffer(0, w * h * 8)
            p.process(d_in.ptr, d_out.ptr)
            assert lib_mod.load().dt_hip_finish(0) == 1
            dev_out = d_out.to_numpy((h, w, 4), np.uint16)
            p.close()
        else:
            dev_out = canon_out
        res["pipes"][which] = {"modules": [n.op for n in hn], "device_vs_release_lsb": lsb_hist(dev_out, fast_out),
                               "strict_vs_release_lsb": lsb_hist(strict_out, fast_out),
                               "device_vs_strict_build_lsb": lsb_hist(dev_out, strict_out),
                               "device_equals_oracle_chain": bool(np.array_equal(dev_out, canon_out))}
        print("pipe %-6s %s" % (which, json.dumps(res["pipes"][which])), file=sys.stderr, flush=True)
    res["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
