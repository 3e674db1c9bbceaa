"""Shared helpers for the -m gpu parity tests: run a module through the C-ABI of libansel_hip
on device 0 and through the CPU checkers (tests/checkers.py) on the same seeded input."""
import ctypes as C

import numpy as np

import checkers as ck
from ansel_amd import abi, lib

_state = {}


def hip():
    if "lib" not in _state:
        _state["lib"] = lib.init()  # raises loudly when the .so or the GPU is missing
    return _state["lib"]


def run_hip(fn_name, piece, data, inp, out_shape, out_dtype=np.float32, pre_fill=None):
    """upload `inp`, call dt_hip_iop_<fn>_process(devid 0, piece, data, in, out), download"""
    h = hip()
    din = lib.DeviceBuffer.from_numpy(0, inp)
    out_nbytes = int(np.prod(out_shape)) * np.dtype(out_dtype).itemsize
    if pre_fill is not None:
        dout = lib.DeviceBuffer.from_numpy(0, pre_fill)
    else:
        dout = lib.DeviceBuffer.from_numpy(0, np.zeros(out_shape, dtype=out_dtype))
    fn = getattr(h, fn_name)
    rc = fn(0, C.byref(piece), C.byref(data), din.ptr, dout.ptr)
    lib.check(rc, fn_name)
    assert h.dt_hip_finish(0) == 1, h.dt_hip_last_error()
    out = dout.to_numpy(out_shape, out_dtype)
    din.release()
    dout.release()
    return out


def run_cpu(which, name, piece, data, inp, out_shape, out_dtype=np.float32):
    """which: 'ref' (the reference's own code, oracle/_ref) or 'oracle' (restatement)"""
    l = ck.ref() if which == "ref" else ck.oracle()
    if l is None:
        return None
    out = np.zeros(out_shape, dtype=out_dtype)
    rc = ck.call(l, ("ref_" if which == "ref" else "oracle_") + name, piece, data, np.ascontiguousarray(inp), out)
    assert rc == 0
    return out


def assert_bit_exact(a, b, what, mask=None):
    d = ck.ulp_diff(a, b) if a.dtype == np.float32 else (a.astype(np.int64) != b.astype(np.int64)).astype(np.int64)
    if mask is not None:
        d = d * (mask == 0)
    n = int((d > 0).sum())
    assert n == 0, "%s: %d of %d values differ, max %d ulp" % (what, n, d.size, int(d.max()))


def checkers_available():
    out = []
    if ck.ref() is not None:
        out.append("ref")
    if ck.oracle() is not None:
        out.append("oracle")
    return out
