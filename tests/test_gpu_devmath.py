"""-m gpu: the device build of devmath.h (glibc-exact powf/log2f/exp2f/expf) vs the host's libm."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hostlib():
    so = os.path.join(ROOT, "tests", "native", "libdevmath_host.so")
    assert os.path.exists(so), "run __graft_entry__.build() first"
    return C.CDLL(so)


def _args(n, seed):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = np.concatenate([bits, rng.random(n, dtype=np.float32) * 4.0, np.abs(bits)])
    y = np.concatenate([rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32),
                        rng.random(n, dtype=np.float32) * 6.0, rng.random(n, dtype=np.float32) * 16.0 - 8.0])
    return np.ascontiguousarray(x), np.ascontiguousarray(y)


@pytest.mark.parametrize("fn", ["powf", "log2f", "logf", "exp2f", "expf", "atan2f", "hypotf", "sinf", "cosf", "fmodf"])
def test_device_libm_matches_host_libm(fn):
    n = 2_000_000
    x, y = _args(n, 17)
    if fn == "exp2f":
        x = np.concatenate([x[:n], np.random.default_rng(3).random(2 * n, dtype=np.float32) * 300.0 - 160.0])
    if fn in ("atan2f", "hypotf"):
        rng = np.random.default_rng(5)
        x = np.concatenate([x[:n], (rng.random(n, dtype=np.float32) - 0.5).astype(np.float32), (rng.random(n, dtype=np.float32) * 2e-3 - 1e-3).astype(np.float32)])
        y = np.concatenate([y[:n], (rng.random(n, dtype=np.float32) - 0.5).astype(np.float32), (rng.random(n, dtype=np.float32) * 400.0 - 200.0).astype(np.float32)])
    if fn in ("sinf", "cosf"):
        rng = np.random.default_rng(6)
        x = np.concatenate([x[:n], (rng.random(n, dtype=np.float32) * 13.0 - 6.5).astype(np.float32),
                            (rng.random(n, dtype=np.float32) * 2e7 - 1e7).astype(np.float32)])
    if fn == "fmodf":
        rng = np.random.default_rng(7)
        x = np.concatenate([x[:n], (rng.random(n, dtype=np.float32) * 4.0 - 1.0).astype(np.float32), (rng.random(n, dtype=np.float32) * 2000.0 - 1000.0).astype(np.float32)])
        y = np.concatenate([y[:n], np.ones(n, np.float32), (rng.random(n, dtype=np.float32) * 7.0 + 0.01).astype(np.float32)])
    if fn == "logf":
        x = np.concatenate([x[:n], np.random.default_rng(8).random(2 * n, dtype=np.float32)])  # the Box-Muller deviates
    if fn == "expf":
        x = np.concatenate([x[:n], np.random.default_rng(4).random(2 * n, dtype=np.float32) * 205.0 - 110.0])
    h = hc.hip()
    dx = lib.DeviceBuffer.from_numpy(0, x)
    dy = lib.DeviceBuffer.from_numpy(0, y)
    do = lib.DeviceBuffer(0, x.nbytes)
    f = getattr(h, "dt_hip_test_" + fn)
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.check(f(0, dx.ptr, dy.ptr, do.ptr, x.size), fn)
    got = do.to_numpy(x.shape, np.float32)
    exp = np.empty_like(x)
    hl = _hostlib()
    if fn == "fmodf":
        hl.libm_fmodf(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), exp.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    elif fn in ("atan2f", "hypotf"):
        # first operand of the hook is y for atan2f(y, x)
        getattr(hl, "libm_" + fn)(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), exp.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    elif fn == "powf":
        hl.libm_powf(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), exp.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    else:
        getattr(hl, "libm_" + fn)(x.ctypes.data_as(C.c_void_p), exp.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
    assert same.all(), "%s: %d of %d results differ from libm" % (fn, int((~same).sum()), x.size)
