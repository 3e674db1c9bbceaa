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


# ---- ieee_inrange.h: division / reciprocal / square root without the range scaffolding (round 5) -------------------------
def _hook(name):
    f = getattr(hc.hip(), "dt_hip_test_" + name)
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    return f


def _run_hook(name, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(x if y is None else y, np.float32)
    dx, dy = lib.DeviceBuffer.from_numpy(0, x), lib.DeviceBuffer.from_numpy(0, y)
    do = lib.DeviceBuffer(0, x.nbytes)
    lib.check(_hook(name)(0, dx.ptr, dy.ptr, do.ptr, x.size), name)
    return do.to_numpy(x.shape, np.float32)


def _floats(rng, n, e_lo, e_hi, signed=True):
    """n floats with uniformly drawn exponents in [e_lo, e_hi] (unbiased), random 23-bit mantissas (the all-zero and all-one
    mantissas over-represented: the rounding boundaries)"""
    e = rng.integers(e_lo, e_hi + 1, n, dtype=np.int64)
    m = rng.integers(0, 1 << 23, n, dtype=np.int64)
    pick = rng.integers(0, 8, n)
    m = np.where(pick == 0, 0, np.where(pick == 1, (1 << 23) - 1, np.where(pick == 2, 1, m)))
    s = rng.integers(0, 2, n, dtype=np.int64) if signed else np.zeros(n, np.int64)
    return (((s << 31) | ((e + 127) << 23) | m).astype(np.uint32)).view(np.float32)


def test_inrange_division_is_the_ieee_division_on_its_domain():
    """div_core(a, b), ansel_amd/csrc/ieee_inrange.h: b normal below 2^126, |a| >= 2^-103, -125 <= e(a) - e(b) < 96 -- the
    domain in which v_div_scale / v_div_fmas / v_div_fixup do nothing.  Against the host's (correctly rounded) division."""
    rng = np.random.default_rng(51)
    n = 4_000_000
    b = _floats(rng, n, -126, 125)
    eb = ((b.view(np.uint32) >> 23) & 0xff).astype(np.int64) - 127
    lo = np.maximum(-103, eb - 125)
    hi = np.minimum(127, eb + 95)
    ea = lo + (rng.random(n) * (hi - lo + 1)).astype(np.int64)
    ea = np.clip(ea, lo, hi)
    a = _floats(rng, n, 0, 0)
    a = ((a.view(np.uint32) & 0x807fffff) | ((ea + 127).astype(np.uint32) << 23)).view(np.float32)
    a = np.concatenate([a, (rng.random(n, dtype=np.float32) * 2 - 1)])
    b = np.concatenate([b, rng.random(n, dtype=np.float32) + np.float32(1e-8)])
    got = _run_hook("div_core", a, b)
    with np.errstate(all="ignore"):
        exp = (a / b).astype(np.float32)
    bad = got.view(np.uint32) != exp.view(np.uint32)
    assert not bad.any(), "%d of %d quotients differ, first a=%r b=%r got=%r exp=%r" % (
        int(bad.sum()), a.size, a[bad][0], b[bad][0], got[bad][0], exp[bad][0])
    # what the diffusion kernel feeds it (diffuse.hip ratio2_rgb()): |h| <= 2^64 -- any magnitude below, subnormals and zeros
    # included -- over divisors in [1e-8, 2^64].  Outside div_core()'s own domain (small numerators, subnormal quotients) the
    # quotient may differ in its last bits, and the kernel only ever SQUARES it: the squares must be the same bits
    b2 = np.abs(_floats(rng, n, -27, 63)) + np.float32(1e-8)
    a2 = np.concatenate([_floats(rng, n // 2, -126, 64), (1e-39 * rng.random(n // 4)).astype(np.float32),
                         _floats(rng, n // 4, -110, -95)])
    got = _run_hook("div_core", a2, b2)
    with np.errstate(all="ignore"):
        exp = (a2 / b2).astype(np.float32)
        sq_got, sq_exp = (got * got).astype(np.float32), (exp * exp).astype(np.float32)
    bad = sq_got.view(np.uint32) != sq_exp.view(np.uint32)
    assert not bad.any(), "%d squared ratios differ, first h=%r safe=%r got=%r exp=%r" % (
        int(bad.sum()), a2[bad][0], b2[bad][0], got[bad][0], exp[bad][0])
    # zero numerators: a zero (its sign is not the quotient's: the callers square it)
    z = _run_hook("div_core", np.array([0.0, -0.0, 0.0, -0.0], np.float32), np.array([1e-8, 1e-8, -3.5, 2.0 ** 100], np.float32))
    assert ((z.view(np.uint32) & 0x7fffffff) == 0).all()


def test_division_by_a_shared_divisor():
    """div_uniform(n, d, rcp_refined(d)), ieee_inrange.h: the quotient of nlm3_body.h's CENTER weight, d = 1 + center_weight in
    [1, 2^20].  Against the host's division: every bit for |n| in [2^-103, 2^96 d) and for zero, infinite and NaN numerators;
    outside that interval the kernel only needs what the weight makes of the quotient -- w = 2^-max(0, q s - 2) for a sharpness
    s in [2^-60, 2^60]: the same bits (1 for the small quotients, 0 for the large ones)."""
    rng = np.random.default_rng(61)
    n = 4_000_000
    d = np.concatenate([(1.0 + rng.random(n // 2) * 3.0).astype(np.float32), np.abs(_floats(rng, n // 2, 0, 19)),
                        np.array([1.0, 1.1, 2.0, 1048576.0], np.float32)])
    m = d.size
    a = np.concatenate([_floats(rng, m // 2, -103, 95), np.abs(_floats(rng, m - m // 2, -40, 40))])
    got = _run_hook("div_uniform", a, d)
    with np.errstate(all="ignore"):
        exp = (a / d).astype(np.float32)
    bad = got.view(np.uint32) != exp.view(np.uint32)
    assert not bad.any(), "%d of %d quotients differ, first n=%r d=%r got=%r exp=%r" % (
        int(bad.sum()), a.size, a[bad][0], d[bad][0], got[bad][0], exp[bad][0])
    # the numerators the five operations do not take
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 3.0e38, -3.0e38], np.float32)
    for dv in (1.0, 1.1, 7.5, 1048576.0):
        g = _run_hook("div_uniform", sp, np.full(sp.shape, dv, np.float32))
        with np.errstate(all="ignore"):
            e = (sp / np.float32(dv)).astype(np.float32)
        same = (g.view(np.uint32) == e.view(np.uint32)) | (np.isnan(g) & np.isnan(e))
        assert same.all(), (dv, g, e)
    # beyond the interval: the weight's bits
    a2 = np.concatenate([np.abs(_floats(rng, n // 2, -149 + 23, -100)), (1e-39 * rng.random(n // 4)).astype(np.float32),
                         np.abs(_floats(rng, n // 4, 96, 127)), -np.abs(_floats(rng, n // 8, -140 + 23, -90))])
    d2 = (1.0 + rng.random(a2.size) * 9.0).astype(np.float32)
    got = _run_hook("div_uniform", a2, d2)
    with np.errstate(all="ignore"):
        exp = (a2 / d2).astype(np.float32)

        def weight(q, s):
            t = np.maximum(np.float32(0.0), (q * np.float32(s) - np.float32(2.0)).astype(np.float32))
            v = (t * np.float32(-8388608.0)).astype(np.float32)
            cv = np.where(np.isfinite(v) & (np.abs(v) < 2147483648.0), np.nan_to_num(v, posinf=0, neginf=0).astype(np.int64), -(1 << 31))
            k0 = ((0x3f800000 + cv) & 0xffffffff).astype(np.uint32).view(np.int32)
            return np.where(k0 >= 0x800000, k0, 0)
        for s_ in (2.0 ** -60, 1.3, 2.0 ** 60):
            assert np.array_equal(weight(got, s_), weight(exp, s_)), s_


def test_inrange_reciprocal_and_square_root():
    rng = np.random.default_rng(52)
    n = 4_000_000
    b = np.concatenate([_floats(rng, n, -126, 125), _floats(rng, n, -48, 64, signed=False), np.ones(8, np.float32)])
    got = _run_hook("rcp_core", b)
    exp = (np.float32(1.0) / b).astype(np.float32)
    bad = got.view(np.uint32) != exp.view(np.uint32)
    assert not bad.any(), "%d reciprocals differ, first b=%r got=%r exp=%r" % (int(bad.sum()), b[bad][0], got[bad][0], exp[bad][0])
    x = np.concatenate([_floats(rng, n, -96, 127, signed=False), np.zeros(8, np.float32),
                        (rng.random(n, dtype=np.float32) ** 4).astype(np.float32) * np.float32(1e-3) + np.float32(2.0 ** -96),
                        np.array([2.0 ** -96, np.finfo(np.float32).max, 1.0, 4.0, 2.0], np.float32)])
    got = _run_hook("sqrt_core", x)
    exp = np.sqrt(x).astype(np.float32)
    bad = got.view(np.uint32) != exp.view(np.uint32)
    assert not bad.any(), "%d square roots differ, first x=%r got=%r exp=%r" % (int(bad.sum()), x[bad][0], got[bad][0], exp[bad][0])


def test_inrange_guard_is_the_stated_interval():
    """zero_or_above_2m96(): +0 or 2^-96 <= x < inf, nothing else (NaN, infinities, negatives, -0, everything below 2^-96 fail)"""
    rng = np.random.default_rng(53)
    x = np.concatenate([rng.integers(0, 2**32, 3_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32),
                        _floats(rng, 1_000_000, -100, -92), _floats(rng, 200_000, -126, -110), (1e-39 * rng.random(200_000)).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 2.0 ** -96,
                        np.nextafter(np.float32(2.0 ** -96), np.float32(0)), np.finfo(np.float32).max, 1e-45], np.float32)])
    got = _run_hook("zero_or_above_2m96", x) != 0
    with np.errstate(all="ignore"):
        exp = (x.view(np.uint32) == 0) | ((x >= np.float32(2.0 ** -96)) & np.isfinite(x))
    assert (got == exp).all()
