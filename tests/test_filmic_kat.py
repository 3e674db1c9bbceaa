"""not gpu: the known-answer tests of the reference's only hot-path unit test,
tests/unittests/iop/test_filmicrgb.c (clamp_simd :89-106, pixel_rgb_norm_power :108-178,
get_pixel_norm :180-250, log_tonemapping :252-320), run against the oracle restatement with the
reference's own synthetic generators (tests/unittests/util/testimg.c:109-268) and its tolerance
E = 1e-6 (test_filmicrgb.c:59); filmic_desaturate_v1 :350-455 and linear_saturation :459-520 of the 2019-2020
colour sciences included."""
import ctypes as C

import numpy as np
import pytest

import checkers as ck

E = 1e-6
STD_WIDTH = 16        # TESTIMG_STD_WIDTH
DYN_RANGE_EV = 15.0   # TESTIMG_STD_DYN_RANGE_EV
f32 = np.float32


def val_to_exp(val):  # testimg_val_to_exp()
    return f32(np.exp2(f32(DYN_RANGE_EV) * (f32(val) - f32(1.0))))


def val_to_log(val):  # testimg_val_to_log()
    return f32(1.0) - f32(np.log2(f32(1.0) / f32(val))) / f32(DYN_RANGE_EV)


def grey_space(width=STD_WIDTH):
    return [val_to_exp(f32(x) / f32(width - 1)) for x in range(width)]


def rgb_space(width=STD_WIDTH):
    tmp = grey_space(width)
    return [(tmp[x], tmp[y // width], tmp[y % width]) for y in range(width * width) for x in range(width)]


def grey_max_dr():
    fi = np.finfo(np.float32)
    return [f32(fi.tiny), f32(1e-20), f32(1e-10), f32(1e-5), f32(1e-1), f32(1.0), f32(1e5), f32(1e10), f32(1e20), f32(fi.max)]


def grey_max_dr_neg():
    return [-v for v in reversed(grey_max_dr())] + [f32(-0.0)]


@pytest.fixture(scope="module")
def o(oracle_lib):
    oracle_lib.oracle_kat_clamp_simd.restype = C.c_float
    oracle_lib.oracle_kat_clamp_simd.argtypes = [C.c_float]
    oracle_lib.oracle_kat_pixel_norm.restype = C.c_float
    oracle_lib.oracle_kat_pixel_norm.argtypes = [C.POINTER(C.c_float), C.c_int]
    oracle_lib.oracle_kat_log_tonemapping.restype = C.c_float
    oracle_lib.oracle_kat_log_tonemapping.argtypes = [C.c_float] * 4
    oracle_lib.oracle_kat_desaturate_v1.restype = C.c_float
    oracle_lib.oracle_kat_desaturate_v1.argtypes = [C.c_float] * 4
    oracle_lib.oracle_kat_linear_saturation.restype = C.c_float
    oracle_lib.oracle_kat_linear_saturation.argtypes = [C.c_float] * 3
    return oracle_lib


def _norm(o, p, variant):
    px = (C.c_float * 4)(p[0], p[1], p[2], 2.0)  # p[3] = 2.0: alpha must have no influence
    return o.oracle_kat_pixel_norm(px, variant)


def test_clamp_simd(o):
    x = f32(-0.5)
    while x <= 1.5:
        exp = 0.0 if x < 0 else (1.0 if x > 1 else float(x))
        assert abs(o.oracle_kat_clamp_simd(float(x)) - exp) <= E
        x = f32(x + f32(0.1))


def test_pixel_rgb_norm_power(o):
    for p in rgb_space():
        norm = _norm(o, p, 3)
        num = p[0] * p[0] * p[0] + p[1] * p[1] * p[1] + p[2] * p[2] * p[2]
        den = p[0] * p[0] + p[1] * p[1] + p[2] * p[2]
        assert abs(norm - float(num / den)) <= E
        assert 0.0 < norm <= 1.0 + 1e-6
    for v in grey_space():
        assert abs(_norm(o, (v, v, v), 3) - float(v)) <= E
    for v in grey_max_dr():
        if 1e-6 < v < 1e6:
            assert 0.0 < _norm(o, (v, v, v), 3) <= np.finfo(np.float32).max
    for v in grey_max_dr_neg():
        n = _norm(o, (v, v, v), 3)
        if 1e-6 < abs(v) < 1e6:
            assert 0.0 < n <= np.finfo(np.float32).max
        if v == 0:
            assert abs(n) <= np.finfo(np.float32).tiny


def test_get_pixel_norm_max_rgb(o):
    for p in rgb_space():
        norm = _norm(o, p, 1)
        assert abs(norm - float(max(p))) <= E
        assert 0.0 < norm <= 1.0 + E
    for v in grey_space():
        assert abs(_norm(o, (v, v, v), 1) - float(v)) <= E
    for v in grey_max_dr():
        assert 0.0 < _norm(o, (v, v, v), 1) <= np.finfo(np.float32).max
    for v in grey_max_dr_neg():
        assert _norm(o, (v, v, v), 1) <= np.finfo(np.float32).max


def test_log_tonemapping(o):
    grey = f32(0.1845)
    dyn = f32(DYN_RANGE_EV)
    black = f32(np.log2(f32(1.0) / grey)) - dyn
    for v in grey_space():
        ret = o.oracle_kat_log_tonemapping(float(v), float(grey), float(black), float(dyn))
        exp = float(val_to_log(v))
        assert abs(ret - (0.0 if exp < 0 else exp)) <= E
    for v in grey_space():
        ret = o.oracle_kat_log_tonemapping(float(v), float(grey / f32(2.0)), float(black), float(dyn))
        exp = float(val_to_log(v * f32(2.0)))
        exp = 0.0 if exp < 0 else (1.0 if exp > 1 else exp)
        assert abs(ret - exp) <= E
    for v in grey_max_dr() + grey_max_dr_neg():
        ret = o.oracle_kat_log_tonemapping(float(v), float(grey), float(black), float(dyn))
        assert 0.0 <= ret <= 1.0


def _saturation_gui_to_internal(percent):  # test_filmicrgb.c:337-347, "copied from filmicrgb.c"
    return f32(2.0) * f32(percent) / f32(100.0) + f32(1.0)


def test_filmic_desaturate_v1(o):
    """test_filmicrgb.c:350-455; as there, the sigmas come from a latitude of 0.2 on both sides"""
    sigma = f32(np.power(f32(0.2) / f32(3.0), f32(2.0)))
    saturation = _saturation_gui_to_internal(5.0)
    log_space = [val_to_log(v) for v in grey_space(21)]
    width = len(log_space)
    for x, v in enumerate(log_space):
        ret = o.oracle_kat_desaturate_v1(float(v), float(sigma), float(sigma), float(saturation))
        mirrored = o.oracle_kat_desaturate_v1(float(log_space[width - x - 1]), float(sigma), float(sigma), float(saturation))
        assert abs(ret - mirrored) <= E
        if x == 0 or x == width - 1:
            assert abs(ret - float(f32(1.0) - f32(1.0) / saturation)) <= E
        if 0.2 * width < x < (1.0 - 0.2) * width - 1:
            assert abs(ret - 1.0) <= 1e-2
    for v in log_space:
        assert abs(o.oracle_kat_desaturate_v1(float(v), float(sigma), float(sigma), float(_saturation_gui_to_internal(1e6))) - 1.0) <= 1e-2
    for v in grey_max_dr() + grey_max_dr_neg():
        ret = o.oracle_kat_desaturate_v1(float(v), float(sigma), float(sigma), float(saturation))
        assert 0.0 < ret <= 1.0


def test_linear_saturation(o):
    """test_filmicrgb.c:459-520"""
    ratios = (f32(0.2126), f32(0.7152), f32(0.0722))
    for v in grey_space():
        assert abs(o.oracle_kat_linear_saturation(float(v), float(v), 0.05) - float(v)) <= E
    for p in rgb_space():
        lum = p[0] * ratios[0] + p[1] * ratios[1] + p[2] * ratios[2]
        for c in range(3):
            assert abs(o.oracle_kat_linear_saturation(float(p[c]), float(lum), 1.0) - float(p[c])) <= E
            assert abs(o.oracle_kat_linear_saturation(float(p[c]), float(lum), 0.0) - float(lum)) <= E
