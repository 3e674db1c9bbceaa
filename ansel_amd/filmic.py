"""filmic RGB: user parameters (dt_iop_filmicrgb_params_t, src/iop/filmicrgb.c:255-286) and the
commit step that turns them into the per-piece data the kernel reads (commit_params(),
filmicrgb.c:4005-4110, with the spline solver dt_iop_filmic_rgb_compute_spline(), :3686-4003).

commit() below is the product's host-side implementation (binary32 arithmetic in the reference's
order, via numpy scalars and the platform libm).  tests/test_oracle_vs_ref.py
(test_filmic_commit_matches_reference_solver) checks it field by field against the reference's own solver (oracle/_ref)."""
import ctypes as C
import math

import numpy as np

from . import abi, params

f32 = np.float32

# enums
METHOD_NONE, METHOD_MAX_RGB, METHOD_LUMINANCE, METHOD_POWER_NORM, METHOD_EUCLIDEAN_V1, METHOD_EUCLIDEAN_V2 = 0, 1, 2, 3, 4, 5
CURVE_POLY_4, CURVE_POLY_3, CURVE_RATIONAL, CURVE_SIGMOID = 0, 1, 2, 3
V4_2020, V6_2022, V7_2023, AGX_NONE, AGX_LOW, AGX_MEDIUM, AGX_HIGH, AGX_EXTRA = 1, 3, 4, 5, 6, 7, 8, 9
SPLINE_V1, SPLINE_V2, SPLINE_V3 = 0, 1, 2
SAFETY_MARGIN = f32(0.01)


class UserParams(C.Structure):
    """mirror of ref_filmic_params_t (oracle/ref_wrap/ref_filmic.c) == the subset of
    dt_iop_filmicrgb_params_t the tone mapping depends on; defaults are the $DEFAULTs"""
    _fields_ = [("grey_point_source", C.c_float), ("black_point_source", C.c_float), ("white_point_source", C.c_float),
                ("security_factor", C.c_float), ("grey_point_target", C.c_float), ("black_point_target", C.c_float),
                ("white_point_target", C.c_float), ("output_power", C.c_float), ("latitude", C.c_float),
                ("contrast", C.c_float), ("saturation", C.c_float), ("balance", C.c_float),
                ("preserve_color", C.c_int), ("version", C.c_int), ("auto_hardness", C.c_int), ("custom_grey", C.c_int),
                ("shadows", C.c_int), ("highlights", C.c_int), ("spline_version", C.c_int)]

    @classmethod
    def defaults(cls, **kw):
        p = cls(18.45, -8.0, 4.0, 0.0, 18.45, 0.01517634, 100.0, 4.0, 10.0, 1.18, 0.0, 0.0,
                METHOD_MAX_RGB, AGX_MEDIUM, 1, 0, CURVE_SIGMOID, CURVE_SIGMOID, SPLINE_V3)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


_libm = C.CDLL("libm.so.6")
_libm.powf.restype = C.c_float
_libm.powf.argtypes = [C.c_float, C.c_float]


def _powf(a, b):
    # the C library's powf, as commit_params() calls it (numpy's float32 power is a different routine)
    return f32(_libm.powf(float(f32(a)), float(f32(b))))


def _clamp(x, lo, hi):  # glib CLAMP
    x, lo, hi = f32(x), f32(lo), f32(hi)
    return hi if x > hi else (lo if x < lo else x)


def _gauss_solve(A, b):
    """src/math/gaussian_elimination.h: double-precision elimination with partial pivoting"""
    return np.linalg.solve(np.asarray(A, dtype=np.float64), np.asarray(b, dtype=np.float64))


def _sigmoid_scale(limit_x, limit_y, transition_x, transition_y, slope, power):
    projected_rise = f32(slope) * max(f32(1e-6), f32(limit_x) - f32(transition_x))
    actual_rise = max(f32(1e-6), f32(limit_y) - f32(transition_y))
    base = max(f32(1e-6), _powf(actual_rise, -f32(power)) - _powf(projected_rise, -f32(power)))
    return min(f32(1e9), _powf(base, f32(-1.0) / f32(power)))


def compute_spline(p):
    """dt_iop_filmic_rgb_compute_spline(), spline_version >= v3 node geometry
    (filmic_v3_compute_geometry / _compute_nodes_from_legacy, filmicrgb.c:497-566)"""
    if p.spline_version < SPLINE_V3:
        raise NotImplementedError("legacy spline geometries (v1, v2) are not implemented host-side")
    one = f32(1.0)
    out_pow = f32(p.output_power)
    if p.custom_grey:
        grey_display = _powf(_clamp(p.grey_point_target, p.black_point_target, p.white_point_target) / f32(100.0), one / out_pow)
    else:
        grey_display = _powf(f32(0.1845), one / out_pow)
    dynamic_range = f32(p.white_point_source) - f32(p.black_point_source)
    grey_log = f32(abs(f32(p.black_point_source))) / dynamic_range
    black_display = _powf(_clamp(p.black_point_target, 0.0, p.grey_point_target) / f32(100.0), one / out_pow)
    white_display = _powf(max(f32(p.white_point_target), f32(p.grey_point_target)) / f32(100.0), one / out_pow)
    slope = f32(p.contrast) * dynamic_range / f32(8.0)
    min_contrast = f32(1.0)
    min_contrast = max(min_contrast, (white_display - grey_display) / (one - grey_log))
    min_contrast = max(min_contrast, (grey_display - black_display) / grey_log)
    min_contrast = f32(min_contrast + SAFETY_MARGIN)
    contrast = slope / (out_pow * _powf(grey_display, out_pow - one))
    contrast = _clamp(contrast, min_contrast, 100.0)
    linear_intercept = grey_display - contrast * grey_log
    safety_margin = SAFETY_MARGIN * (white_display - black_display)
    xmin = (black_display + safety_margin - linear_intercept) / contrast
    xmax = (white_display - safety_margin - linear_intercept) / contrast
    latitude = _clamp(p.latitude, 0.0, 100.0) / f32(100.0)
    balance = _clamp(p.balance, -50.0, 50.0) / f32(100.0)
    toe_log = (one - latitude) * grey_log + latitude * xmin
    shoulder_log = (one - latitude) * grey_log + latitude * xmax
    if balance > 0:
        corr = f32(2.0) * balance * (shoulder_log - grey_log)
    else:
        corr = f32(2.0) * balance * (grey_log - toe_log)
    toe_log = f32(toe_log - corr)
    shoulder_log = f32(shoulder_log - corr)
    toe_log = max(toe_log, xmin)
    shoulder_log = min(shoulder_log, xmax)
    toe_display = toe_log * contrast + linear_intercept
    shoulder_display = shoulder_log * contrast + linear_intercept

    s = abi.FilmicSpline()
    x = [f32(0.0), toe_log, grey_log, shoulder_log, f32(1.0)]
    y = [black_display, toe_display, grey_display, shoulder_display, white_display]
    for i in range(5):
        s.x[i] = x[i]
        s.y[i] = y[i]
    s.latitude_min = x[1]
    s.latitude_max = x[3]
    s.type[0] = p.shadows
    s.type[1] = p.highlights
    M = [[f32(0)] * 4 for _ in range(5)]  # M1..M5
    M[1][2] = contrast
    M[0][2] = f32(y[1] - M[1][2] * x[1])
    sig_toe_power = f32(1.5)
    sig_slope = M[1][2]
    if p.shadows == CURVE_SIGMOID or p.highlights == CURVE_SIGMOID:
        M[2][2] = y[0]
        M[3][2] = y[4]
    Tl = float(x[1])
    Sl = float(x[3])
    # toe
    if p.shadows == CURVE_SIGMOID:
        tx, ty, y0 = x[1], y[1], y[0]
        dx = max(f32(1e-6), tx)
        dy = max(f32(1e-6), f32(ty - y0))
        M[0][0] = f32(-_sigmoid_scale(one, one - y0, one - tx, one - ty, sig_slope, sig_toe_power))
        M[1][0] = sig_toe_power
        M[3][0] = sig_slope * dx / dy
        M[2][0] = dy / _powf(dx, M[3][0])
        M[4][0] = f32(1.0) if (dy / dx > sig_slope) else f32(0.0)
    elif p.shadows == CURVE_POLY_4:
        A = [[0, 0, 0, 0, 1], [0, 0, 0, 1, 0], [Tl**4, Tl**3, Tl**2, Tl, 1], [4 * Tl**3, 3 * Tl**2, 2 * Tl, 1, 0],
             [12 * Tl**2, 6 * Tl, 2, 0, 0]]
        b = _gauss_solve(A, [float(y[0]), 0.0, float(y[1]), float(M[1][2]), 0.0])
        M[4][0], M[3][0], M[2][0], M[1][0], M[0][0] = (f32(v) for v in b)
    elif p.shadows == CURVE_POLY_3:
        A = [[0, 0, 0, 1], [Tl**3, Tl**2, Tl, 1], [3 * Tl**2, 2 * Tl, 1, 0], [6 * Tl, 2, 0, 0]]
        b = _gauss_solve(A, [float(y[0]), float(y[1]), float(M[1][2]), 0.0])
        M[4][0] = f32(0)
        M[3][0], M[2][0], M[1][0], M[0][0] = (f32(v) for v in b)
    else:
        xx = f32(x[1] - x[0])
        yy = f32(y[1] - y[0])
        g = contrast
        sq = lambda v: f32(v) * f32(v)
        b = g / (f32(2) * yy) + (f32(np.sqrt(sq(xx * g / yy + one) - f32(4))) - one) / (f32(2) * xx)
        c = yy / g * (b * sq(xx) + xx) / (b * sq(xx) + xx - (yy / g))
        a = c * g
        M[0][0], M[1][0], M[2][0], M[3][0] = f32(a), f32(b), f32(c), toe_display
    # shoulder
    if p.highlights == CURVE_SIGMOID:
        sx, sy, y4 = x[3], y[3], y[4]
        dx = max(f32(1e-6), f32(one - sx))
        dy = max(f32(1e-6), f32(y4 - sy))
        M[3][1] = sig_slope * dx / dy
        M[2][1] = dy / _powf(dx, M[3][1])
        M[4][1] = f32(1.0)
    elif p.highlights == CURVE_POLY_3:
        A = [[1, 1, 1, 1], [Sl**3, Sl**2, Sl, 1], [3 * Sl**2, 2 * Sl, 1, 0], [6 * Sl, 2, 0, 0]]
        b = _gauss_solve(A, [float(y[4]), float(y[3]), float(M[1][2]), 0.0])
        M[4][1] = f32(0)
        M[3][1], M[2][1], M[1][1], M[0][1] = (f32(v) for v in b)
    elif p.highlights == CURVE_POLY_4:
        A = [[1, 1, 1, 1, 1], [4, 3, 2, 1, 0], [Sl**4, Sl**3, Sl**2, Sl, 1], [4 * Sl**3, 3 * Sl**2, 2 * Sl, 1, 0],
             [12 * Sl**2, 6 * Sl, 2, 0, 0]]
        b = _gauss_solve(A, [float(y[4]), 0.0, float(y[3]), float(M[1][2]), 0.0])
        M[4][1], M[3][1], M[2][1], M[1][1], M[0][1] = (f32(v) for v in b)
    else:
        xx = f32(x[4] - x[3])
        yy = f32(y[4] - y[3])
        g = contrast
        sq = lambda v: f32(v) * f32(v)
        b = g / (f32(2) * yy) + (f32(np.sqrt(sq(xx * g / yy + one) - f32(4))) - one) / (f32(2) * xx)
        c = yy / g * (b * sq(xx) + xx) / (b * sq(xx) + xx - (yy / g))
        a = c * g
        M[0][1], M[1][1], M[2][1], M[3][1] = f32(a), f32(b), f32(c), shoulder_display
    for name, row in zip(("M1", "M2", "M3", "M4", "M5"), M):
        for k in range(4):
            getattr(s, name)[k] = row[k]
    return s


def commit(p, work_in=params.WORK_IN, work_out=params.WORK_OUT, export_in=params.SRGB_IN, export_out=params.SRGB_OUT,
           use_output_profile=True):
    """commit_params(): user parameters -> dt_hip_filmicrgb_data_t"""
    d = abi.FilmicrgbData()
    d.white_source = p.white_point_source
    d.black_source = p.black_point_source
    d.grey_source = f32(p.grey_point_source) / f32(100.0) if p.custom_grey else f32(0.1845)
    d.dynamic_range = f32(p.white_point_source) - f32(p.black_point_source)
    d.output_power = p.output_power
    d.version = p.version
    d.preserve_color = p.preserve_color
    d.spline = compute_spline(p)
    if p.version >= V6_2022:
        d.saturation = f32(p.saturation) / f32(100.0)
    else:
        d.saturation = f32(2.0) * f32(p.saturation) / f32(100.0) + f32(1.0)
    axis = f32(p.saturation) / f32(100.0)
    axis = axis if axis >= -1 else f32(-1)
    axis = axis if axis <= 1 else f32(1)
    d.agx_beta_hue = f32(0.5) * (axis + f32(1.0))
    set_profiles(d, work_in, work_out, export_in, export_out, use_output_profile)
    return d


def set_profiles(d, work_in=params.WORK_IN, work_out=params.WORK_OUT, export_in=params.SRGB_IN,
                 export_out=params.SRGB_OUT, use_output_profile=True):
    abi.set_m34(d.work_matrix_in, np.asarray(work_in, dtype=np.float32))
    abi.set_m34(d.work_matrix_out, np.asarray(work_out, dtype=np.float32))
    abi.set_m34(d.export_matrix_in, np.asarray(export_in, dtype=np.float32))
    abi.set_m34(d.export_matrix_out, np.asarray(export_out, dtype=np.float32))
    d.use_output_profile = 1 if use_output_profile else 0
    return d


def default_data():
    return commit(UserParams.defaults())
