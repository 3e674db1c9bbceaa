"""Host-side parameter preparation for the hot modules: the arithmetic of each module's
commit_params() that turns user parameters into the per-piece `data` the kernels read.

These are the inputs of the device path, not part of it.  Colour-profile matrices and tone
curves come from lcms2 in the reference (src/colorprofiles/conversion.c:216-487,
colorspaces.c); here they are derived analytically from the published primaries / transfer
functions, which is all a matrix profile is.
"""
import math

import numpy as np

from . import abi, synth

# ---- colour science constants -----------------------------------------------------------
D50_XYZ = np.array([0.9642119944211994, 1.0, 0.8251882845188288])
D65_xy = (0.3127, 0.3290)

REC2020 = ((0.708, 0.292), (0.170, 0.797), (0.131, 0.046))
REC709 = ((0.64, 0.33), (0.30, 0.60), (0.15, 0.06))

BRADFORD = np.array([[0.8951, 0.2664, -0.1614], [-0.7502, 1.7135, 0.0367], [0.0389, -0.0685, 1.0296]])


def _xy_to_XYZ(x, y):
    return np.array([x / y, 1.0, (1.0 - x - y) / y])


def rgb_to_xyz_d50(primaries, white_xy=D65_xy):
    """RGB -> XYZ(D50) matrix of an RGB space given by its primaries and white point
    (Bradford-adapted to the D50 PCS, as every ICC matrix profile is)."""
    P = np.stack([_xy_to_XYZ(*p) for p in primaries], axis=1)
    W = _xy_to_XYZ(*white_xy)
    S = np.linalg.solve(P, W)
    M = P * S[None, :]
    src = BRADFORD @ W
    dst = BRADFORD @ D50_XYZ
    A = np.linalg.inv(BRADFORD) @ np.diag(dst / src) @ BRADFORD
    return A @ M


WORK_IN = rgb_to_xyz_d50(REC2020)            # linear Rec2020 (the default work profile) -> XYZ D50
WORK_OUT = np.linalg.inv(WORK_IN)
SRGB_IN = rgb_to_xyz_d50(REC709)
SRGB_OUT = np.linalg.inv(SRGB_IN)

# a camera RGB -> XYZ(D50) matrix of a typical CMOS sensor after white balance
CAMERA_TO_XYZ = np.array([[0.6137, 0.2598, 0.0907],
                          [0.2359, 0.8604, -0.0963],
                          [0.0318, -0.1949, 0.9882]])


def srgb_encode_lut(n=abi.DT_HIP_LUT_SAMPLES):
    """sRGB OETF sampled like dt_colorspaces_prepare_conversion() samples a curve: n points on [0,1]"""
    x = np.linspace(0.0, 1.0, n, dtype=np.float64)
    y = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)
    return y.astype(np.float32)


def srgb_decode_lut(n=abi.DT_HIP_LUT_SAMPLES):
    x = np.linspace(0.0, 1.0, n, dtype=np.float64)
    y = np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4))
    return y.astype(np.float32)


def unbounded_coeffs(lut):
    """dt_ioppr_init_unbounded_coeffs-style fit y = b*(a*x)^c through the top of the curve
    (x in [0.9, 1.0]); the exact fit is a parameter, any {a,b,c} is legal input."""
    n = len(lut)
    x0, x1 = 0.9, 1.0
    y0 = float(lut[int(round(x0 * (n - 1)))])
    y1 = float(lut[n - 1])
    c = math.log(y1 / y0) / math.log(x1 / x0)
    return (1.0, y1, c)


def conversion(matrix, clip_matrix=None, lut_source=None, lut_target=None, blue_mapping=False):
    """Fill a dt_hip_conversion_t.  lut_* are 3-sequences of device or host pointers (ints) plus
    the matching first samples, given as [(ptr, first_value, (a, b, c)), ...] or None."""
    d = abi.Conversion()
    abi.set_m34(d.matrix, np.asarray(matrix, dtype=np.float32))
    if clip_matrix is not None:
        abi.set_m34(d.clip_matrix, np.asarray(clip_matrix, dtype=np.float32))
        d.has_clipping = 1
    d.blue_mapping = 1 if blue_mapping else 0
    for name, luts in (("source", lut_source), ("target", lut_target)):
        if luts is None:
            continue
        n = 0
        for c, (ptr, first, coeff) in enumerate(luts):
            getattr(d, "lut_" + name)[c] = ptr
            getattr(d, "lut_%s_first" % name)[c] = first
            for j in range(3):
                getattr(d, "coeffs_" + name)[c][j] = coeff[j]
            n += 1 if first >= 0 else 0
        setattr(d, "nonlinear_" + name, n)
    return d


# ---- color calibration (src/iop/channelmixerrgb.c:2974-3056) -------------------------------
XYZ_TO_CAT16 = np.array([[0.401288, 0.650173, -0.051461], [-0.250268, 1.204414, 0.045854],
                         [-0.002079, 0.048952, 0.953127]], dtype=np.float32)
XYZ_TO_BRADFORD = np.array([[0.8951, 0.2664, -0.1614], [-0.7502, 1.7135, 0.0367],
                            [0.0389, -0.0685, 1.0296]], dtype=np.float32)


def channelmixerrgb(adaptation=abi.DT_HIP_ADAPTATION_CAT16, illuminant_xy=(0.34567, 0.35850),
                    red=(1, 0, 0), green=(0, 1, 0), blue=(0, 0, 1), saturation=(0, 0, 0), lightness=(0, 0, 0),
                    grey=(0, 0, 0), gamut=1.0, clip=True, version=2, normalize_grey=True,
                    rgb_to_xyz=WORK_IN, xyz_to_rgb=WORK_OUT):
    d = abi.ChannelmixerrgbData()
    abi.set_m34(d.RGB_to_XYZ, np.asarray(rgb_to_xyz, dtype=np.float32))
    abi.set_m34(d.XYZ_to_RGB, np.asarray(xyz_to_rgb, dtype=np.float32))
    abi.set_m34(d.MIX, np.asarray([red, green, blue], dtype=np.float32))
    f = np.float32
    for i in range(3):
        d.saturation[i] = -f(saturation[2 - i] if version == 0 and i != 1 else saturation[i])
        d.lightness[i] = f(lightness[i])
    norm_grey = f(grey[0]) + f(grey[1]) + f(grey[2])
    d.apply_grey = 1 if any(g != 0 for g in grey) else 0
    if not normalize_grey or norm_grey == 0:
        norm_grey = f(1)
    for i in range(3):
        d.grey[i] = f(grey[i]) / norm_grey
    d.adaptation = adaptation
    d.clip = 1 if clip else 0
    d.gamut = gamut if gamut == 0 else f(1) / f(gamut)
    x, y = f(illuminant_xy[0]), f(illuminant_xy[1])
    XYZ = np.array([x / y, 1.0, (f(1) - x - y) / y], dtype=np.float32)  # illuminant_xy_to_XYZ()
    if adaptation in (abi.DT_HIP_ADAPTATION_FULL_BRADFORD, abi.DT_HIP_ADAPTATION_LINEAR_BRADFORD):
        lms = XYZ_TO_BRADFORD @ XYZ
    elif adaptation == abi.DT_HIP_ADAPTATION_CAT16:
        lms = XYZ_TO_CAT16 @ XYZ
    else:
        lms = XYZ
    for i in range(3):
        d.illuminant[i] = f(lms[i])
    d.illuminant[3] = 0.0
    d.p = float(np.power(np.float32(0.818155) / np.float32(d.illuminant[2]), np.float32(0.0834)))
    d.version = version
    return d


# ---- diffuse or sharpen (src/iop/diffuse.c) ---------------------------------------------------
def diffuse(preset="default", iscale=1.0, **over):
    """dt_hip_diffuse_data_t from the module defaults ($DEFAULT annotations, diffuse.c:79-101) or one
    of the presets of init_presets() (diffuse.c:298-583)"""
    base = dict(iterations=1, sharpness=0.0, radius=8, regularization=0.0, variance_threshold=0.0,
                anisotropy_first=0.0, anisotropy_second=0.0, anisotropy_third=0.0, anisotropy_fourth=0.0,
                threshold=0.0, first=0.0, second=0.0, third=0.0, fourth=0.0, radius_center=0)
    presets = {
        "default": {},
        # "lens deblur: soft", diffuse.c:304-325
        "lens_deblur_soft": dict(regularization=1.0, anisotropy_first=2.0, anisotropy_third=2.0, first=-0.25,
                                 second=0.125, third=-0.125, fourth=0.0625, radius=8, iterations=8),
        # "inpaint highlights", diffuse.c:520-538: everything above the threshold is seeded with noise and diffused
        "inpaint_highlights": dict(iterations=32, radius=4, threshold=1.41, anisotropy_fourth=2.0, fourth=0.5),
        # "fast local contrast", diffuse.c:561-583
        "fast_local_contrast": dict(radius_center=512, radius=512, anisotropy_third=5.0, third=-0.5, iterations=1),
    }
    base.update(presets[preset])
    base.update(over)
    d = abi.DiffuseData()
    for k, v in base.items():
        setattr(d, k, v)
    d.iscale = iscale
    return d


# ---- denoise (profiled), wavelets (src/iop/denoiseprofile.c) ------------------------------------
def denoiseprofile(color_mode=abi.DT_HIP_DENOISEPROFILE_Y0U0V0, use_new_vst=True, fix=True, wb_adaptive=True,
                   strength=1.0, shadows=1.0, bias=0.0, a=2.0e-5 * 4, b=-2.0e-7, force=None, wb=None,
                   mode=abi.DT_HIP_DENOISEPROFILE_WAVELETS, radius=1.0, nbhood=7.0, scattering=0.0,
                   central_pixel_weight=0.1):
    """module defaults ($DEFAULT annotations, denoiseprofile.c:270-305) with a Sony-like ISO 400 noise
    profile {a, b} and flat 0.5 force curves"""
    d = abi.DenoiseprofileData()
    d.radius, d.nbhood, d.strength, d.shadows, d.bias = radius, nbhood, strength, shadows, bias
    d.scattering, d.central_pixel_weight, d.overshooting = scattering, central_pixel_weight, 1.0
    for k in range(3):
        d.a[k] = a
        d.b[k] = b
    d.mode = mode
    for c in range(6):
        for band in range(7):
            d.force[c][band] = 0.5 if force is None else float(force[c][band])
    d.wb_adaptive_anscombe = 1 if wb_adaptive else 0
    d.fix_anscombe_and_nlmeans_norm = 1 if fix else 0
    d.use_new_vst = 1 if use_new_vst else 0
    d.wavelet_color_mode = color_mode
    wbc = synth.WB_COEFFS if wb is None else wb
    for k in range(4):
        d.wb_coeffs[k] = wbc[k] if k < len(wbc) else 0.0
    return d
