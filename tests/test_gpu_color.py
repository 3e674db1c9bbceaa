"""-m gpu: colorin / colorout (matrix path) and color calibration vs the CPU checkers."""
import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import abi, lib, params, synth

pytestmark = pytest.mark.gpu

W, H = 1200, 801


def _images():
    return {"scene": synth.rgba_image(W, H, seed=2, lo=-0.05, hi=1.6), "adversarial": synth.adversarial_rgba(W, H)}


def _luts():
    enc = params.srgb_encode_lut()
    dec = params.srgb_decode_lut()
    return enc, dec, params.unbounded_coeffs(enc), params.unbounded_coeffs(dec)


def _both(name, fn, piece, host_data, dev_data, img, max_ulp=0):
    got = hc.run_hip(fn, piece, dev_data, img, img.shape)
    for which in hc.checkers_available():
        exp = hc.run_cpu(which, name, piece, host_data, img, img.shape)
        hc.assert_bit_exact(got, exp, "%s vs %s" % (name, which))


@pytest.mark.parametrize("imgname", ["scene", "adversarial"])
@pytest.mark.parametrize("variant", ["matrix", "blue", "encode", "decode_clip_encode"])
def test_conversion(imgname, variant):
    img = _images()[imgname]
    piece = abi.Piece.make(W, H)
    enc, dec, ce, cd = _luts()
    d_enc = lib.DeviceBuffer.from_numpy(0, enc) if variant in ("encode", "decode_clip_encode") else None
    d_dec = lib.DeviceBuffer.from_numpy(0, dec) if variant == "decode_clip_encode" else None
    cam = params.WORK_OUT @ params.CAMERA_TO_XYZ
    out = params.SRGB_OUT @ params.WORK_IN

    def make(dev):
        lt = ls = None
        if variant in ("encode", "decode_clip_encode"):
            lt = [((d_enc.ptr if dev else enc.ctypes.data), float(enc[0]), ce)] * 3
        if variant == "decode_clip_encode":
            ls = [((d_dec.ptr if dev else dec.ctypes.data), float(dec[0]), cd)] * 3
        if variant == "matrix":
            return params.conversion(cam)
        if variant == "blue":
            return params.conversion(cam, blue_mapping=True)
        if variant == "encode":
            return params.conversion(out, lut_target=lt)
        return params.conversion(out, clip_matrix=np.eye(3), lut_source=ls, lut_target=lt)

    name = "colorin" if variant in ("matrix", "blue") else "colorout"
    _both(name, "dt_hip_iop_%s_process" % name, piece, make(False), make(True), img)


@pytest.mark.parametrize("imgname", ["scene", "adversarial"])
@pytest.mark.parametrize("adaptation", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("version", [0, 1, 2])
def test_channelmixerrgb(imgname, adaptation, version):
    img = _images()[imgname]
    piece = abi.Piece.make(W, H)
    d = params.channelmixerrgb(adaptation=adaptation, version=version, saturation=(0.1, -0.2, 0.05),
                               lightness=(0.05, 0.0, -0.1))
    _both("channelmixerrgb", "dt_hip_iop_channelmixerrgb_process", piece, d, d, img)


def test_channelmixerrgb_defaults_and_grey():
    img = _images()["scene"]
    piece = abi.Piece.make(W, H)
    for d in (params.channelmixerrgb(), params.channelmixerrgb(grey=(0.3, 0.5, 0.2), clip=False, gamut=2.0),
              params.channelmixerrgb(gamut=0.0, red=(1.1, -0.05, -0.05))):
        _both("channelmixerrgb", "dt_hip_iop_channelmixerrgb_process", piece, d, d, img)


@pytest.mark.parametrize("imgname", ["scene", "adversarial"])
@pytest.mark.parametrize("channels", [(0, 0, 0), (1, 1, 1), (1, 0, 1)])
def test_lab_glue_linear_and_with_the_tone_curves_of_the_work_profile(imgname, channels):
    """RGB <-> Lab around Lab modules (iop_profile.c:377-463); a work profile with tone curves runs _apply_tonecurves()
    (:332-372) ahead of the matrix on the way in and behind it on the way out"""
    img = _images()[imgname]
    piece = abi.Piece.make(W, H)
    enc, dec, ce, cd = _luts()
    d_enc, d_dec = lib.DeviceBuffer.from_numpy(0, enc), lib.DeviceBuffer.from_numpy(0, dec)

    def make(m, lut, dlut, co, dev):
        if not any(channels):
            return abi.LabData.make(m)
        return abi.LabData.make(m, [((dlut.ptr if dev else lut.ctypes.data) if on else None, float(lut[0]), co) for on in channels])
    _both("rgb_to_lab", "dt_hip_transform_rgb_to_lab", piece, make(params.WORK_IN, dec, d_dec, cd, False),
          make(params.WORK_IN, dec, d_dec, cd, True), img)
    lab = hc.run_cpu("oracle", "rgb_to_lab", piece, abi.LabData.make(params.WORK_IN), img, img.shape)
    lab = np.where(np.isfinite(lab), lab, 0).astype(np.float32)
    _both("lab_to_rgb", "dt_hip_transform_lab_to_rgb", piece, make(params.WORK_OUT, enc, d_enc, ce, False),
          make(params.WORK_OUT, enc, d_enc, ce, True), lab)
