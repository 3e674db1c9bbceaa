"""-m gpu: HIP kernels of the glue modules vs the CPU checkers, bit for bit, through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import abi, synth

pytestmark = pytest.mark.gpu

SIZES = [(1024, 768), (1501, 1003), (66, 34)]


def _cfa(w, h, seed=3):
    return synth.bayer_mosaic(w, h, seed=seed)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("crop", [(0, 0), (4, 2), (3, 1)])
def test_rawprepare_u16(w, h, crop):
    cx, cy = crop
    cfa = _cfa(w + cx + 8, h + cy + 6)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, datatype=abi.DT_HIP_TYPE_UINT16,
                           roi_in=abi.Roi.make(0, 0, cfa.shape[1], cfa.shape[0]), roi_out=abi.Roi.make(0, 0, w, h))
    d = abi.RawprepareData(cx, cy, 8 - cx, 6 - cy, abi.f4(512, 510, 514, 512),
                           abi.f4(*[synth.WHITE - 512, synth.WHITE - 510, synth.WHITE - 514, synth.WHITE - 512]))
    got = hc.run_hip("dt_hip_iop_rawprepare_process", piece, d, cfa, (h, w))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "rawprepare", piece, d, cfa, (h, w)), "rawprepare vs " + which)


@pytest.mark.parametrize("w,h", SIZES)
def test_rawprepare_float(w, h):
    cfa = _cfa(w, h).astype(np.float32)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, datatype=abi.DT_HIP_TYPE_FLOAT)
    d = abi.RawprepareData(0, 0, 0, 0, abi.f4(512, 512, 512, 512), abi.f4(*[15871.0] * 4))
    got = hc.run_hip("dt_hip_iop_rawprepare_process", piece, d, cfa, (h, w))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "rawprepare", piece, d, cfa, (h, w)), "rawprepare f32 vs " + which)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("roi_xy", [(0, 0), (1, 1), (2, 3)])
def test_temperature_bayer(w, h, roi_xy):
    rng = np.random.default_rng(5)
    img = rng.random((h, w), dtype=np.float32) * 1.2 - 0.01
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, roi_in=abi.Roi.make(*roi_xy, w, h),
                           roi_out=abi.Roi.make(*roi_xy, w, h))
    d = abi.TemperatureData(abi.f4(*synth.WB_COEFFS))
    got = hc.run_hip("dt_hip_iop_temperature_process", piece, d, img, (h, w))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "temperature", piece, d, img, (h, w)), "temperature vs " + which)


def test_temperature_rgba():
    w, h = 640, 480
    img = synth.adversarial_rgba(w, h)
    img[..., 3] = np.random.default_rng(1).random((h, w), dtype=np.float32)
    piece = abi.Piece.make(w, h, filters=0, channels=4)
    d = abi.TemperatureData(abi.f4(1.9, 1.0, 1.4, 1.0))
    got = hc.run_hip("dt_hip_iop_temperature_process", piece, d, img, (h, w, 4))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "temperature", piece, d, img, (h, w, 4)), "temperature rgba vs " + which)


@pytest.mark.parametrize("w,h", SIZES + [(1023, 511)])
@pytest.mark.parametrize("nclipped", [0, 7, 24, 25, 26, 5000])
def test_highlights_clip_bayer(w, h, nclipped):
    """the <25 clipped photosites bypass (highlights.c:266-300) must be reproduced exactly"""
    rng = np.random.default_rng(11)
    img = rng.random((h, w), dtype=np.float32) * 0.98
    flat = img.reshape(-1)
    idx = rng.choice(flat.size, size=min(nclipped, flat.size), replace=False)
    flat[idx] = 1.0 + rng.random(len(idx), dtype=np.float32)
    pm = (2.29, 1.0, 1.65, 1.0)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=pm)
    d = abi.HighlightsData(abi.DT_HIP_HIGHLIGHTS_CLIP, 1.0)
    got = hc.run_hip("dt_hip_iop_highlights_process", piece, d, img, (h, w))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "highlights", piece, d, img, (h, w)), "highlights vs " + which)
    if nclipped < 25:
        assert np.array_equal(got, img)
    else:
        assert got.max() <= 1.0


@pytest.mark.parametrize("nclipped", [3, 40])
def test_highlights_clip_rgba(nclipped):
    w, h = 321, 123
    rng = np.random.default_rng(2)
    img = rng.random((h, w, 4), dtype=np.float32) * 0.9
    ys = rng.integers(0, h, nclipped)
    xs = rng.integers(0, w, nclipped)
    img[ys, xs, rng.integers(0, 3, nclipped)] = 1.5
    piece = abi.Piece.make(w, h, filters=0, channels=4, processed_maximum=(0, 0, 0, 0))
    d = abi.HighlightsData(abi.DT_HIP_HIGHLIGHTS_CLIP, 1.0)
    got = hc.run_hip("dt_hip_iop_highlights_process", piece, d, img, (h, w, 4))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "highlights", piece, d, img, (h, w, 4)), "highlights rgba vs " + which)


@pytest.mark.parametrize("w,h", SIZES)
def test_exposure(w, h):
    img = synth.adversarial_rgba(w, h)
    piece = abi.Piece.make(w, h, channels=4)
    d = abi.ExposureData(-0.000244140625, 1.4142135)
    got = hc.run_hip("dt_hip_iop_exposure_process", piece, d, img, (h, w, 4))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "exposure", piece, d, img, (h, w, 4)), "exposure vs " + which)


def test_exposure_raw_channel():
    w, h = 333, 77
    img = np.random.default_rng(9).random((h, w), dtype=np.float32)
    piece = abi.Piece.make(w, h, channels=1, filters=synth.FILTERS_RGGB)
    d = abi.ExposureData(0.01, 2.0)
    got = hc.run_hip("dt_hip_iop_exposure_process", piece, d, img, (h, w))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "exposure", piece, d, img, (h, w)), "exposure 1ch vs " + which)


@pytest.mark.parametrize("kind", ["u16", "u8"])
def test_export_convert(kind):
    w, h = 1000, 700
    img = synth.adversarial_rgba(w, h)
    img[0, :16, 0] = [np.nan, np.inf, -np.inf, 0.5 / 65535, 1.5 / 65535, 2.5 / 65535, 1.0, 1.0000001, -0.0, 0.9999999,
                      0.5 / 255, 1.5 / 255, 254.5 / 255, 2.0, -1.0, 0.5]
    h_ = hc.hip()
    from ansel_amd import lib
    din = lib.DeviceBuffer.from_numpy(0, img)
    dt = np.uint16 if kind == "u16" else np.uint8
    dout = lib.DeviceBuffer(0, w * h * 4 * np.dtype(dt).itemsize)
    fn = h_.dt_hip_export_convert_u16 if kind == "u16" else h_.dt_hip_export_convert_u8
    lib.check(fn(0, w, h, din.ptr, dout.ptr), "export_convert")
    got = dout.to_numpy((h, w, 4), dt)
    import checkers as ck
    for which in hc.checkers_available():
        l = ck.ref() if which == "ref" else ck.oracle()
        exp = np.zeros((h, w, 4), dt)
        f = getattr(l, ("ref_" if which == "ref" else "oracle_") + "export_convert_" + kind)
        f(w, h, ck.ptr(img), ck.ptr(exp))
        assert np.array_equal(got, exp), "export %s vs %s: %d differ" % (kind, which, int((got != exp).sum()))
