"""-m gpu: denoise (profiled), wavelets mode, against the CPU checkers.

HIP == oracle bit for bit (both sum the band statistics in the canonical binary64 order); against the
reference's own code (oracle/_ref) the agreement is bounded instead, because its band statistics are
an OpenMP float reduction whose value depends on the host's thread count."""
import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
from ansel_amd import abi, params, synth

pytestmark = pytest.mark.gpu

CASES = [
    (dict(), (640, 400)),
    (dict(color_mode=abi.DT_HIP_DENOISEPROFILE_RGB), (517, 333)),
    (dict(use_new_vst=False), (300, 200)),
    (dict(use_new_vst=False, fix=False), (300, 200)),
    (dict(color_mode=abi.DT_HIP_DENOISEPROFILE_RGB, wb_adaptive=False, strength=1.7, shadows=0.6, bias=-3.0), (1000, 700)),
    (dict(wb=(0.0, 0.0, 0.0, 0.0), strength=0.4), (257, 259)),
    (dict(force=[[0.5, 0.6, 0.7, 0.4, 0.3, 0.8, 0.2]] * 6), (1536, 1100)),   # 7 bands, uneven force curves
    # one, two and three bands: the synthesis forms the details of bands 1 .. from consecutive coarse planes (round 6), and with a
    # single band its residue is that band's coarse plane
    (dict(), (24, 20)),
    (dict(color_mode=abi.DT_HIP_DENOISEPROFILE_RGB), (40, 30)),
    (dict(use_new_vst=False), (64, 48)),
]


def _noisy(w, h, seed):
    rng = np.random.default_rng(seed)
    img = synth.rgba_image(w, h, seed=seed, lo=0.0, hi=0.9)
    img[..., :3] += rng.normal(0.0, 0.01, size=(h, w, 3)).astype(np.float32) * np.sqrt(np.maximum(img[..., :3], 0.01))
    return np.ascontiguousarray(img.astype(np.float32))


@pytest.mark.parametrize("case", range(len(CASES)))
def test_denoiseprofile_matches_oracle(case):
    over, (w, h) = CASES[case]
    img = _noisy(w, h, 40 + case)
    d = params.denoiseprofile(**over)
    piece = abi.Piece.make(w, h, processed_maximum=synth.WB_COEFFS)
    got = hc.run_hip("dt_hip_iop_denoiseprofile_process", piece, d, img, img.shape)
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_denoiseprofile", piece, d, img, want) == 0
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d values differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
    ref = ck.ref()
    if ref is not None:
        r = np.zeros_like(img)
        assert ck.call(ref, "ref_denoiseprofile", piece, d, img, r) == 0
        err = np.abs(got[..., :3] - r[..., :3]) / np.maximum(np.abs(r[..., :3]), 1e-3)
        assert float(err.max()) < 5e-5, float(err.max())


def test_denoiseprofile_small_frames():
    """below 2x the coarsest dilation the reference copies the input through (denoiseprofile.c:1325)"""
    w, h = 40, 24
    img = _noisy(w, h, 3)
    d = params.denoiseprofile()
    piece = abi.Piece.make(w, h, processed_maximum=synth.WB_COEFFS)
    got = hc.run_hip("dt_hip_iop_denoiseprofile_process", piece, d, img, img.shape)
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_denoiseprofile", piece, d, img, want) == 0
    assert np.array_equal(got, want)


@pytest.mark.parametrize("auto,manual", [(abi.DT_HIP_DENOISEPROFILE_WAVELETS_AUTO, abi.DT_HIP_DENOISEPROFILE_WAVELETS),
                                         (abi.DT_HIP_DENOISEPROFILE_NLMEANS_AUTO, abi.DT_HIP_DENOISEPROFILE_NLMEANS)])
def test_auto_modes_run_like_their_manual_mode(auto, manual):
    """MODE_*_AUTO: the auto sliders are resolved into the same fields at commit time and process() dispatches
    them like the manual modes (denoiseprofile.c:2617-2621)"""
    w, h = 300, 200
    img = _noisy(w, h, 9)
    piece = abi.Piece.make(w, h, processed_maximum=synth.WB_COEFFS)
    got = hc.run_hip("dt_hip_iop_denoiseprofile_process", piece, params.denoiseprofile(mode=auto), img, img.shape)
    same = hc.run_hip("dt_hip_iop_denoiseprofile_process", piece, params.denoiseprofile(mode=manual), img, img.shape)
    assert got.tobytes() == same.tobytes()
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_denoiseprofile", piece, params.denoiseprofile(mode=auto), img, want) == 0
    assert got.tobytes() == want.tobytes()


def test_variance_mode_is_refused():
    import ctypes as C
    from ansel_amd import lib
    l = hc.hip()
    buf = lib.DeviceBuffer(0, 64 * 64 * 16)
    piece = abi.Piece.make(64, 64)
    d = params.denoiseprofile(mode=2)
    assert l.dt_hip_iop_denoiseprofile_process(0, C.byref(piece), C.byref(d), buf.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG
