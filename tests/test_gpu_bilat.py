"""-m gpu: local contrast, bilateral-grid mode, bit for bit against the oracle (= the reference on
one thread: its splat sums per OpenMP slice, see oracle/src/bilat.c) and within rounding of the
reference at its default thread count."""
import math

import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
from ansel_amd import abi, synth

pytestmark = pytest.mark.gpu


def _lab_image(w, h, seed):
    rng = np.random.default_rng(seed)
    rgb = synth.rgba_image(w, h, seed=seed, lo=0.0, hi=1.0)
    lab = np.zeros((h, w, 4), np.float32)
    lab[..., 0] = 100.0 * rgb[..., 1] + rng.normal(0, 1.5, (h, w))
    lab[..., 1] = 80.0 * (rgb[..., 0] - rgb[..., 1]) + rng.normal(0, 2.0, (h, w))
    lab[..., 2] = 80.0 * (rgb[..., 1] - rgb[..., 2]) + rng.normal(0, 2.0, (h, w))
    lab[::7, ::5, 0] = -3.0
    lab[3::11, 2::9, 0] = 140.0
    lab[..., 3] = 0.25
    return np.ascontiguousarray(lab.astype(np.float32))


@pytest.mark.parametrize("w,h", [(300, 200), (123, 457), (1500, 1000)])
# (150, ...) on the large frame: a footprint wider than the LDS table of column weights holds (the gather computes them inline)
@pytest.mark.parametrize("ss,sr,detail", [(50.0, 25.0, 0.33), (8.0, 5.0, -0.5), (0.3, 2.0, 1.5), (20.0, 60.0, 4.0),
                                          (150.0, 10.0, 0.7)])
def test_bilat(w, h, ss, sr, detail):
    if ss < 1.0 and w * h > 500000:
        pytest.skip("sub-pixel sigma on the large frame: the grid has millions of nodes, covered on the small frames")
    if ss > 100.0 and min(w, h) < 200:
        pytest.skip("a grid line shorter than 4 nodes: refused, see test_bilat_refuses_a_grid_the_reference_overruns")
    img = _lab_image(w, h, 29)
    d = abi.BilatData.bilateral(ss, sr, detail)
    piece = abi.Piece.make(w, h)
    got = hc.run_hip("dt_hip_iop_bilat_process", piece, d, img, img.shape)
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_bilat", piece, d, img, want) == 0
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d values differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
    assert np.array_equal(got[..., 1:], img[..., 1:])  # a, b, alpha pass through
    ref = ck.ref()
    if ref is not None:
        r = np.zeros_like(img)
        assert ck.call(ref, "ref_bilat", piece, d, img, r) == 0
        assert float(np.abs(r[..., 0] - got[..., 0]).max()) < 1e-3


@pytest.mark.parametrize("sr,cells", [(60.0, 5), (25.0, 5), (20.0, 6), (16.0, 7), (14.0, 8), (12.5, 9), (11.0, 10)])
@pytest.mark.parametrize("w,h,ss", [(700, 500, 12.0), (1500, 1000, 50.0)])
def test_bilat_lightness_axes(w, h, ss, sr, cells):
    """grids of 5 .. 10 lightness cells (dt_bilateral_grid_size() allows no fewer than 4 intervals), the image with lightness below
    0 and above 100 -- the clamped ends of the axis: the same words as the oracle"""
    img = _lab_image(w, h, 37)
    d = abi.BilatData.bilateral(ss, sr, 0.45)
    piece = abi.Piece.make(w, h)
    assert int(math.ceil(100.0 / (100.0 / max(4, round(100.0 / sr))))) + 1 in (cells, cells + 1)
    got = hc.run_hip("dt_hip_iop_bilat_process", piece, d, img, img.shape)
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_bilat", piece, d, img, want) == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("w,h,ss,sr", [(300, 200, 8.0, 5.0), (123, 457, 0.3, 2.0), (1500, 1000, 50.0, 25.0), (640, 480, 160.0, 25.0)])
def test_bilat_blur_with_the_x_pass_in_its_own_launch(w, h, ss, sr):
    """grids beyond 2^22 cells take the blur's x-pass in a launch of its own (bilat_blur_x + bilat_blur_yz<false>); the test hook
    sends an ordinary grid that way: same words as the fused launch and as the oracle; the last case is a 4-node grid line"""
    from ansel_amd import lib
    img = _lab_image(w, h, 31)
    d = abi.BilatData.bilateral(ss, sr, 0.6)
    piece = abi.Piece.make(w, h)
    fused = hc.run_hip("dt_hip_iop_bilat_process", piece, d, img, img.shape)
    lib.test_dispatch("bilat_blur_split", 1)
    try:
        split = hc.run_hip("dt_hip_iop_bilat_process", piece, d, img, img.shape)
    finally:
        lib.test_dispatch("bilat_blur_split", 0)
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_bilat", piece, d, img, want) == 0
    assert np.array_equal(split.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(fused.view(np.uint32), want.view(np.uint32))


def test_bilat_refuses_a_grid_the_reference_overruns():
    """blur_line() touches four nodes of every grid line unconditionally (bilateral.c:266-340): on a 3-node line the
    reference writes past its buffer, so there is nothing to be identical to"""
    from ansel_amd import lib
    w, h = 123, 457
    img = _lab_image(w, h, 3)
    with pytest.raises(lib.AnselHipError, match="below the 4 entries"):
        hc.run_hip("dt_hip_iop_bilat_process", abi.Piece.make(w, h), abi.BilatData.bilateral(150.0, 10.0, 0.7), img, img.shape)


@pytest.mark.parametrize("w,h", [(300, 200), (123, 457), (64, 64), (257, 130), (1500, 1000)])
@pytest.mark.parametrize("hl,sh,detail,mid", [(0.5, 0.5, 0.25, 0.5), (1.0, 0.2, 1.5, 0.3), (0.1, 1.3, -0.6, 0.8)])
def test_local_laplacian(w, h, hl, sh, detail, mid):
    """the module's default mode"""
    img = _lab_image(w, h, 33)
    d = abi.BilatData.local_laplacian(hl, sh, detail, mid)
    piece = abi.Piece.make(w, h)
    pre = np.full(img.shape, -5.0, np.float32)
    got = hc.run_hip("dt_hip_iop_bilat_process", piece, d, img, img.shape, pre_fill=pre)
    want = pre.copy()
    assert ck.call(ck.oracle(), "oracle_bilat", piece, d, img, want) == 0
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d values differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
    ref = ck.ref()
    if ref is not None and w * h < 500000:
        r = pre.copy()
        assert ck.call(ref, "ref_bilat", piece, d, img, r) == 0
        assert int((ck.ulp_diff(got, r) > 0).sum()) == 0
