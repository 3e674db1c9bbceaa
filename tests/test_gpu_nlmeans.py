"""-m gpu: the non-local-means core (nlm_chunks) through both modules that use it, bit for bit
against the CPU checkers: denoise (non-local means) on Lab input and denoise (profiled) in
non-local-means mode on RGB input."""
import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
from ansel_amd import abi, params, synth

pytestmark = pytest.mark.gpu


def _lab_image(w, h, seed):
    rng = np.random.default_rng(seed)
    rgb = synth.rgba_image(w, h, seed=seed, lo=0.0, hi=1.0)
    lab = np.zeros((h, w, 4), np.float32)
    lab[..., 0] = 100.0 * rgb[..., 1] + rng.normal(0, 1.5, (h, w))
    lab[..., 1] = 80.0 * (rgb[..., 0] - rgb[..., 1]) + rng.normal(0, 2.0, (h, w))
    lab[..., 2] = 80.0 * (rgb[..., 1] - rgb[..., 2]) + rng.normal(0, 2.0, (h, w))
    return np.ascontiguousarray(lab.astype(np.float32))


def _noisy(w, h, seed):
    rng = np.random.default_rng(seed)
    img = synth.rgba_image(w, h, seed=seed, lo=0.0, hi=0.9)
    img[..., :3] += rng.normal(0.0, 0.01, size=(h, w, 3)).astype(np.float32) * np.sqrt(np.maximum(img[..., :3], 0.01))
    return np.ascontiguousarray(img.astype(np.float32))


@pytest.fixture
def dispatch():
    """dt_hip_test_dispatch(): a fallback kernel on a frame the primary kernel takes; cleared behind the test"""
    from ansel_amd import lib
    keys = []

    def force(key, value=1):
        lib.test_dispatch(key, value)
        keys.append(key)
    yield force
    for k in keys:
        lib.test_dispatch(k, 0)


def _check(op, piece, d, img):
    got = hc.run_hip("dt_hip_iop_%s_process" % op, piece, d, img, img.shape)
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_" + op, piece, d, img, want) == 0
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d values differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
    ref = ck.ref()
    if ref is not None:
        r = np.zeros_like(img)
        assert ck.call(ref, "ref_" + op, piece, d, img, r) == 0
        assert int((ck.ulp_diff(got, r) > 0).sum()) == 0
    return got


@pytest.mark.parametrize("w,h", [(150, 131), (73, 61), (300, 64), (503, 397)])
@pytest.mark.parametrize("radius,strength,luma,chroma", [(2.0, 50.0, 0.5, 1.0), (1.0, 20.0, 1.0, 1.0), (3.0, 200.0, 0.3, 0.8)])
def test_nlmeans(w, h, radius, strength, luma, chroma):
    img = _lab_image(w, h, 17)
    got = _check("nlmeans", abi.Piece.make(w, h), abi.NlmeansData(radius, strength, luma, chroma), img)
    assert float(np.abs(got[..., :3] - img[..., :3]).max()) > 1e-3


@pytest.mark.parametrize("w,h,luma,chroma", [(330, 168, 0.5, 1.0), (260, 168, 1.0, 1.0), (170, 150, 0.3, 0.8), (1200, 560, 0.5, 1.0)])
def test_nlmeans_third_version_chunk_grids(w, h, luma, chroma, dispatch):
    """frames whose chunk grid the third version of the interior-chunk kernel takes (nlm3_body.h: chunks of at most 56
    rows, the module's defaults); the same frames through the second version (the "nlm_v2" test hook) give the same words"""
    img = _lab_image(w, h, 31)
    d = abi.NlmeansData(2.0, 50.0, luma, chroma)
    got3 = _check("nlmeans", abi.Piece.make(w, h), d, img)
    dispatch("nlm_v2")
    got2 = hc.run_hip("dt_hip_iop_nlmeans_process", abi.Piece.make(w, h), d, img, img.shape)
    assert np.array_equal(got2.view(np.uint32), got3.view(np.uint32))


# (width, height) -> chunk: (260, 192) 72 x 64 (the 45 MP / 60 MP frames' grid); (330, 171) 72 x 57; (170, 183) 64 x 61;
# (700, 315) 72 x 63; (1200, 640) 72 x 64, 16 x 10 chunks; (150, 128) 64-row chunks all in the border ring
@pytest.mark.parametrize("w,h,luma,chroma", [(260, 192, 0.5, 1.0), (330, 171, 1.0, 1.0), (170, 183, 0.3, 0.8), (700, 315, 0.5, 1.0),
                                             (1200, 640, 0.5, 1.0), (150, 128, 0.5, 1.0), (293, 247, 0.5, 1.0)])
def test_nlmeans_fused_variant_chunk_grids(w, h, luma, chroma, dispatch):
    """frames whose chunks have 57 - 64 rows: the fused variant of the third version (nlm3_body.h FUSED, nlm_chunks_v4:
    three tables, the row recurrence inside the weights' waves); the second version (the "nlm_v2" test hook) gives the same
    words"""
    img = _lab_image(w, h, 37)
    d = abi.NlmeansData(2.0, 50.0, luma, chroma)
    got4 = _check("nlmeans", abi.Piece.make(w, h), d, img)
    dispatch("nlm_v2")
    got2 = hc.run_hip("dt_hip_iop_nlmeans_process", abi.Piece.make(w, h), d, img, img.shape)
    assert np.array_equal(got2.view(np.uint32), got4.view(np.uint32))


# (width, height) -> chunk: (260, 207) 72 x 69 (the 24 MP frame's grid); (260, 204) 72 x 68 (42 MP); (170, 201) 64 x 67 (150 MP's
# height); (260, 198) 72 x 66; (250, 195) 68 x 65: one tail row; (1200, 690) 72 x 69, 17 x 10 chunks, 8 x 15 of them interior;
# (150, 138) 69-row chunks all in the border ring; (400, 483) 69 rows, SEVEN chunk rows, the last of them the frame's last;
# (151, 274) 64 x 69, the last row of chunks 67 rows (a tail of three) and the last 23 columns; (200, 196) 68 x 66, the last row of
# chunks 64 rows: a head without a tail.  The outermost ring runs the BORDER bodies of head and tail
@pytest.mark.parametrize("w,h,luma,chroma", [(260, 207, 0.5, 1.0), (260, 204, 1.0, 1.0), (170, 201, 0.3, 0.8), (260, 198, 0.5, 1.0),
                                             (250, 195, 0.5, 0.9), (1200, 690, 0.5, 1.0), (150, 138, 0.5, 1.0), (400, 483, 0.5, 1.0),
                                             (151, 274, 0.5, 0.9), (200, 196, 0.5, 1.0)])
def test_nlmeans_tall_chunk_grids(w, h, luma, chroma, dispatch):
    """frames whose chunks have 65 - 69 rows (24 MP: 69, 42 MP: 68, 150 MP: 67): the fused variant on the first 64 rows of
    every interior chunk + nlm_tail on the rows that are left (nlm_tail_body.h, round 5); the second version (the "nlm_v2"
    test hook), which took these frames alone until then, gives the same words"""
    img = _lab_image(w, h, 43)
    d = abi.NlmeansData(2.0, 50.0, luma, chroma)
    got = _check("nlmeans", abi.Piece.make(w, h), d, img)
    dispatch("nlm_v2")
    got2 = hc.run_hip("dt_hip_iop_nlmeans_process", abi.Piece.make(w, h), d, img, img.shape)
    assert np.array_equal(got2.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("w,h", [(330, 168), (1200, 560)])
def test_nlmeans_fused_variant_on_the_third_versions_grids(w, h, dispatch):
    """the "nlm_fused" test hook: the fused variant on chunk grids the third version takes"""
    img = _lab_image(w, h, 41)
    d = abi.NlmeansData(2.0, 50.0, 0.5, 1.0)
    dispatch("nlm_fused")
    _check("nlmeans", abi.Piece.make(w, h), d, img)


@pytest.mark.parametrize("radius", [5.0, 9.0])
def test_nlmeans_large_patch_radius_takes_the_fallback_kernels(radius):
    """patch radii whose column-sum table is wider than the pipelined kernel's fixed pitch (P >= 5) run the
    barrier-per-step kernel, with the window staged in LDS while it fits and read from global beyond"""
    w, h = 230, 150
    img = _lab_image(w, h, 29)
    _check("nlmeans", abi.Piece.make(w, h), abi.NlmeansData(radius, 80.0, 0.7, 0.9), img)


def test_nlmeans_scaled_roi():
    w, h = 240, 170
    img = _lab_image(w, h, 5)
    piece = abi.Piece.make(w, h, roi_in=abi.Roi.make(0, 0, w, h, 0.5), roi_out=abi.Roi.make(0, 0, w, h, 0.5))
    _check("nlmeans", piece, abi.NlmeansData(2.0, 50.0, 0.5, 1.0), img)


@pytest.mark.parametrize("over", [dict(), dict(use_new_vst=False), dict(use_new_vst=False, fix=False),
                                  dict(radius=2.0, nbhood=5.0, scattering=0.6, central_pixel_weight=0.5, strength=1.3),
                                  dict(wb_adaptive=False, shadows=0.5, bias=-2.0, nbhood=3.0)])
def test_denoiseprofile_nlmeans(over):
    w, h = 320, 231
    img = _noisy(w, h, 23)
    d = params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS, **over)
    _check("denoiseprofile", abi.Piece.make(w, h, processed_maximum=synth.WB_COEFFS), d, img)


# ---- round 6: patch radius 1 and the weight with the centre pixel's term on the third version's schedule and its fused variant
#      (nlm3_body.h P / CENTER).  Chunk grids: (330, 168) / (1200, 560) 72 x 56 -- the 100 MP frame's --, (170, 150) 64 x 51: the third
#      version; (260, 192) / (1200, 640) 72 x 64 -- the 45 / 60 MP frames' --, (330, 171) 72 x 57, (170, 183) 64 x 61: fused;
#      (150, 128): every chunk in the outermost ring; (293, 247): odd width, the second version (63-row chunks, 69 columns ...).
#      tall grids (the fused head + nlm_tail): (260, 207) 72 x 69 -- the 24 MP frame's --, (170, 201) 64 x 67, (250, 195) 68 x 65: one tail
#      row, (1200, 690) 17 x 10 chunks, (150, 138): 69-row chunks all in the ring, (151, 274): a last row of chunks of 67 rows
R6_GRIDS = [(330, 168), (170, 150), (1200, 560), (260, 192), (330, 171), (170, 183), (1200, 640), (150, 128), (293, 247),
            (260, 207), (170, 201), (250, 195), (1200, 690), (150, 138), (151, 274)]


@pytest.mark.parametrize("w,h", R6_GRIDS)
def test_nlmeans_patch_radius_one_on_the_third_version(w, h, dispatch):
    """denoise (non-local means) with patch radius 1 (nine A1 chains of <= 7 terms): oracle, reference, and the second version"""
    img = _lab_image(w, h, 47)
    d = abi.NlmeansData(1.0, 35.0, 0.5, 1.0)
    got = _check("nlmeans", abi.Piece.make(w, h), d, img)
    dispatch("nlm_v2")
    got2 = hc.run_hip("dt_hip_iop_nlmeans_process", abi.Piece.make(w, h), d, img, img.shape)
    assert np.array_equal(got2.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("w,h", R6_GRIDS)
@pytest.mark.parametrize("over", [dict(),  # the module's defaults in this mode: patch radius 1, search radius 7, central pixel weight 0.1
                                  dict(radius=2.0, nbhood=5.0, central_pixel_weight=0.5, strength=1.3),
                                  dict(radius=1.0, nbhood=4.0, central_pixel_weight=0.0, use_new_vst=False)])
def test_denoiseprofile_nlmeans_on_the_third_version(w, h, over, dispatch):
    """denoise (profiled), non-local-means mode (the weight with the centre pixel's term, nlmeans_core.c:416-424) on the chunk grids
    of the third version and of its fused variant: oracle, reference, and the second version's body (IEEE division)"""
    img = _noisy(w, h, 53)
    d = params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS, **over)
    piece = abi.Piece.make(w, h, processed_maximum=synth.WB_COEFFS)
    got = _check("denoiseprofile", piece, d, img)
    dispatch("nlm_v2")
    got2 = hc.run_hip("dt_hip_iop_denoiseprofile_process", piece, d, img, img.shape)
    assert np.array_equal(got2.view(np.uint32), got.view(np.uint32))


def test_denoiseprofile_nlmeans_with_adversarial_samples(dispatch):
    """non-finite and extreme samples through the centre term's division (div_uniform(): v_div_fixup_f32 takes the infinite and NaN
    numerators): the same words as the oracle and as the second version's IEEE division"""
    w, h = 330, 168
    img = _noisy(w, h, 59)
    img[40:44, 100:104, 0] = np.float32(3.0e38)
    img[90, 200, 1] = np.float32(np.inf)
    img[91, 201, 2] = np.float32(np.nan)
    img[120:123, 50:60, :3] = np.float32(1e-30)
    img[10, 300:310, :3] = np.float32(-0.5)
    d = params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS)
    piece = abi.Piece.make(w, h, processed_maximum=synth.WB_COEFFS)
    got = hc.run_hip("dt_hip_iop_denoiseprofile_process", piece, d, img, img.shape)
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_denoiseprofile", piece, d, img, want) == 0
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "%d words differ" % int((~same).sum())
    dispatch("nlm_v2")
    got2 = hc.run_hip("dt_hip_iop_denoiseprofile_process", piece, d, img, img.shape)
    same = (got.view(np.uint32) == got2.view(np.uint32)) | (np.isnan(got) & np.isnan(got2))
    assert same.all()
