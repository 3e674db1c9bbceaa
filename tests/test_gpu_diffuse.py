"""-m gpu: diffuse or sharpen (dt_hip_iop_diffuse_process) against the CPU checkers, bit for bit.

Sizes and radii are chosen so that every analysis kernel shape runs (dilation 1, 2, 4, >= 8 with
several column groups), dilations exceed the frame (rows <= dilation), several iterations
ping-pong, and all three kernel kinds (isotrope / isophote / gradient) appear in every order."""
import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
from ansel_amd import abi, params, synth

pytestmark = pytest.mark.gpu

CASES = [
    ("default", {}, (333, 217)),
    ("default", dict(sharpness=0.5, radius=16, first=0.3, third=-0.2), (640, 401)),
    ("lens_deblur_soft", dict(iterations=4), (500, 300)),
    ("lens_deblur_soft", dict(iterations=2, anisotropy_first=-2.0, anisotropy_second=1.5, anisotropy_fourth=-3.0,
                              variance_threshold=-0.5, regularization=2.5), (417, 283)),
    ("fast_local_contrast", {}, (700, 500)),            # 10 scales, dilation up to 512 > frame height
    ("fast_local_contrast", dict(radius=100, radius_center=40), (1211, 160)),
    # frames narrower than a wave's 64 columns, than a landing zone's halo (the PDE's rows arrive by LDS-DMA: every slot of a
    # zone is a clamped column), one column short of / past a 256-column workgroup, and lower than a strip
    ("lens_deblur_soft", dict(iterations=2), (37, 29)),
    ("default", dict(radius=16), (9, 70)),
    ("lens_deblur_soft", dict(iterations=1), (255, 33)),
    ("default", dict(radius=16), (257, 19)),
]


def _cpu(which, piece, d, img):
    l = ck.ref() if which == "ref" else ck.oracle()
    if l is None:
        return None
    out = np.zeros(img.shape, np.float32)
    assert ck.call(l, ("ref_" if which == "ref" else "oracle_") + "diffuse", piece, d, img, out) == 0
    return out


def _with_alpha(img, kind):
    """the fourth channel is a channel like the others (diffuse.c loops over four); the device skips it in waves whose
    supports hold nothing but +0 there: planes that are blank, dense, blank in patches, and blank but for -0 / one sample"""
    h, w = img.shape[:2]
    rng = np.random.default_rng(h * 7 + w)
    if kind == "dense":
        img[..., 3] = rng.random((h, w), dtype=np.float32) * np.float32(1.3) - np.float32(0.1)
    elif kind == "patches":
        img[h // 3: 2 * h // 3, w // 4: w // 2, 3] = rng.random((2 * h // 3 - h // 3, w // 2 - w // 4), dtype=np.float32)
        img[: h // 5, -w // 6:, 3] = np.float32(0.25)
    elif kind == "sparse":
        img[h // 2, w // 2, 3] = np.float32(1.0)
        img[h // 4, : w // 2, 3] = np.float32(-0.0)
        img[3 * h // 4, w // 3, 3] = np.float32(np.nan)
    return img


@pytest.mark.parametrize("preset,over,size", CASES)
@pytest.mark.parametrize("imgname", ["scene", "adversarial", "scene-dense", "scene-patches", "scene-sparse"])
def test_diffuse_matches_cpu(preset, over, size, imgname):
    w, h = size
    img = synth.adversarial_rgba(w, h) if imgname == "adversarial" else synth.rgba_image(w, h, seed=6, lo=-0.02, hi=1.5)
    if "-" in imgname:
        img = _with_alpha(img, imgname.split("-")[1])
    piece = abi.Piece.make(w, h)
    d = params.diffuse(preset, **over)
    got = hc.run_hip("dt_hip_iop_diffuse_process", piece, d, img, img.shape)
    want = _cpu("oracle", piece, d, img)
    assert want is not None
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d px differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
    ref = _cpu("ref", piece, d, img)
    if ref is not None:
        assert int((ck.ulp_diff(got, ref) > 0).sum()) == 0


# the sign patterns of the reference's presets (init_presets(), diffuse.c:298-583): each takes a kernel with the orders'
# kinds spelled at compile time; the last mixes three kinds with unequal anisotropies and takes the run-time kernel
PATTERNS = [
    ("isophote x4", dict(anisotropy_first=4.0, anisotropy_second=4.0, anisotropy_third=4.0, anisotropy_fourth=4.0)),
    ("gradient x4", dict(anisotropy_first=-5.0, anisotropy_second=-5.0, anisotropy_third=-5.0, anisotropy_fourth=-5.0)),
    ("watercolor", dict(anisotropy_third=4.0, anisotropy_fourth=4.0)),
    ("local contrast", dict(anisotropy_first=-2.5, anisotropy_fourth=-2.5)),
    ("inpaint", dict(anisotropy_fourth=2.0)),
    ("fast", dict(anisotropy_third=5.0)),
    ("deblur", dict(anisotropy_first=2.0, anisotropy_third=2.0)),
    ("isotropic", {}),
    ("same kinds, other anisotropies", dict(anisotropy_first=2.0, anisotropy_third=3.0)),
    ("mixed", dict(anisotropy_first=-1.0, anisotropy_second=2.0, anisotropy_third=0.5, anisotropy_fourth=-0.25)),
]


@pytest.mark.parametrize("name,aniso", PATTERNS, ids=[p[0] for p in PATTERNS])
def test_diffuse_kind_patterns(name, aniso):
    w, h = 389, 251
    img = _with_alpha(synth.rgba_image(w, h, seed=16, lo=-0.02, hi=1.5), "patches")
    piece = abi.Piece.make(w, h)
    d = params.diffuse("default", iterations=2, radius=12, regularization=1.5, variance_threshold=0.3, sharpness=0.1,
                       first=-0.25, second=0.125, third=-0.125, fourth=0.0625, **aniso)
    got = hc.run_hip("dt_hip_iop_diffuse_process", piece, d, img, img.shape)
    want = _cpu("oracle", piece, d, img)
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d px differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))


def _extreme_rgba(w, h, seed=23):
    """what the in-range forms of the PDE's divisions and square roots must hand back to the long forms (round 5,
    ansel_amd/csrc/ieee_inrange.h): magnitudes beyond 2^64, infinities and NaNs among the samples, dark pixels one ulp
    apart (squared gradients in (0, 2^-96)), subnormals, and ordinary pixels in between (every wave mixes them)"""
    rng = np.random.default_rng(seed)
    img = synth.rgba_image(w, h, seed=seed, lo=0.0, hi=1.2)
    h8 = max(h // 8, 1)
    img[0 * h8:1 * h8, :, :3] *= np.float32(2.0 ** 70)                       # quotients near the overflow scaling
    img[1 * h8:2 * h8, ::7, 0] = np.float32(np.inf)
    img[1 * h8:2 * h8, 3::11, 1] = np.float32(np.nan)
    img[1 * h8:2 * h8, 5::13, 2] = np.float32(-np.inf)
    base = np.float32(3e-9)
    ulps = rng.integers(0, 3, (min(h8, h - 2 * h8), w, 3)).astype(np.uint32)
    img[2 * h8:3 * h8, :, :3] = (np.full(ulps.shape, base, np.float32).view(np.uint32) + ulps).view(np.float32)  # gradients of an ulp
    img[3 * h8:4 * h8, :, :3] = (1e-39 * rng.random((min(h8, h - 3 * h8), w, 3))).astype(np.float32)  # subnormals
    img[4 * h8:5 * h8, :, :3] *= np.float32(1e-30)
    img[5 * h8:6 * h8, ::5, :3] = np.float32(3.0e38)
    img[5 * h8:6 * h8, 1::5, :3] = np.float32(0.0)
    # gradients whose squares are subnormal or tiny normals (values around 2^-63 .. 2^-55: squared gradient 2^-130 .. 2^-112)
    img[6 * h8:7 * h8, :, :3] = (rng.random((min(h8, h - 6 * h8), w, 3)) * 2.0 ** rng.integers(-63, -54, (min(h8, h - 6 * h8), w, 1))).astype(np.float32)
    return img


@pytest.mark.parametrize("preset,over", [("lens_deblur_soft", dict(iterations=2)), ("default", {}),
                                         ("default", dict(iterations=2, radius=12, anisotropy_first=4.0, anisotropy_second=-3.0,
                                                          anisotropy_third=2.0, anisotropy_fourth=-1.0, first=-0.25, second=0.125,
                                                          third=-0.125, fourth=0.0625))])
def test_diffuse_out_of_range_operands_take_the_long_forms(preset, over):
    w, h = 523, 331
    img = _extreme_rgba(w, h)
    piece = abi.Piece.make(w, h)
    d = params.diffuse(preset, **over)
    got = hc.run_hip("dt_hip_iop_diffuse_process", piece, d, img, img.shape)
    want = _cpu("oracle", piece, d, img)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "%d values differ" % int((~same).sum())
    ref = _cpu("ref", piece, d, img)
    if ref is not None:
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        assert same.all()


INPAINT_CASES = [
    ("inpaint_highlights", dict(iterations=4, threshold=1.0), (333, 217)),
    ("inpaint_highlights", dict(iterations=1, threshold=0.25, radius=8, sharpness=0.2), (640, 401)),
    ("inpaint_highlights", dict(iterations=2, threshold=1.41), (417, 283)),          # the preset's threshold
    ("lens_deblur_soft", dict(iterations=3, threshold=0.6), (500, 300)),
    ("default", dict(threshold=1e-3), (97, 150)),                                      # nearly every pixel masked
]


@pytest.mark.parametrize("preset,over,size", INPAINT_CASES)
@pytest.mark.parametrize("imgname", ["scene", "adversarial"])
def test_diffuse_luminance_masked_inpainting(preset, over, size, imgname):
    """threshold > 0 (build_mask / inpaint_mask, diffuse.c:1106-1152): pixels above the threshold are seeded with
    Box-Muller noise drawn from a generator keyed on the pixel's position -- splitmix32, xoshiro128+, and the C
    library's logf / cosf / sinf, all restated on the device -- and only they are diffused"""
    w, h = size
    img = synth.rgba_image(w, h, seed=8, lo=-0.02, hi=1.8) if imgname == "scene" else synth.adversarial_rgba(w, h)
    piece = abi.Piece.make(w, h)
    d = params.diffuse(preset, **over)
    got = hc.run_hip("dt_hip_iop_diffuse_process", piece, d, img, img.shape)
    want = _cpu("oracle", piece, d, img)
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d values differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
    ref = _cpu("ref", piece, d, img)
    if ref is not None:
        assert int((ck.ulp_diff(got, ref) > 0).sum()) == 0
    if imgname == "scene":
        masked = (img[..., :3] > d.threshold).any(axis=2)
        assert 0 < masked.sum()
        assert not np.array_equal(got[masked], img[masked])


def test_diffuse_tiling_follows_the_reference():
    l = hc.hip()
    import ctypes as C
    t = abi.Tiling()
    d = params.diffuse("lens_deblur_soft")
    l.dt_hip_iop_diffuse_tiling(C.byref(abi.Piece.make(100, 100)), C.byref(d), C.byref(t))
    scales = ck.oracle().oracle_diffuse_scales(C.byref(abi.Piece.make(100, 100)), C.byref(d))
    assert t.overlap == 1 << scales and abs(t.factor - (6.0625 + scales)) < 1e-6


def test_diffuse_is_the_same_bits_every_run():
    """the PDE's support rows arrive by LDS-DMA behind hand-written waits (diffuse.hip, diffuse_pde_strip<.., true>): a missing
    wait would not fail every time.  Eight runs of the bench's preset on a frame of several workgroups and strips per dilation
    class, a NaN-filled output buffer in front of each: one result, and it is the oracle's"""
    w, h = 1211, 777
    img = synth.rgba_image(w, h, seed=11, lo=0.0, hi=1.3)
    piece = abi.Piece.make(w, h)
    d = params.diffuse("lens_deblur_soft", iterations=2)
    want = _cpu("oracle", piece, d, img)
    assert want is not None
    fill = np.full(img.shape, np.float32(np.nan), np.float32)
    for _ in range(8):
        got = hc.run_hip("dt_hip_iop_diffuse_process", piece, d, img, img.shape, pre_fill=fill)
        assert int((ck.ulp_diff(got, want) > 0).sum()) == 0
