"""-m gpu: the blend stage (all four blend colourspaces: uniform / parametric mask, tone curve, 16 + 30 + 27 + 17 operators) through the
C-ABI, bit for bit against the oracle and the reference's own functions."""
import ctypes as C

import numpy as np
import pytest

import blend_cases
import checkers as ck
import hipcheck as hc
from ansel_amd import abi, lib

pytestmark = pytest.mark.gpu
CASES = blend_cases.cases()


def _check(piece, d, a, b, what):
    got = hc.run_hip("dt_hip_develop_blend_process", piece, d, a, b.shape, pre_fill=b)
    want = b.copy()
    assert ck.call(ck.oracle(), "oracle_develop_blend", piece, d, a, want) == 0
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%s: %d values differ, max %d ulp" % (what, int((diff > 0).sum()), int(diff.max()))
    ref = ck.ref()
    if ref is not None:
        r = b.copy()
        assert ck.call(ref, "ref_develop_blend", piece, d, a, r) == 0
        assert int((ck.ulp_diff(got, r) > 0).sum()) == 0, what + " (vs reference)"
    return got


@pytest.mark.parametrize("name,d", CASES, ids=[c[0] for c in CASES])
def test_blend(name, d):
    w, h = 131, 67
    a, b = blend_cases.images(w, h, 41)
    got = _check(abi.Piece.make(w, h), d, a, b, name)
    if name.startswith("param") or name == "multi-c0-0-0":
        # the parametric masks of the test set are real ones: some pixels in, some out, some in between
        m = got[..., 3][np.isfinite(got[..., 3])]
        assert m.min() < m.max(), name


LAB_CASES = blend_cases.lab_cases()


@pytest.mark.parametrize("name,d", LAB_CASES, ids=[c[0] for c in LAB_CASES])
def test_blend_lab(name, d):
    w, h = 131, 67
    a, b = blend_cases.lab_images(w, h, 51)
    got = _check(abi.Piece.make(w, h), d, a, b, name)
    if name.startswith("lab-param") and not name.endswith("-inv"):
        m = got[..., 3][np.isfinite(got[..., 3])]
        assert m.min() < m.max(), name


@pytest.mark.parametrize("mode", abi.BLEND_LAB_REFUSED)
def test_blend_lab_refuses_the_lch_operators(mode):
    w, h = 32, 16
    a, b = blend_cases.lab_images(w, h, 3)
    d = abi.BlendData.uniform(blend_cases.M, 50.0, mode, blend_cst=abi.BLEND_CS_LAB)
    h_ = hc.hip()
    da, db = lib.DeviceBuffer.from_numpy(0, a), lib.DeviceBuffer.from_numpy(0, b)
    assert h_.dt_hip_develop_blend_process(0, C.byref(abi.Piece.make(w, h)), C.byref(d), da.ptr, db.ptr) == -997
    assert h_.dt_hip_finish(0) == 1


DSP_CASES = blend_cases.display_cases()


@pytest.mark.parametrize("name,d", DSP_CASES, ids=[c[0] for c in DSP_CASES])
def test_blend_display(name, d):
    w, h = 131, 67
    a, b = blend_cases.display_images(w, h, 71)
    _check(abi.Piece.make(w, h), d, a, b, name)


BLUR_CASES = blend_cases.blur_cases()


@pytest.mark.parametrize("name,d,kind", BLUR_CASES, ids=[c[0] for c in BLUR_CASES])
@pytest.mark.parametrize("w,h", [(131, 67), (40, 3), (1, 50), (700, 500)])
def test_blend_mask_blur(name, d, kind, w, h):
    """the mask through the recursive gaussian (src/pixel/gaussian.c) between make_mask and the tone curve"""
    if w > 20 and h > 20:
        a, b = blend_cases.images_for(kind, w, h, 81)
    else:
        a, b = [np.ascontiguousarray(z[:h, :w]) for z in blend_cases.images_for(kind, 64, 64, 81)]
    _check(abi.Piece.make(w, h, channels=1 if kind == "raw" else 4), d, a, b, name)


RAW_CASES = blend_cases.raw_cases()


@pytest.mark.parametrize("name,d", RAW_CASES, ids=[c[0] for c in RAW_CASES])
def test_blend_raw(name, d):
    w, h = 133, 65
    a, b = blend_cases.raw_images(w, h, 61)
    _check(abi.Piece.make(w, h, channels=1), d, a, b, name)


def test_blend_roi_offset():
    w, h, iw, ih = 90, 50, 120, 70
    a, b = blend_cases.images(w, h, 43, iw, ih)
    piece = abi.Piece.make(w, h, roi_in=abi.Roi.make(10, 20, iw, ih, 1.0), roi_out=abi.Roi.make(25, 31, w, h, 1.0))
    d = abi.BlendData.uniform(blend_cases.M, 70.0).channel(abi.BLENDIF_GRAY_in, 0.05, 0.3, 0.8, 1.0)
    _check(piece, d, a, b, "roi offset")


def test_blend_full_frame():
    """24 MP, every kind of channel active"""
    w, h = 6000, 4000
    a, b = blend_cases.images(w, h, 47)
    d = dict(CASES)["multi-c0-0.4--0.3"]
    _check(abi.Piece.make(w, h), d, a, b, "24 MP")


@pytest.mark.parametrize("field,value", [("blend_cst", 5), ("blend_cst", 0),
                                         ("mask_mode", abi.MASK_ENABLED | abi.MASK_SHAPE),
                                         ("mask_mode", abi.MASK_ENABLED | abi.MASK_RASTER)])
def test_blend_refuses_what_is_not_built(field, value):
    """never an approximation: unsupported mask sources and colourspaces are an error"""
    w, h = 32, 16
    a, b = blend_cases.images(w, h, 3)
    d = abi.BlendData.uniform(blend_cases.M, 50.0)
    setattr(d, field, value)
    h_ = hc.hip()
    da, db = lib.DeviceBuffer.from_numpy(0, a), lib.DeviceBuffer.from_numpy(0, b)
    rc = h_.dt_hip_develop_blend_process(0, C.byref(abi.Piece.make(w, h)), C.byref(d), da.ptr, db.ptr)
    assert rc == -997  # DT_HIP_INVALID_ARG
    assert h_.dt_hip_finish(0) == 1
    assert np.array_equal(db.to_numpy(b.shape, np.float32).view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("mode", ["uniform", "disabled", "parametric"])
def test_a_stale_details_value_without_the_raw_detail_mask_is_ignored(mode):
    """blend.c:732-790 reads `details` only behind use_masks && !raster_only, and _refine_with_detail_mask() returns
    silently when the pipe holds no raw detail mask (:379): such a blend equals the same blend with details = 0"""
    w, h = 48, 20
    a, b = blend_cases.images(w, h, 5)
    outs = []
    for details in (0.0, 0.5):
        if mode == "parametric":
            d = dict(CASES)["multi-c0-0.4--0.3"]
            d = type(d).from_buffer_copy(d)
        else:
            d = abi.BlendData.uniform(blend_cases.M, 50.0)
            if mode == "disabled":
                d.mask_mode = 0
        d.details = details
        h_ = hc.hip()
        da, db = lib.DeviceBuffer.from_numpy(0, a), lib.DeviceBuffer.from_numpy(0, b)
        rc = h_.dt_hip_develop_blend_process(0, C.byref(abi.Piece.make(w, h)), C.byref(d), da.ptr, db.ptr)
        assert rc == 0, h_.dt_hip_last_error()
        assert h_.dt_hip_finish(0) == 1
        outs.append(db.to_numpy(b.shape, np.float32).view(np.uint32))
    assert np.array_equal(outs[0], outs[1])


# ---- drawn / raster masks and the details threshold: the host-rendered form mask (blend.c:740-790, :1278-1325) -------
FORM_CASES = [(cs, n, d) for cs in (abi.BLEND_CS_RGB_SCENE, abi.BLEND_CS_RGB_DISPLAY, abi.BLEND_CS_LAB)
              for n, d in blend_cases.form_cases(cs)]


@pytest.mark.parametrize("cs,name,d", FORM_CASES, ids=["cs%d-%s" % (c[0], c[1]) for c in FORM_CASES])
def test_blend_with_a_host_rendered_form_mask(cs, name, d):
    """the plane is uploaded once (what blend.c:1285 does with dt_opencl_write_host_to_device) and takes the place of
    the constant form mask; device == oracle == reference"""
    l = hc.hip()
    w, h = 131, 67
    a, b = blend_cases.lab_images(w, h, 53) if cs == abi.BLEND_CS_LAB else blend_cases.images(w, h, 44)
    form = blend_cases.form_plane(w, h)
    dform = lib.DeviceBuffer.from_numpy(0, form)
    piece = abi.Piece.make(w, h)
    d.form_mask = dform.ptr
    got = hc.run_hip("dt_hip_develop_blend_process", piece, d, a, b.shape, pre_fill=b)
    host_form = ck.aligned_empty(form.shape, np.float32)
    host_form[...] = form
    d.form_mask = host_form.ctypes.data
    want = b.copy()
    assert ck.call(ck.oracle(), "oracle_develop_blend", piece, d, a, want) == 0
    assert int((ck.ulp_diff(got, want) > 0).sum()) == 0, name
    ref = ck.ref()
    if ref is not None:
        r = b.copy()
        assert ck.call(ref, "ref_develop_blend", piece, d, a, r) == 0
        assert int((ck.ulp_diff(got, r) > 0).sum()) == 0, name + " (vs reference)"
    # without the plane: refused, with the reason
    d.form_mask = None
    buf = lib.DeviceBuffer.from_numpy(0, b)
    din = lib.DeviceBuffer.from_numpy(0, a)
    rc = l.dt_hip_develop_blend_process(0, C.byref(piece), C.byref(d), din.ptr, buf.ptr)
    if d.mask_mode & (abi.MASK_SHAPE | abi.MASK_RASTER):
        assert rc == abi.DT_HIP_INVALID_ARG
        assert b"form_mask" in l.dt_hip_last_error()
    else:
        # a parametric-only blend whose details threshold has neither plane: the reference's _refine_with_detail_mask()
        # returns silently without the raw detail mask (blend.c:379) -- the blend runs unrefined
        assert rc == 0
    for x in (dform, buf, din):
        x.release()


FEATHER_CASES = blend_cases.feather_cases()


@pytest.mark.parametrize("name,d,kind", FEATHER_CASES, ids=[c[0] for c in FEATHER_CASES])
@pytest.mark.parametrize("w,h", [(131, 67), (640, 530), (40, 3), (1, 50), (1300, 1100)])
def test_blend_mask_feathering(name, d, kind, w, h):
    """the guided filter over the mask (src/pixel/guided_filter.c: its 512-pixel tile grid, the Kahan box means of
    box_filters.c and their 1-wide tail), guided by the module's input or output, before / after the blur: device ==
    oracle == the reference's own code"""
    if w > 20 and h > 20:
        a, b = blend_cases.images_for(kind, w, h, 83)
    else:
        a, b = [np.ascontiguousarray(z[:h, :w]) for z in blend_cases.images_for(kind, 64, 64, 83)]
    _check(abi.Piece.make(w, h, channels=1 if kind == "raw" else 4), d, a, b, name)


def test_blend_feathering_with_a_form_mask():
    """a drawn mask, a parametric condition on top, feathered along the edges of the module's output"""
    l = hc.hip()
    w, h = 700, 560
    a, b = blend_cases.images(w, h, 44)
    form = blend_cases.form_plane(w, h)
    dform = lib.DeviceBuffer.from_numpy(0, form)
    d = dict(blend_cases.form_cases(abi.BLEND_CS_RGB_SCENE))["drawn+raster+parametric-c0"]
    d.feathering_radius, d.feathering_guide = 6.0, abi.MASK_GUIDE_OUT_AFTER_BLUR
    piece = abi.Piece.make(w, h)
    d.form_mask = dform.ptr
    got = hc.run_hip("dt_hip_develop_blend_process", piece, d, a, b.shape, pre_fill=b)
    host_form = ck.aligned_empty(form.shape, np.float32)
    host_form[...] = form
    d.form_mask = host_form.ctypes.data
    want = b.copy()
    assert ck.call(ck.oracle(), "oracle_develop_blend", piece, d, a, want) == 0
    assert int((ck.ulp_diff(got, want) > 0).sum()) == 0
    dform.release()
    assert l.dt_hip_finish(0) == 1


def test_feathering_guided_by_a_larger_input_is_refused():
    """FEATHER_IN with roi_in != roi_out: the reference reads past its input (blend.c:823-824); refused with the reason"""
    l = hc.hip()
    w, h, iw, ih = 90, 50, 120, 70
    a, b = blend_cases.images(w, h, 43, iw, ih)
    piece = abi.Piece.make(w, h, roi_in=abi.Roi.make(10, 20, iw, ih, 1.0), roi_out=abi.Roi.make(25, 31, w, h, 1.0))
    d = abi.BlendData.uniform(blend_cases.M, 70.0).channel(abi.BLENDIF_GRAY_in, 0.05, 0.3, 0.8, 1.0)
    d.feathering_radius, d.feathering_guide = 3.0, abi.MASK_GUIDE_IN_BEFORE_BLUR
    buf, din = lib.DeviceBuffer.from_numpy(0, b), lib.DeviceBuffer.from_numpy(0, a)
    assert l.dt_hip_develop_blend_process(0, C.byref(piece), C.byref(d), din.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG
    assert b"roi_in == roi_out" in l.dt_hip_last_error()
    d.feathering_guide = abi.MASK_GUIDE_OUT_AFTER_BLUR  # guided by the output: any roi
    _check(piece, d, a, b, "feathering guided by the output under a roi offset")


# ---- the details threshold on the device: the hidden "detailmask" stage and _refine_with_detail_mask() ---------------
@pytest.mark.parametrize("w,h,wb", [(1200, 801, (2.1, 1.0, 1.6)), (37, 23, (1.0, 1.0, 1.0)), (9, 9, (1.7, 1.0, 1.2)), (3, 3, (1.0, 1.0, 1.0))])
def test_detailmask_stage(w, h, wb):
    """src/iop/detailmask.c: the input copied, the raw detail mask (dt_masks_calc_rawdetail_mask) in the side-band plane;
    device == oracle == reference, negative and non-finite samples included"""
    from ansel_amd import synth
    img = synth.rgba_image(w, h, seed=8, lo=-0.05, hi=1.4)
    if w > 10:
        img[3, 4, 0] = np.nan
        img[5, 6, 1] = np.inf
        img[7, 2, 2] = -3.0
    piece = abi.Piece.make(w, h)
    dplane = lib.DeviceBuffer.from_numpy(0, np.full((h, w), -9.0, np.float32))
    got = hc.run_hip("dt_hip_iop_detailmask_process", piece, abi.DetailmaskData.make(wb, dplane.ptr), img, img.shape)
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))
    plane = dplane.to_numpy((h, w), np.float32)
    for which in hc.checkers_available():
        want = ck.aligned_empty((h, w), np.float32)
        l = ck.ref() if which == "ref" else ck.oracle()
        assert ck.call(l, which + "_detailmask", piece, abi.DetailmaskData.make(wb, want.ctypes.data), img, np.zeros_like(img)) == 0
        assert int((ck.ulp_diff(plane, want) > 0).sum()) == 0, which
    dplane.release()


DETAIL_CASES = [(cs, n, d, f) for cs in (abi.BLEND_CS_RGB_SCENE, abi.BLEND_CS_RGB_DISPLAY, abi.BLEND_CS_LAB)
                for n, d, f in blend_cases.detail_cases(cs)]


@pytest.mark.parametrize("cs,name,d,with_form", DETAIL_CASES, ids=["cs%d-%s" % (c[0], c[1]) for c in DETAIL_CASES])
def test_blend_details_threshold_from_the_raw_detail_mask(cs, name, d, with_form):
    """the stage's plane stays on the device; the blend refines its form mask with it (blend.c:361-425)"""
    from ansel_amd import synth
    w, h = 131, 67
    a, b = blend_cases.lab_images(w, h, 53) if cs == abi.BLEND_CS_LAB else blend_cases.images(w, h, 44)
    rgb = blend_cases.images(w, h, 44)[0]
    piece = abi.Piece.make(w, h)
    drm = lib.DeviceBuffer(0, w * h * 4)
    hc.run_hip("dt_hip_iop_detailmask_process", piece, abi.DetailmaskData.make((2.0, 1.0, 1.5), drm.ptr), rgb, rgb.shape)
    rm = ck.aligned_empty((h, w), np.float32)
    rm[...] = drm.to_numpy((h, w), np.float32)
    form = host_form = dform = None
    if with_form:
        form = blend_cases.form_plane(w, h)
        dform = lib.DeviceBuffer.from_numpy(0, form)
        host_form = ck.aligned_empty(form.shape, np.float32)
        host_form[...] = form
    d.detail_mask, d.form_mask = drm.ptr, (dform.ptr if with_form else None)
    got = hc.run_hip("dt_hip_develop_blend_process", piece, d, a, b.shape, pre_fill=b)
    d.detail_mask, d.form_mask = rm.ctypes.data, (host_form.ctypes.data if with_form else None)
    for which in hc.checkers_available():
        want = b.copy()
        l = ck.ref() if which == "ref" else ck.oracle()
        assert ck.call(l, which + "_develop_blend", piece, d, a, want) == 0
        assert int((ck.ulp_diff(got, want) > 0).sum()) == 0, name + " vs " + which
    drm.release()
    if dform:
        dform.release()


def test_blend_details_threshold_full_frame():
    """24 MP: the stage and a drawn + parametric mask refined by it"""
    from ansel_amd import synth
    w, h = 6000, 4000
    img = synth.rgba_image(1024, 1024, seed=3, lo=0.0, hi=1.0)
    a = np.tile(img, (4, 6, 1))[:h, :w].copy()
    b = np.ascontiguousarray(a[::-1])
    piece = abi.Piece.make(w, h)
    drm = lib.DeviceBuffer(0, w * h * 4)
    hc.run_hip("dt_hip_iop_detailmask_process", piece, abi.DetailmaskData.make((2.0, 1.0, 1.5), drm.ptr), a, a.shape)
    rm = ck.aligned_empty((h, w), np.float32)
    rm[...] = drm.to_numpy((h, w), np.float32)
    d = abi.BlendData.uniform(blend_cases.M, 85.0).channel(abi.BLENDIF_GRAY_in, 0.05, 0.2, 0.6, 0.9, boost=1.0)
    d.details = 0.3
    d.detail_mask = drm.ptr
    got = hc.run_hip("dt_hip_develop_blend_process", piece, d, a, b.shape, pre_fill=b)
    d.detail_mask = rm.ctypes.data
    want = b.copy()
    assert ck.call(ck.oracle(), "oracle_develop_blend", piece, d, a, want) == 0
    assert int((ck.ulp_diff(got, want) > 0).sum()) == 0
    drm.release()
