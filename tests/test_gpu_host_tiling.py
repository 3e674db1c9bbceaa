"""-m gpu: dt_hip_default_process_tiling_ptp() -- default_process_tiling_cl() for modules that do not move pixels
(src/develop/tiling.c:842-1067): the frame stays in host memory, tiles go through the device.

A tile is an ordinary module run on a cropped frame, so the checker is the oracle run tile by tile over the plan
the reference computes (tests/tile_plan.py); for pointwise modules that must also be the untiled result."""
import ctypes as C

import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
import tile_plan as tp
from ansel_amd import abi, filmic, lib, params, synth

pytestmark = pytest.mark.gpu


def _tiling_of(l, op, piece, data):
    t = abi.Tiling()
    fn = {"demosaic": "dt_hip_iop_demosaic_tiling", "diffuse": "dt_hip_iop_diffuse_tiling",
          "nlmeans": "dt_hip_iop_nlmeans_tiling", "denoiseprofile": "dt_hip_iop_denoiseprofile_tiling"}.get(op)
    if fn:
        getattr(l, fn)(C.byref(piece), C.byref(data), C.byref(t))
    else:
        l.dt_hip_default_tiling(C.byref(piece), 0, C.byref(t))
    return t


def _case(op, w, h):
    rng = np.random.default_rng(11)
    if op == "demosaic":
        piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
        src = (rng.random((h, w)) * 0.8).astype(np.float32)
        return piece, abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_PPG, 0.0), src, (h, w, 4), 4, 16
    piece = abi.Piece.make(w, h, channels=4)
    src = (rng.random((h, w, 4)) * 1.2).astype(np.float32)
    data = {"exposure": lambda: abi.ExposureData(-0.01, 1.7), "filmicrgb": filmic.default_data,
            "diffuse": lambda: params.diffuse("lens_deblur_soft", iterations=1),
            "nlmeans": lambda: abi.NlmeansData(2.0, 50.0, 0.5, 1.0),
            "denoiseprofile": lambda: params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS)}[op]()
    if op == "nlmeans":
        src[..., 0] *= 80.0
        src[..., 1:3] = (src[..., 1:3] - 0.5) * 60.0
    return piece, data, src, (h, w, 4), 16, 16


@pytest.mark.parametrize("op,pointwise", [("exposure", True), ("filmicrgb", True), ("demosaic", False), ("diffuse", False),
                                          ("nlmeans", False), ("denoiseprofile", False)])
@pytest.mark.parametrize("w,h,frac", [(512, 384, 0.3), (333, 517, 0.12), (512, 384, 4.0)])
def test_host_tiling_equals_the_oracle_over_the_same_tiles(op, pointwise, w, h, frac):
    l = hc.hip()
    o = ck.oracle()
    assert o is not None
    piece, data, src, oshape, ib, ob = _case(op, w, h)
    t = _tiling_of(l, op, piece, data)
    avail = int(w * h * max(ib, ob) * max(t.factor_cl, 1.0) * frac)
    out = np.zeros(oshape, np.float32)
    rc = l.dt_hip_default_process_tiling_ptp(0, op.encode(), C.byref(piece), C.cast(C.byref(data), C.c_void_p), C.sizeof(data),
                                             C.byref(t), src.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), ib, ob,
                                             avail)
    lib.check(rc, "tiling " + op)
    mw, mh = C.c_int(0), C.c_int(0)
    l.dt_hip_get_device_max_image_size(0, C.byref(mw), C.byref(mh))
    plan = tp.plan(w, h, ib, ob, t, piece.filters, avail, l.dt_hip_get_device_memalloc(0), mw.value, mh.value)
    assert (plan["tiles_x"] * plan["tiles_y"] > 1) == (frac < 1.0)
    want = np.zeros(oshape, np.float32)
    tp.run_tiled(lambda p, i, ot: ck.call(o, "oracle_" + op, p, data, i, ot), piece, plan, src, want)
    hc.assert_bit_exact(out, want, "%s tiled %dx%d tiles" % (op, plan["tiles_x"], plan["tiles_y"]))
    if pointwise or frac >= 1.0:
        whole = np.zeros(oshape, np.float32)
        assert ck.call(o, "oracle_" + op, piece, data, src, whole) == 0
        hc.assert_bit_exact(out, whole, op + " tiled vs untiled")
    assert np.isfinite(out).all() and out.std() > 0


def test_modules_that_move_pixels_are_refused():
    l = hc.hip()
    piece = abi.Piece.make(64, 64, channels=4)
    piece.roi_out.width = 32
    d = abi.ExposureData(0.0, 1.0)
    t = abi.Tiling()
    l.dt_hip_default_tiling(C.byref(piece), 0, C.byref(t))
    buf = np.zeros((64, 64, 4), np.float32)
    rc = l.dt_hip_default_process_tiling_ptp(0, b"exposure", C.byref(piece), C.cast(C.byref(d), C.c_void_p), C.sizeof(d), C.byref(t),
                                             buf.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), 16, 16, 0)
    assert rc == abi.DT_HIP_INVALID_ARG and b"roi_in != roi_out" in l.dt_hip_last_error()


# ---- roi_in != roi_out: _default_process_tiling_cl_roi() (src/develop/tiling.c:1076-1390), finalscale ----------------
@pytest.mark.parametrize("interp", [0, 2])
@pytest.mark.parametrize("iw,ih,scale,frac", [(1500, 1000, 0.5, 0.3), (1201, 803, 0.37, 0.15), (900, 1300, 0.25, 0.2),
                                              (800, 600, 0.61, 4.0)])
def test_roi_host_tiling_of_finalscale_equals_the_oracle_over_the_same_tiles(iw, ih, scale, frac, interp):
    """every tile is resampled as an image of its own, like the reference's tiles (finalscale ignores the region
    origins): the checker is the oracle run on each tile's input region, its good part pasted into the frame"""
    import tile_plan_roi as tr
    l = hc.hip()
    o = ck.oracle()
    roi_out = tr.R(0, 0, int(iw * scale + 0.5), int(ih * scale + 0.5), scale)
    roi_in = tr.modify_roi_in(roi_out)
    roi_in.width, roi_in.height = min(roi_in.width, iw), min(roi_in.height, ih)
    iw, ih = roi_in.width, roi_in.height
    ow, oh = roi_out.width, roi_out.height
    img = synth.rgba_image(iw, ih, seed=33, lo=-0.05, hi=1.3)
    img[..., 3] = 0.25
    piece = abi.Piece.make(ow, oh, roi_in=abi.Roi.make(0, 0, iw, ih, 1.0), roi_out=abi.Roi.make(0, 0, ow, oh, scale))
    d = abi.FinalscaleData(interp)
    t = abi.Tiling()
    l.dt_hip_default_tiling(C.byref(piece), 0, C.byref(t))
    t.overlap = 4
    avail = int(iw * ih * 16 * max(t.factor_cl, 1.0) * frac)
    out = np.zeros((oh, ow, 4), np.float32)
    rc = l.dt_hip_default_process_tiling_roi(0, b"finalscale", C.byref(piece), C.cast(C.byref(d), C.c_void_p), C.sizeof(d),
                                             C.byref(t), img.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 16, 16, avail)
    lib.check(rc, "dt_hip_default_process_tiling_roi")
    p = tr.plan(roi_in, roi_out, 16, 16, t, 0, avail, l.dt_hip_get_device_memalloc(0), 1 << 30, 1 << 30)
    if frac < 1.0:
        assert p["tiles_x"] * p["tiles_y"] > 1
    want = np.zeros_like(out)
    for tx in range(p["tiles_x"]):
        for ty in range(p["tiles_y"]):
            rois = tr.tile_rois(p, roi_in, roi_out, tx, ty)
            if rois is None:
                continue
            a, b, g = rois
            tp_ = abi.Piece.make(b.width, b.height, roi_in=abi.Roi.make(*a.tup()), roi_out=abi.Roi.make(*b.tup()))
            tile_in = np.ascontiguousarray(img[a.y:a.y + a.height, a.x:a.x + a.width])
            tile_out = np.zeros((b.height, b.width, 4), np.float32)
            assert ck.call(o, "oracle_finalscale", tp_, d, tile_in, tile_out) == 0
            want[g.y:g.y + g.height, g.x:g.x + g.width] = tile_out[g.y - b.y:g.y - b.y + g.height, g.x - b.x:g.x - b.x + g.width]
    assert int((ck.ulp_diff(out, want) > 0).sum()) == 0
    # one tile = the untiled module
    if p["tiles_x"] * p["tiles_y"] == 1:
        whole = np.zeros_like(out)
        assert ck.call(o, "oracle_finalscale", piece, d, img, whole) == 0
        assert np.array_equal(out, whole)


def test_roi_host_tiling_refuses_other_modules():
    l = hc.hip()
    piece = abi.Piece.make(64, 64)
    d = abi.ExposureData(0.0, 1.0)
    t = abi.Tiling()
    l.dt_hip_default_tiling(C.byref(piece), 0, C.byref(t))
    buf = np.zeros((64, 64, 4), np.float32)
    rc = l.dt_hip_default_process_tiling_roi(0, b"exposure", C.byref(piece), C.cast(C.byref(d), C.c_void_p), C.sizeof(d), C.byref(t),
                                             buf.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), 16, 16, 0)
    assert rc == abi.DT_HIP_INVALID_ARG and b"modify_roi_in" in l.dt_hip_last_error()
