"""-m gpu: RCD and PPG demosaic on the GPU vs the CPU checkers, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
from ansel_amd import abi, synth

pytestmark = pytest.mark.gpu


def _normalised_cfa(w, h, seed=3):
    cfa = synth.bayer_mosaic(w, h, seed=seed).astype(np.float32)
    out = (cfa - 512.0) / np.float32(synth.WHITE - 512)
    # white balance, as temperature does upstream
    rows = np.arange(h)[:, None]
    cols = np.arange(w)[None, :]
    wb = np.asarray(synth.WB_COEFFS, dtype=np.float32)[synth.fc(rows, cols)]
    return (out * wb).astype(np.float32)


def _stale_mask(w, h, filters):
    m = np.zeros((h, w), np.uint8)
    ck.oracle().oracle_rcd_stale_mask(ck.ptr(m), w, h, C.c_uint32(filters))
    return m


# single tile, partial tile, multi-tile with even and odd last-tile widths, non-RGGB phases
RCD_SIZES = [(112, 112), (100, 90), (300, 200), (207, 131), (512, 384), (1502, 1002), (2000, 1300)]


@pytest.mark.parametrize("w,h", RCD_SIZES)
@pytest.mark.parametrize("roi_xy", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_rcd(w, h, roi_xy):
    if roi_xy != (0, 0) and (w, h) not in [(300, 200), (512, 384)]:
        pytest.skip("CFA phase variants on two sizes only")
    img = _normalised_cfa(w, h)
    pm = synth.WB_COEFFS
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=pm,
                           roi_in=abi.Roi.make(*roi_xy, w, h), roi_out=abi.Roi.make(*roi_xy, w, h))
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD, 0.0)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4))
    exp = hc.run_cpu("oracle", "demosaic", piece, d, img, (h, w, 4))
    hc.assert_bit_exact(got, exp, "rcd vs oracle")
    ref = hc.run_cpu("ref", "demosaic", piece, d, img, (h, w, 4))
    if ref is not None:
        # the reference reads stale per-thread scratch on <= 3 columns of a partial last tile
        # column (oracle/src/demosaic_rcd.c header); everywhere else it must agree exactly
        filters = hc.hip().dt_hip_crop_dcraw_filters(synth.FILTERS_RGGB, roi_xy[0], roi_xy[1])
        mask = _stale_mask(w, h, filters)
        hc.assert_bit_exact(got, ref, "rcd vs reference", mask=mask[..., None])


@pytest.mark.parametrize("w,h", [(300, 200), (1502, 1002), (65, 40)])
def test_ppg(w, h):
    img = _normalised_cfa(w, h)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_PPG, 0.0)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4))
    for which in hc.checkers_available():
        hc.assert_bit_exact(got, hc.run_cpu(which, "demosaic", piece, d, img, (h, w, 4)), "ppg vs " + which)


def test_rcd_negative_and_clipped_input():
    w, h = 300, 200
    img = _normalised_cfa(w, h, seed=8)
    img[50:60, 50:80] = -0.05
    img[100:110, :] = 3.0
    img[150, 10:20] = np.nan
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD, 0.0)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4))
    hc.assert_bit_exact(got, hc.run_cpu("oracle", "demosaic", piece, d, img, (h, w, 4)), "rcd edge input vs oracle")


def test_demosaic_rejects_unsupported():
    from ansel_amd import lib
    h_ = hc.hip()
    piece = abi.Piece.make(64, 64, filters=9, channels=1)
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD, 0.0)
    buf = lib.DeviceBuffer(0, 64 * 64 * 16)
    assert h_.dt_hip_iop_demosaic_process(0, C.byref(piece), C.byref(d), buf.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG
    d2 = abi.DemosaicData(0, 0, 6, 0.0)  # LMMSE: not on the device
    piece2 = abi.Piece.make(64, 64, filters=synth.FILTERS_RGGB, channels=1)
    assert h_.dt_hip_iop_demosaic_process(0, C.byref(piece2), C.byref(d2), buf.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG


EXTRAS = [
    # (method, green_eq, colour smoothing passes, PPG median threshold, green_eq threshold)
    (abi.DT_HIP_DEMOSAIC_PPG, 0, 0, 0.05, 0.0),
    (abi.DT_HIP_DEMOSAIC_PPG, 0, 0, 1.0, 0.0),
    (abi.DT_HIP_DEMOSAIC_PPG, 1, 2, 0.02, 0.08),
    (abi.DT_HIP_DEMOSAIC_PPG, 0, 5, 0.0, 0.0),
    (abi.DT_HIP_DEMOSAIC_RCD, 1, 0, 0.0, 0.64),
    (abi.DT_HIP_DEMOSAIC_RCD, 1, 3, 0.0, 0.01),
    (abi.DT_HIP_DEMOSAIC_AMAZE, 1, 1, 0.0, 0.32),
]


@pytest.mark.parametrize("w,h,xy", [(640, 400, (0, 0)), (207, 131, (1, 1)), (1504, 1000, (1, 0)), (120, 96, (0, 1))])
@pytest.mark.parametrize("method,geq,smooth,median,geq_thr", EXTRAS)
def test_demosaic_optional_steps(w, h, xy, method, geq, smooth, median, geq_thr):
    """green equilibration (local average), PPG's median pre-filter, colour smoothing (demosaic.c:1137-1250,
    demosaic/basic.c:136-293) around each interpolation, for every CFA phase of the roi origin: HIP == oracle"""
    import ctypes as C
    import numpy as np
    import checkers as ck
    import hipcheck as hc
    from ansel_amd import synth
    rng = np.random.default_rng(w + 3 * h + 7 * method)
    cfa = synth.bayer_mosaic(w, h, seed=5).astype(np.float32)
    img = ((cfa - 512.0) / np.float32(synth.WHITE - 512)).astype(np.float32)
    if method != abi.DT_HIP_DEMOSAIC_AMAZE:  # AMaZE with non-finite samples is not reproduced
        img[rng.integers(4, h - 4, 6), rng.integers(4, w - 4, 6)] = [0.0, -0.01, 2.0, np.nan, np.inf, 1e-30]
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS,
                           roi_in=abi.Roi.make(xy[0], xy[1], w, h), roi_out=abi.Roi.make(xy[0], xy[1], w, h))
    d = abi.DemosaicData(geq, smooth, method, median, geq_thr)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4), pre_fill=np.zeros((h, w, 4), np.float32))
    want = np.zeros((h, w, 4), np.float32)
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, img, want) == 0
    hc.assert_bit_exact(got, want, "demosaic method %d geq %d smooth %d median %g" % (method, geq, smooth, median))
    plain = hc.run_hip("dt_hip_iop_demosaic_process", piece, abi.DemosaicData(0, 0, method, 0.0, 0.0), img, (h, w, 4),
                       pre_fill=np.zeros((h, w, 4), np.float32))
    assert not np.array_equal(np.nan_to_num(got), np.nan_to_num(plain))  # the steps did something


@pytest.mark.parametrize("w,h,xy", [(640, 400, (0, 0)), (207, 131, (1, 1)), (1504, 1000, (1, 0)), (120, 96, (0, 1)), (4001, 3001, (0, 0))])
@pytest.mark.parametrize("geq,method", [(2, abi.DT_HIP_DEMOSAIC_RCD), (3, abi.DT_HIP_DEMOSAIC_PPG), (3, abi.DT_HIP_DEMOSAIC_RCD)])
def test_full_average_green_equilibration(w, h, xy, geq, method):
    """green_equilibration_favg() (demosaic/basic.c:296-329), alone and ahead of the local average.  The two frame-wide
    binary64 sums are order dependent in the reference (OpenMP reduction); the device carries them as double-double
    pairs in a fixed order.  Tolerance: one ulp of binary32 per pixel against the index-order sum of the oracle; on
    these frames no pixel differs."""
    import numpy as np
    import checkers as ck
    import hipcheck as hc
    from ansel_amd import synth
    cfa = synth.bayer_mosaic(w, h, seed=11).astype(np.float32)
    img = ((cfa - 512.0) / np.float32(synth.WHITE - 512)).astype(np.float32)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS,
                           roi_in=abi.Roi.make(xy[0], xy[1], w, h), roi_out=abi.Roi.make(xy[0], xy[1], w, h))
    d = abi.DemosaicData(geq, 0, method, 0.0, 0.08)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4), pre_fill=np.zeros((h, w, 4), np.float32))
    want = np.zeros((h, w, 4), np.float32)
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, img, want) == 0
    ulp = np.abs(got[..., :3].view(np.int32).astype(np.int64) - want[..., :3].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, int(ulp.max())
    assert int((ulp != 0).sum()) == 0, "pixels off by one ulp: %d" % int((ulp != 0).sum())
    plain = hc.run_hip("dt_hip_iop_demosaic_process", piece, abi.DemosaicData(0, 0, method, 0.0, 0.0), img, (h, w, 4),
                       pre_fill=np.zeros((h, w, 4), np.float32))
    assert not np.array_equal(got, plain)


@pytest.mark.parametrize("bad,where", [(float("nan"), (10, 11)), (float("inf"), (10, 11)), (float("inf"), (11, 10)), (-1e9, (10, 11))])
def test_full_average_green_equilibration_with_sums_that_are_not_positive_numbers(bad, where):
    import numpy as np
    import checkers as ck
    import hipcheck as hc
    from ansel_amd import synth
    w, h = 64, 48
    img = np.random.default_rng(0).random((h, w)).astype(np.float32)
    img[where] = bad
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1)
    d = abi.DemosaicData(2, 0, abi.DT_HIP_DEMOSAIC_PPG, 0.0, 0.0)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4), pre_fill=np.zeros((h, w, 4), np.float32))
    want = np.zeros((h, w, 4), np.float32)
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, img, want) == 0
    hc.assert_bit_exact(got, want, "favg with %r" % bad)


def test_states_that_are_not_states_of_the_module_are_refused():
    import ctypes as C
    import hipcheck as hc
    from ansel_amd import lib, synth
    l = hc.hip()
    buf = lib.DeviceBuffer(0, 64 * 48 * 16)
    piece = abi.Piece.make(64, 48, filters=synth.FILTERS_RGGB, channels=1)
    d = abi.DemosaicData(4, 0, abi.DT_HIP_DEMOSAIC_PPG, 0.0, 0.1)
    assert l.dt_hip_iop_demosaic_process(0, C.byref(piece), C.byref(d), buf.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD, 0.3, 0.0)   # a median threshold only exists for PPG
    assert l.dt_hip_iop_demosaic_process(0, C.byref(piece), C.byref(d), buf.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG


@pytest.mark.parametrize("w,h,roi_xy", [(300, 200, (0, 0)), (207, 131, (1, 1)), (1, 1, (0, 0)), (1502, 1002, (0, 1))])
@pytest.mark.parametrize("method,geq,smooth", [(abi.DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME, 0, 0), (abi.DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR, 0, 0),
                                               (abi.DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR, 1, 2)])
def test_passthrough(w, h, roi_xy, method, geq, smooth):
    """the passthrough states of the module (passthrough.c:21-67; demosaic.c:1111-1118): device == oracle == reference, alpha
    left as it was"""
    if smooth and (w < 3 or h < 3):
        pytest.skip("colour smoothing on a frame without an interior")
    img = _normalised_cfa(w, h, seed=8)
    piece = abi.Piece.make(w, h, filters=0x61616161, channels=1, processed_maximum=synth.WB_COEFFS,
                           roi_in=abi.Roi.make(*roi_xy, w, h), roi_out=abi.Roi.make(*roi_xy, w, h))
    d = abi.DemosaicData(geq, smooth, method, 0.0)
    pre = np.full((h, w, 4), -3.0, np.float32)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4), pre_fill=pre)
    exp = pre.copy()
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, img, exp) == 0
    hc.assert_bit_exact(got, exp, "passthrough vs oracle")
    if not smooth:
        assert (got[..., 3] == -3.0).all()
    ref = ck.ref()
    if ref is not None:
        r = pre.copy()
        assert ck.call(ref, "ref_demosaic", piece, d, img, r) == 0
        hc.assert_bit_exact(got, r, "passthrough vs reference")


# ---- VNG4 and the dual demosaic (vng.c:34-221, dual.c:35-110) -------------------------------------------------------------
@pytest.mark.parametrize("w,h,roi_xy", [(300, 200, (0, 0)), (207, 131, (1, 1)), (120, 96, (1, 0)), (64, 40, (0, 1)), (19, 17, (0, 0)),
                                         (1502, 1002, (0, 0))])
def test_vng4(w, h, roi_xy):
    rng = np.random.default_rng(w)
    img = _normalised_cfa(w, h, seed=8)
    img[rng.integers(4, h - 4, 5), rng.integers(4, w - 4, 5)] = [0.0, -0.01, 2.0, np.inf, 1e-30]
    img[h // 2:h // 2 + 6, 3:12] = 0.25  # flat: every gradient zero, the pixel keeps its linear interpolation
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS,
                           roi_in=abi.Roi.make(*roi_xy, w, h), roi_out=abi.Roi.make(*roi_xy, w, h))
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_VNG4, 0.0)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4))
    exp = hc.run_cpu("oracle", "demosaic", piece, d, img, (h, w, 4))
    hc.assert_bit_exact(got, exp, "vng4 vs oracle")
    ref = hc.run_cpu("ref", "demosaic", piece, d, img, (h, w, 4))
    if ref is not None:
        hc.assert_bit_exact(got, ref, "vng4 vs reference")


@pytest.mark.parametrize("w,h,roi_xy", [(300, 200, (0, 0)), (207, 131, (1, 1)), (1502, 1002, (0, 1))])
@pytest.mark.parametrize("base,geq,smooth,thrs", [(abi.DT_HIP_DEMOSAIC_RCD, 0, 0, 0.2), (abi.DT_HIP_DEMOSAIC_RCD, 1, 2, 0.05),
                                                   (abi.DT_HIP_DEMOSAIC_AMAZE, 0, 1, 0.6), (abi.DT_HIP_DEMOSAIC_RCD, 0, 0, 0.0)])
def test_dual_demosaic(w, h, roi_xy, base, geq, smooth, thrs):
    """RCD + VNG4 and AMaZE + VNG4: VNG4 of the un-equilibrated mosaic, two smoothing passes, the blurred sigmoid of the
    high-frequency image's raw detail mask, the blend -- every word equal to the oracle's (which equals the reference's
    dual_demosaic() wherever that is a function of its input, tests/test_oracle_vs_ref.py)"""
    img = _normalised_cfa(w, h, seed=9)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS,
                           roi_in=abi.Roi.make(*roi_xy, w, h), roi_out=abi.Roi.make(*roi_xy, w, h))
    d = abi.DemosaicData(geq, smooth, base | abi.DT_HIP_DEMOSAIC_DUAL, 0.0, 0.04, thrs, (C.c_float * 4)(*synth.WB_COEFFS))
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4))
    exp = hc.run_cpu("oracle", "demosaic", piece, d, img, (h, w, 4))
    hc.assert_bit_exact(got, exp, "dual vs oracle")
    if thrs > 0:
        plain = hc.run_hip("dt_hip_iop_demosaic_process", piece, abi.DemosaicData(geq, smooth, base, 0.0, 0.04), img, (h, w, 4))
        assert not np.array_equal(got, plain)


def test_dual_demosaic_needs_the_white_balance_coefficients():
    """a caller that never filled wb_coeffs (zeros) would get the plain RCD result under the dual method's name: refused"""
    from ansel_amd import lib
    w, h = 300, 200
    img = _normalised_cfa(w, h, seed=9)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD | abi.DT_HIP_DEMOSAIC_DUAL, 0.0, 0.04, 0.2)
    with pytest.raises(lib.AnselHipError, match="white-balance"):
        hc.run_hip("dt_hip_iop_demosaic_process", piece, d, img, (h, w, 4))


def test_dual_demosaic_in_a_pipe():
    """the executor takes the dual method as a demosaic node of its own (never fused, never on row bands)"""
    import torch
    from ansel_amd import filmic, params, pipe
    w, h = 640, 480
    lut = params.srgb_encode_lut()
    d_lut = torch.from_numpy(lut).to("cuda:0")
    coeffs = params.unbounded_coeffs(lut)
    raw = synth.bayer_mosaic(w, h, seed=4)

    def nodes(ptr):
        ns = pipe.light_pipe_nodes(w, h, ptr, float(lut[0]), coeffs, with_filmic=True, filmic=filmic.default_data())
        for n in ns:
            if n.op == "demosaic":
                n.data = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD | abi.DT_HIP_DEMOSAIC_DUAL, 0.0, 0.0, 0.2,
                                          (C.c_float * 4)(*synth.WB_COEFFS))
        return ns
    p = pipe.DevicePipe(0, nodes(d_lut.data_ptr()), fusion=True)
    d_in = torch.from_numpy(raw.view(np.int16)).to("cuda:0")
    d_out = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda:0")
    p.process(d_in.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    p.close()
    got = d_out.cpu().numpy().view(np.uint16)
    src = raw
    o = ck.oracle()
    for n in nodes(lut.ctypes.data):
        if n.op == "export_u16":
            exp = np.zeros((h, w, 4), np.uint16)
            o.oracle_export_convert_u16(w, h, ck.ptr(src), ck.ptr(exp))
            break
        dst = np.zeros((h, w) if n.op in ("rawprepare", "temperature", "highlights") else (h, w, 4), np.float32)
        assert ck.call(o, "oracle_" + n.op, n.piece, n.data, np.ascontiguousarray(src), dst) == 0, n.op
        src = dst
    assert np.array_equal(got, exp)
