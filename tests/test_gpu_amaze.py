"""-m gpu: AMaZE demosaic on the device, bit for bit against the oracle (buffer zeroed per tile), and
against the reference's own code outside oracle_amaze_stale_mask() (the reference keeps its tile buffer
from tile to tile, so a few pixels depend on OpenMP scheduling; tests/test_oracle_vs_ref.py pins the
restatement against the one-thread reference on every pixel)."""
import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
from ansel_amd import abi, synth

pytestmark = pytest.mark.gpu

SIZES = [(300, 200), (517, 389), (401, 333), (160, 160), (130, 97), (273, 273), (64, 64), (47, 53), (1504, 1000)]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("filters", [0x94949494, 0x49494949, 0x61616161, 0x16161616])
def test_amaze(w, h, filters):
    if (w, h) == (1504, 1000) and filters != 0x94949494:
        pytest.skip("CFA phase variants on the small frames")
    raw = synth.bayer_mosaic(w, h, seed=w + h).astype(np.float32)
    cfa = ((raw - 512) / np.float32(synth.WHITE - 512) * np.float32(1.7)).astype(np.float32)
    piece = abi.Piece.make(w, h, filters=filters, channels=1, processed_maximum=(1.5, 1.0, 1.2, 1.0))
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0)
    pre = np.full((h, w, 4), -7.0, np.float32)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, cfa, (h, w, 4), pre_fill=pre)
    want = pre.copy()
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, cfa, want) == 0
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d values differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
    ref = ck.ref()
    if ref is not None:
        r = pre.copy()
        assert ck.call(ref, "ref_demosaic", piece, d, cfa, r) == 0
        mask = np.zeros((h, w), np.uint8)
        ck.oracle().oracle_amaze_stale_mask(ck.ptr(mask), w, h)
        assert int(((ck.ulp_diff(got, r).max(-1) > 0) & (mask == 0)).sum()) == 0


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


@pytest.mark.parametrize("which", ["amaze_unfused", "amaze_slab"])
@pytest.mark.parametrize("filters", [0x94949494, 0x16161616])
def test_amaze_kernel_variants(which, filters, dispatch):
    """the frame's full tiles run on chip (amaze_frame: one launch, the workgroups draw cut tiles and full tiles from a
    queue); the same frame with one launch per kind of tile (the measuring configuration) and with the first kernel for
    every tile must be the same bits"""
    w, h = 1504, 1000
    raw = synth.bayer_mosaic(w, h, seed=33).astype(np.float32)
    cfa = ((raw - 512) / np.float32(synth.WHITE - 512) * np.float32(1.7)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    cfa[100:700, 200:1300] *= (0.55 + 0.45 * ((xx[100:700, 200:1300] + yy[100:700, 200:1300]) & 1)).astype(np.float32)
    piece = abi.Piece.make(w, h, filters=filters, channels=1, processed_maximum=(1.5, 1.0, 1.2, 1.0))
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0)
    pre = np.full((h, w, 4), -7.0, np.float32)
    base = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, cfa, (h, w, 4), pre_fill=pre)
    dispatch(which)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, cfa, (h, w, 4), pre_fill=pre)
    assert np.array_equal(got.view(np.uint32), base.view(np.uint32))
    want = pre.copy()
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, cfa, want) == 0
    assert int((ck.ulp_diff(got, want) > 0).sum()) == 0


def test_amaze_highlights_and_flat_areas():
    """clipped highlights (the > clip_pt branches), a constant plane (all weights 0/0-guarded by eps)"""
    w, h = 450, 330  # (six of its twelve tiles are full ones: on chip)
    raw = synth.bayer_mosaic(w, h, seed=9).astype(np.float32)
    cfa = ((raw - 512) / np.float32(synth.WHITE - 512) * np.float32(3.0)).astype(np.float32)
    cfa[60:120, 80:200] = 1.0
    cfa[150:200, 30:90] = 0.0
    cfa = np.minimum(cfa, 1.0).astype(np.float32)
    piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=(1.0, 1.0, 1.0, 1.0))
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0)
    pre = np.zeros((h, w, 4), np.float32)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, cfa, (h, w, 4), pre_fill=pre)
    want = pre.copy()
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, cfa, want) == 0
    assert int((ck.ulp_diff(got, want) > 0).sum()) == 0


@pytest.mark.parametrize("filters", [0x94949494, 0x49494949, 0x61616161, 0x16161616])
def test_amaze_many_tiles_per_workgroup(filters, dispatch):
    """three workgroups walk a frame of 96 tiles (a 24 MP frame has 1 500 tiles for 256 workgroups), cut tiles and full ones
    as the queue hands them out: the tile buffer and the LDS rings are zeroed between tiles, the frame is the oracle's.  Fine checkerboards and stripes switch the Nyquist branches on."""
    w, h = 1504, 1000
    raw = synth.bayer_mosaic(w, h, seed=21).astype(np.float32)
    cfa = ((raw - 512) / np.float32(synth.WHITE - 512) * np.float32(1.7)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    cfa[100:420, 200:900] *= (0.55 + 0.45 * ((xx[100:420, 200:900] + yy[100:420, 200:900]) & 1)).astype(np.float32)
    cfa[500:900, 300:1300] *= (0.6 + 0.4 * ((xx[500:900, 300:1300] >> 1) & 1)).astype(np.float32)
    cfa[300:700, 1000:1450] *= (0.6 + 0.4 * (yy[300:700, 1000:1450] & 1)).astype(np.float32)
    piece = abi.Piece.make(w, h, filters=filters, channels=1, processed_maximum=(1.5, 1.0, 1.2, 1.0))
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0)
    dispatch("amaze_blocks", 3)
    pre = np.full((h, w, 4), -7.0, np.float32)
    got = hc.run_hip("dt_hip_iop_demosaic_process", piece, d, cfa, (h, w, 4), pre_fill=pre)
    want = pre.copy()
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, cfa, want) == 0
    diff = ck.ulp_diff(got, want)
    assert int((diff > 0).sum()) == 0, "%d values differ, max %d ulp" % (int((diff > 0).sum()), int(diff.max()))
