"""-m gpu: the first node of the pipe, basebuffer (src/iop/basebuffer.c:118-160) -- the region roi_out of the full
sensor buffer lands in the first cacheline; on the device that is the frame's 2-D upload."""
import ctypes as C

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import abi, lib

pytestmark = pytest.mark.gpu


def _expected(full, x, y, w, h, fill):
    """the reference's loop: in_width / in_height clipped to the sensor, rows beyond left as they were"""
    ih, iw = full.shape[:2]
    out = np.full((h, w) + full.shape[2:], fill, full.dtype)
    x0, y0 = max(x, 0), max(y, 0)
    cw, chh = min(w, iw - x0), min(h, ih - y0)
    if cw > 0 and chh > 0:
        out[:chh, :cw] = full[y0:y0 + chh, x0:x0 + cw]
    return out


@pytest.mark.parametrize("dtype,bpp", [(np.uint16, 2), (np.float32, 4)])
@pytest.mark.parametrize("roi", [(0, 0, 301, 203), (16, 8, 200, 150), (100, 60, 201, 143), (250, 180, 120, 90), (-4, -2, 64, 48)])
def test_basebuffer_crop_upload(dtype, bpp, roi):
    iw, ih = 301, 203
    rng = np.random.default_rng(3)
    full = (rng.random((ih, iw)) * 60000).astype(dtype)
    x, y, w, h = roi
    fill = dtype(7)
    h_ = hc.hip()
    dout = lib.DeviceBuffer.from_numpy(0, np.full((h, w), fill, dtype))
    piece = abi.Piece.make(w, h, roi_in=abi.Roi.make(0, 0, iw, ih, 1.0), roi_out=abi.Roi.make(x, y, w, h, 1.0), channels=1)
    rc = h_.dt_hip_iop_basebuffer_process(0, C.byref(piece), iw, ih, bpp, full.ctypes.data_as(C.c_void_p), dout.ptr)
    lib.check(rc, "basebuffer")
    assert h_.dt_hip_finish(0) == 1
    got = dout.to_numpy((h, w), dtype)
    assert np.array_equal(got, _expected(full, x, y, w, h, fill))


def test_basebuffer_rgba_float():
    iw, ih, w, h = 120, 80, 100, 70
    full = np.random.default_rng(4).random((ih, iw, 4), dtype=np.float32)
    h_ = hc.hip()
    dout = lib.DeviceBuffer.from_numpy(0, np.zeros((h, w, 4), np.float32))
    piece = abi.Piece.make(w, h, roi_in=abi.Roi.make(0, 0, iw, ih, 1.0), roi_out=abi.Roi.make(10, 5, w, h, 1.0), channels=4)
    lib.check(h_.dt_hip_iop_basebuffer_process(0, C.byref(piece), iw, ih, 16, full.ctypes.data_as(C.c_void_p), dout.ptr), "basebuffer")
    assert h_.dt_hip_finish(0) == 1
    assert np.array_equal(dout.to_numpy((h, w, 4), np.float32), full[5:75, 10:110])
