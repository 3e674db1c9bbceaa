"""-m gpu: dt_hip_raw_unpack() -- packed sensor data (10 / 12 / 14 bits per photosite, either bit order) to the u16 mosaic on
the device, against the oracle's bit-by-bit unpacker; and in front of the light pipe."""
import ctypes as C

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import abi, lib
from test_raw_unpack import oracle_unpack, pack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order", [abi.RAW_PACK_MSB, abi.RAW_PACK_LSB])
@pytest.mark.parametrize("bits", [8, 10, 12, 14, 16])
@pytest.mark.parametrize("w,h,pad", [(64, 8, 0), (131, 7, 5), (1, 3, 2), (7, 1, 0), (6000, 400, 3)])
def test_raw_unpack(bits, order, w, h, pad):
    l = hc.hip()
    rng = np.random.default_rng(bits * 11 + order + w)
    values = rng.integers(0, 1 << bits, size=(h, w)).astype(np.uint16)
    row_bytes = (w * bits + 7) // 8 + pad
    packed = np.ascontiguousarray(pack(values, bits, order, row_bytes))
    d_in, d_out = lib.DeviceBuffer.from_numpy(0, packed), lib.DeviceBuffer(0, w * h * 2)
    lib.check(l.dt_hip_raw_unpack(0, d_in.ptr, w, h, row_bytes, bits, order, d_out.ptr), "raw_unpack")
    assert l.dt_hip_finish(0) == 1
    got = d_out.to_numpy((h, w), np.uint16)
    assert np.array_equal(got, oracle_unpack(packed, w, h, row_bytes, bits, order))
    assert np.array_equal(got, values)
    d_in.release()
    d_out.release()


def test_raw_unpack_refuses_what_it_does_not_unpack():
    l = hc.hip()
    buf = lib.DeviceBuffer(0, 4096)
    assert l.dt_hip_raw_unpack(0, buf.ptr, 16, 4, 24, 11, 0, buf.ptr) == abi.DT_HIP_INVALID_ARG   # 11 bits
    assert l.dt_hip_raw_unpack(0, buf.ptr, 16, 4, 24, 12, 2, buf.ptr) == abi.DT_HIP_INVALID_ARG   # unknown order
    assert l.dt_hip_raw_unpack(0, buf.ptr, 16, 4, 23, 12, 0, buf.ptr) == abi.DT_HIP_INVALID_ARG   # rows too short
    assert b"cannot hold" in l.dt_hip_last_error()
    buf.release()
