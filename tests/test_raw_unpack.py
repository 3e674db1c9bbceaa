"""The checker of dt_hip_raw_unpack() (oracle_raw_unpack, oracle/src/basic.c) against numpy's own bit unpacking.  The
reference has no code for this step in its tree (rawspeed, an absent submodule, unpacks sensor data on the CPU), so the
two layouts are pinned here from their definitions: a most-significant-bit-first stream (TIFF / DNG FillOrder 1) and a
little-endian one (the vendor containers' 10 / 12 / 14-bit packing)."""
import ctypes as C

import numpy as np
import pytest

import checkers as ck


def pack(values, bits, order, row_bytes):
    """values [h, w] -> bytes [h, row_bytes] by numpy's packbits"""
    h, w = values.shape
    shifts = np.arange(bits - 1, -1, -1) if order == 0 else np.arange(bits)
    stream = ((values[..., None].astype(np.uint32) >> shifts) & 1).astype(np.uint8).reshape(h, w * bits)
    rows = np.packbits(stream, axis=1, bitorder="big" if order == 0 else "little")
    out = np.full((h, row_bytes), 0xA5, np.uint8)  # padding bytes that must not matter
    out[:, :rows.shape[1]] = rows
    return out


def oracle_unpack(packed, w, h, row_bytes, bits, order):
    o = ck.oracle()
    o.oracle_raw_unpack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    o.oracle_raw_unpack.restype = C.c_int
    out = np.zeros((h, w), np.uint16)
    assert o.oracle_raw_unpack(packed.ctypes.data, w, h, row_bytes, bits, order, out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("bits", [8, 10, 12, 14, 16])
@pytest.mark.parametrize("w,h,pad", [(64, 8, 0), (131, 7, 5), (1, 3, 2), (7, 1, 0), (4000, 3, 16)])
def test_oracle_unpack_equals_numpy(bits, order, w, h, pad):
    if ck.oracle() is None:
        pytest.skip("oracle/liboracle.so not built")
    rng = np.random.default_rng(bits * 7 + order + w)
    values = rng.integers(0, 1 << bits, size=(h, w)).astype(np.uint16)
    values[0, 0] = (1 << bits) - 1
    row_bytes = (w * bits + 7) // 8 + pad
    packed = np.ascontiguousarray(pack(values, bits, order, row_bytes))
    assert np.array_equal(oracle_unpack(packed, w, h, row_bytes, bits, order), values)


def test_the_twelve_bit_layouts_byte_by_byte():
    """the two layouts as the header states them"""
    if ck.oracle() is None:
        pytest.skip("oracle/liboracle.so not built")
    p0, p1 = 0xABC, 0x123
    msb = np.array([[0xAB, 0xC1, 0x23]], np.uint8)
    lsb = np.array([[p0 & 0xFF, (p0 >> 8) | ((p1 & 0xF) << 4), p1 >> 4]], np.uint8)
    assert oracle_unpack(msb, 2, 1, 3, 12, 0).tolist() == [[p0, p1]]
    assert oracle_unpack(lsb, 2, 1, 3, 12, 1).tolist() == [[p0, p1]]
