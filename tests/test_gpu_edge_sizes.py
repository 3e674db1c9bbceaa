"""-m gpu: every module on frames smaller than its tiles, chunks, pyramid levels or grid cells -- one pixel, one
row, one column -- bit for bit against the oracle; sizes on which the reference itself is undefined (it reads
outside its buffers) must be refused, not computed."""
import ctypes as C

import numpy as np
import pytest

import checkers as ck
import edge_cases as ec
import hipcheck as hc
from ansel_amd import lib

pytestmark = pytest.mark.gpu

_FN = {"rgb_to_lab": "dt_hip_transform_rgb_to_lab", "lab_to_rgb": "dt_hip_transform_lab_to_rgb",
       "develop_blend": "dt_hip_develop_blend_process"}


@pytest.fixture(autouse=True, scope="module")
def _few_oracle_threads():
    """the oracle's OpenMP loops over a handful of rows cost seconds per call on a 256-thread host"""
    omp = C.CDLL("libgomp.so.1")
    omp.omp_get_max_threads.restype = C.c_int
    before = omp.omp_get_max_threads()
    omp.omp_set_num_threads(4)
    yield
    omp.omp_set_num_threads(before)


_DEV_LUT = []


def _dev_lut():
    if not _DEV_LUT:
        from ansel_amd import params
        hc.hip()
        _DEV_LUT.append(lib.DeviceBuffer.from_numpy(0, params.srgb_encode_lut()))
    return _DEV_LUT[0].ptr


@pytest.mark.parametrize("size", ec.SIZES, ids=["%dx%d" % s for s in ec.SIZES])
@pytest.mark.parametrize("module", ec.MODULES)
def test_tiny_frames(module, size):
    w, h = size
    op, piece, data, inp, shape, pre = ec.case(module, w, h)
    fn = _FN.get(op, "dt_hip_iop_%s_process" % op)
    if ec.undefined_in_reference(module, w, h):
        h_ = hc.hip()
        din, dout = lib.DeviceBuffer.from_numpy(0, inp), lib.DeviceBuffer.from_numpy(0, np.zeros(shape, np.float32))
        assert getattr(h_, fn)(0, C.byref(piece), C.byref(data), din.ptr, dout.ptr) == -997  # DT_HIP_INVALID_ARG
        assert h_.dt_hip_finish(0) == 1
        return
    want = np.zeros(shape, np.float32) if pre is None else pre.copy()
    assert ck.call(ck.oracle(), "oracle_" + op, piece, data, np.ascontiguousarray(inp), want) == 0
    if module == "colorout":  # the device reads its tone curve from device memory
        op, piece, data, inp, shape, pre = ec.case(module, w, h, lut_ptr=_dev_lut())
    got = hc.run_hip(fn, piece, data, inp, shape, pre_fill=pre)
    d = ck.ulp_diff(got, want)
    if module == "demosaic_amaze":
        m = np.zeros((h, w), np.uint8)
        ck.oracle().oracle_amaze_stale_mask(ck.ptr(m), w, h)  # rows / columns whose reference value is not a function of the input
        d = d * (m[..., None] == 0)
    assert int((d > 0).sum()) == 0, "%s %dx%d: %d differ, max %d ulp" % (module, w, h, int((d > 0).sum()), int(d.max()))


@pytest.mark.parametrize("module", ec.STENCIL_MODULES)
def test_stencils_on_adversarial_input(module):
    """NaN, +-Inf, denormals, +-1e30, -0, negatives inside the neighbourhoods of the stencil modules: bit for bit what
    the oracle (and through it the reference) makes of them"""
    op, piece, data, inp, shape, pre = ec.adversarial(module)
    want = np.zeros(shape, np.float32) if pre is None else pre.copy()
    assert ck.call(ck.oracle(), "oracle_" + op, piece, data, np.ascontiguousarray(inp), want) == 0
    got = hc.run_hip(_FN.get(op, "dt_hip_iop_%s_process" % op), piece, data, inp, shape, pre_fill=pre)
    d = ck.ulp_diff(got, want)
    if module == "demosaic_amaze":
        h, w = shape[:2]
        m = np.zeros((h, w), np.uint8)
        ck.oracle().oracle_amaze_stale_mask(ck.ptr(m), w, h)
        d = d * (m[..., None] == 0)
    assert int((d > 0).sum()) == 0, "%s: %d differ, max %d ulp" % (module, int((d > 0).sum()), int(d.max()))
