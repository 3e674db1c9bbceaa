"""-m gpu: a stream of frames through one pipe (dt_hip_batch_*): upload, kernels and download of consecutive frames
overlap on three streams; every frame must come back byte for byte what the serial path gives."""
import ctypes as C

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import filmic, lib, params, pipe, synth

pytestmark = pytest.mark.gpu


def _pinned(l, nbytes):
    p = l.dt_hip_alloc_host_pinned(nbytes)
    assert p and l.dt_hip_is_pinned_memory(p) == 1
    return p


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_batch_of_frames_equals_serial(depth):
    l = hc.hip()
    w, h, nframes = 1504, 1000, 7
    lut = params.srgb_encode_lut()
    d_lut = lib.DeviceBuffer.from_numpy(0, lut)
    nodes = pipe.light_pipe_nodes(w, h, d_lut.ptr, float(lut[0]), params.unbounded_coeffs(lut), with_filmic=True,
                                  filmic=filmic.default_data())
    p = pipe.DevicePipe(0, nodes, fusion=True)
    frames = [synth.bayer_mosaic(w, h, seed=10 + k) for k in range(nframes)]
    # serial reference
    want = []
    din, dout = lib.DeviceBuffer(0, w * h * 2), lib.DeviceBuffer(0, w * h * 8)
    for f in frames:
        din.upload(f)
        p.process(din.ptr, dout.ptr)
        assert l.dt_hip_finish(0) == 1
        want.append(dout.to_numpy((h, w, 4), np.uint16))
    # the batch: one pinned input and output per frame
    nb_in, nb_out = w * h * 2, w * h * 8
    pin_in = [_pinned(l, nb_in) for _ in frames]
    pin_out = [_pinned(l, nb_out) for _ in frames]
    for f, pi in zip(frames, pin_in):
        C.memmove(pi, f.ctypes.data, nb_in)
    b = l.dt_hip_batch_new(p.handle, depth, nb_in, nb_out)
    assert b
    slots = []
    for pi, po in zip(pin_in, pin_out):
        s = l.dt_hip_batch_submit(b, pi, po)
        assert 0 <= s < depth, l.dt_hip_last_error()
        slots.append(s)
    assert slots == [k % depth for k in range(nframes)]
    assert l.dt_hip_batch_drain(b) == 0
    for k, po in enumerate(pin_out):
        got = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint16)), shape=(h, w, 4))
        assert np.array_equal(got, want[k]), "frame %d" % k
    l.dt_hip_batch_free(b)
    for q in pin_in + pin_out:
        l.dt_hip_free_host_pinned(q)
    p.close()


def test_batch_rejects_bad_arguments():
    l = hc.hip()
    assert not l.dt_hip_batch_new(None, 2, 16, 16)
    assert l.dt_hip_batch_submit(None, None, None) == -997
