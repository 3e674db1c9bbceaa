"""-m gpu: a stream of frames through one pipe (dt_hip_batch_*): upload, kernels and download of consecutive frames
overlap on three streams; every frame must come back byte for byte what the serial path gives.  With a writer
(dt_hip_batch_set_writer: the format's write_image() of imageio_core.c:965) frame n is encoded on a host thread while the
frames behind it are on the device."""
import ctypes as C
import time

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


WRITER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_size_t)


def _light_pipe(w, h):
    lut = params.srgb_encode_lut()
    d_lut = lib.DeviceBuffer.from_numpy(0, lut)
    nodes = pipe.light_pipe_nodes(w, h, d_lut.ptr, float(lut[0]), params.unbounded_coeffs(lut), with_filmic=True,
                                  filmic=filmic.default_data())
    return pipe.DevicePipe(0, nodes, fusion=True), d_lut


def test_writer_gets_every_frame_in_order_while_later_frames_are_on_the_device():
    l = hc.hip()
    w, h, nframes, depth = 1504, 1000, 7, 3
    p, d_lut = _light_pipe(w, h)
    frames = [synth.bayer_mosaic(w, h, seed=30 + k) for k in range(nframes)]
    want = []
    din, dout = lib.DeviceBuffer(0, w * h * 2), lib.DeviceBuffer(0, w * h * 8)
    for f in frames:
        din.upload(f)
        p.process(din.ptr, dout.ptr)
        assert l.dt_hip_finish(0) == 1
        want.append(dout.to_numpy((h, w, 4), np.uint16))
    nb_in, nb_out = w * h * 2, w * h * 8
    pin_in = [_pinned(l, nb_in) for _ in range(depth)]
    pin_out = [_pinned(l, nb_out) for _ in range(depth)]
    written, t_end, t_submitted = [], {}, {}

    def write_image(user, seq, host_out, nbytes):
        # the "encoder": keeps a copy, and takes its time
        assert nbytes == nb_out and host_out == pin_out[seq % depth]
        written.append((seq, np.ctypeslib.as_array(C.cast(host_out, C.POINTER(C.c_uint16)), shape=(h, w, 4)).copy()))
        time.sleep(0.15)
        t_end[seq] = time.monotonic()
        return 0

    cb = WRITER(write_image)
    b = l.dt_hip_batch_new(p.handle, depth, nb_in, nb_out)
    assert b and l.dt_hip_batch_set_writer(b, cb, None) == 0
    for k, f in enumerate(frames):
        # the host buffers go round with the slots: submit() waits for the slot's writer before it lets them be reused
        if k >= depth:
            assert l.dt_hip_batch_wait(b, k % depth) == 0
        C.memmove(pin_in[k % depth], f.ctypes.data, nb_in)
        assert l.dt_hip_batch_submit(b, pin_in[k % depth], pin_out[k % depth]) == k % depth, l.dt_hip_last_error()
        t_submitted[k] = time.monotonic()
    assert l.dt_hip_batch_drain(b) == 0
    assert [s for s, _ in written] == list(range(nframes))
    for k, (_, got) in enumerate(written):
        assert np.array_equal(got, want[k]), "frame %d" % k
    # frames 1 and 2 were handed to the device while frame 0 was still being written
    assert t_submitted[1] < t_end[0] and t_submitted[2] < t_end[0]
    # without a writer again: the plain three-leg batch
    assert l.dt_hip_batch_set_writer(b, None, None) == 0
    C.memmove(pin_in[0], frames[3].ctypes.data, nb_in)
    s = l.dt_hip_batch_submit(b, pin_in[0], pin_out[0])
    assert s >= 0 and l.dt_hip_batch_wait(b, s) == 0
    got = np.ctypeslib.as_array(C.cast(pin_out[0], C.POINTER(C.c_uint16)), shape=(h, w, 4))
    assert np.array_equal(got, want[3]) and len(written) == nframes
    l.dt_hip_batch_free(b)
    for q in pin_in + pin_out:
        l.dt_hip_free_host_pinned(q)
    p.close()


def test_a_writer_that_fails_is_reported_for_its_frame_only():
    l = hc.hip()
    w, h, depth = 752, 500, 2
    p, d_lut = _light_pipe(w, h)
    nb_in, nb_out = w * h * 2, w * h * 8
    pin_in, pin_out = _pinned(l, nb_in), [_pinned(l, nb_out) for _ in range(depth)]
    f = synth.bayer_mosaic(w, h, seed=3)
    C.memmove(pin_in, f.ctypes.data, nb_in)
    seen = []

    def write_image(user, seq, host_out, nbytes):
        seen.append(seq)
        return 1 if seq == 1 else 0  # "disk full" on the second frame

    cb = WRITER(write_image)
    b = l.dt_hip_batch_new(p.handle, depth, nb_in, nb_out)
    assert b and l.dt_hip_batch_set_writer(b, cb, None) == 0
    assert l.dt_hip_batch_submit(b, pin_in, pin_out[0]) == 0
    assert l.dt_hip_batch_submit(b, pin_in, pin_out[1]) == 1
    assert l.dt_hip_batch_wait(b, 0) == 0
    assert l.dt_hip_batch_wait(b, 1) == -996 and b"writer" in l.dt_hip_last_error()
    # the slot is free again and the stream goes on
    assert l.dt_hip_batch_submit(b, pin_in, pin_out[0]) == 0
    assert l.dt_hip_batch_submit(b, pin_in, pin_out[1]) == 1
    assert l.dt_hip_batch_drain(b) == 0
    assert seen == [0, 1, 2, 3]
    assert l.dt_hip_batch_set_writer(None, None, None) == -997
    l.dt_hip_batch_free(b)
    for q in [pin_in] + pin_out:
        l.dt_hip_free_host_pinned(q)
    p.close()
