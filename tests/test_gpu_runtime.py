"""-m gpu: the device runtime of include/ansel_hip.h section 1 -- the peers of the dt_opencl_* calls the host files
on this path make (src/develop/pixelpipe_gpu.c, tiling.c, src/caches/pixelpipe_cache.c, blend.c), with the return
conventions of src/common/opencl.h."""
import ctypes as C
import threading

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import abi, lib

pytestmark = pytest.mark.gpu


def test_reserve_and_release_follow_the_reference_conventions():
    # opencl.h:351-419: _for_pipe -> devid or -1; try_ -> 0 when reserved, never waits; _by_id blocks
    l = hc.hip()
    n = l.dt_hip_get_num_devices()
    assert n >= 1
    got = [l.dt_hip_reserve_device_for_pipe(0) for _ in range(n)]
    assert sorted(got) == list(range(n))
    assert l.dt_hip_reserve_device_for_pipe(0) == -1          # every device busy
    assert l.dt_hip_try_reserve_device_by_id(0) != 0           # busy: non-zero, did not wait
    assert l.dt_hip_try_reserve_device_by_id(n) != 0           # out of range
    # a blocked _by_id gets the device when it is released
    acquired = threading.Event()

    def waiter():
        l.dt_hip_reserve_device_by_id(0)
        acquired.set()
        l.dt_hip_release_device(0)

    t = threading.Thread(target=waiter)
    t.start()
    assert not acquired.wait(0.2)
    l.dt_hip_release_device(0)
    assert acquired.wait(5.0)
    t.join()
    for d in range(1, n):
        l.dt_hip_release_device(d)
    assert l.dt_hip_try_reserve_device_by_id(0) == 0
    l.dt_hip_release_device(0)
    l.dt_hip_reserve_device_by_id(-3)  # out-of-range ids are ignored
    l.dt_hip_release_device(99)


def test_device_queries():
    l = hc.hip()
    w, h = C.c_int(0), C.c_int(0)
    assert l.dt_hip_get_device_max_image_size(0, C.byref(w), C.byref(h)) == 1
    assert w.value >= 65536 and h.value >= 65536
    assert l.dt_hip_get_device_max_image_size(7, C.byref(w), C.byref(h)) == 0
    total = l.dt_hip_get_device_max_global_mem(0)
    assert total >= 250 * 2 ** 30                                   # 288 GB of HBM3E
    assert l.dt_hip_get_device_available(0) <= total
    assert l.dt_hip_get_device_max_global_mem(9) == 0
    assert l.dt_hip_use_pinned_memory(0) == 1 and l.dt_hip_avoid_atomics(0) == 0 and l.dt_hip_micro_nap(0) == 0
    l.dt_hip_check_tuning(0)
    assert l.dt_hip_dev_roundup_width(131, 0) == 131 and l.dt_hip_dev_roundup_height(77, 0) == 77  # linear: nothing rounded
    assert l.dt_hip_is_enabled() == 1 and l.dt_hip_update_settings() == 1
    assert l.dt_hip_enqueue_barrier(0) == abi.DT_HIP_SUCCESS and l.dt_hip_enqueue_barrier(5) != abi.DT_HIP_SUCCESS


def test_fits_device_reason():
    l = hc.hip()
    need, lim = C.c_size_t(0), C.c_size_t(0)
    assert l.dt_hip_image_fits_device_reason(0, 11648, 8736, 16, 5.0, 0, C.byref(need), C.byref(lim)) == 0
    assert need.value == 11648 * 8736 * 16 * 5 and lim.value == l.dt_hip_get_device_available(0)
    assert l.dt_hip_image_fits_device(0, 11648, 8736, 16, 5.0, 0) == 1
    # 10^6 x 10^6 float4: one buffer is already larger than anything allocatable
    assert l.dt_hip_image_fits_device_reason(0, 10 ** 6, 10 ** 6, 16, 2.0, 0, C.byref(need), C.byref(lim)) == 1
    assert lim.value == l.dt_hip_get_device_memalloc(0)
    # each buffer fits, the module's total does not
    side = 60000
    assert l.dt_hip_image_fits_device_reason(0, side, side, 16, 8.0, 0, None, None) == 2
    assert l.dt_hip_image_fits_device(0, side, side, 16, 8.0, 0) == 0


def test_image_geometry_is_remembered():
    l = hc.hip()
    m = l.dt_hip_alloc_device(0, 640, 480, 16)
    assert (l.dt_hip_get_image_width(m), l.dt_hip_get_image_height(m), l.dt_hip_get_image_element_size(m)) == (640, 480, 16)
    assert l.dt_hip_get_mem_context_id(m) == 0 and l.dt_hip_get_mem_object_size(m) >= 640 * 480 * 16
    b = l.dt_hip_alloc_device_buffer(0, 4096)
    assert (l.dt_hip_get_image_width(b), l.dt_hip_get_image_height(b), l.dt_hip_get_image_element_size(b)) == (0, 0, 0)
    l.dt_hip_release_mem_object(m)
    l.dt_hip_release_mem_object(b)
    # a pooled block handed out again as a plain buffer does not keep the old geometry
    b2 = l.dt_hip_alloc_device_buffer(0, 640 * 480 * 16)
    assert l.dt_hip_get_image_width(b2) == 0
    l.dt_hip_release_mem_object(b2)
    assert l.dt_hip_get_mem_context_id(C.c_void_p(12345)) == -1


def test_copy_host_to_device_variants():
    l = hc.hip()
    rng = np.random.default_rng(5)
    a = rng.standard_normal((37, 53, 4)).astype(np.float32)
    m = l.dt_hip_copy_host_to_device(0, a.ctypes.data_as(C.c_void_p), 53, 37, 16)
    assert m and l.dt_hip_get_image_width(m) == 53
    back = np.empty_like(a)
    assert l.dt_hip_copy_device_to_host(0, back.ctypes.data_as(C.c_void_p), m, 53, 37, 16) == abi.DT_HIP_SUCCESS
    assert back.tobytes() == a.tobytes()
    l.dt_hip_release_mem_object(m)
    # a host image with padded rows
    padded = np.zeros((37, 64, 4), np.float32)
    padded[:, :53] = a
    m = l.dt_hip_copy_host_to_device_rowpitch(0, padded.ctypes.data_as(C.c_void_p), 53, 37, 16, 64 * 16)
    assert m
    assert l.dt_hip_copy_device_to_host(0, back.ctypes.data_as(C.c_void_p), m, 53, 37, 16) == abi.DT_HIP_SUCCESS
    assert back.tobytes() == a.tobytes()
    l.dt_hip_release_mem_object(m)
    # constants (lookup tables): a plain buffer, then partial reads / writes at an offset
    lut = np.arange(4096, dtype=np.float32)
    c = l.dt_hip_copy_host_to_device_constant(0, lut.nbytes, lut.ctypes.data_as(C.c_void_p))
    assert c
    part = np.empty(100, np.float32)
    assert l.dt_hip_read_buffer_from_device(0, part.ctypes.data_as(C.c_void_p), c, 400, 400, 1) == abi.DT_HIP_SUCCESS
    assert np.array_equal(part, lut[100:200])
    patch = np.full(10, -1.0, np.float32)
    assert l.dt_hip_write_buffer_to_device(0, patch.ctypes.data_as(C.c_void_p), c, 40, 40, 1) == abi.DT_HIP_SUCCESS
    assert l.dt_hip_read_buffer_from_device(0, part.ctypes.data_as(C.c_void_p), c, 0, 400, 1) == abi.DT_HIP_SUCCESS
    want = lut[:100].copy()
    want[10:20] = -1.0
    assert np.array_equal(part, want)
    l.dt_hip_release_mem_object(c)
    assert not l.dt_hip_copy_host_to_device(0, a.ctypes.data_as(C.c_void_p), 0, 37, 16)


def test_host_pointer_objects_map_and_compute():
    """zero copy: a module reads pinned host memory through its device view, and the host maps the result"""
    l = hc.hip()
    w, h = 96, 40
    nbytes = w * h * 16
    pin_in, pin_out = l.dt_hip_alloc_host_pinned(nbytes), l.dt_hip_alloc_host_pinned(nbytes)
    assert pin_in and pin_out
    a = np.random.default_rng(2).random((h, w, 4)).astype(np.float32)
    C.memmove(pin_in, a.ctypes.data, nbytes)
    d_in = l.dt_hip_alloc_device_use_host_pointer(0, w, h, 16, pin_in, 0)
    d_out = l.dt_hip_alloc_device_use_host_pointer(0, w, h, 16, pin_out, 0)
    assert d_in and d_out and l.dt_hip_get_image_height(d_in) == h
    piece = abi.Piece.make(w, h)
    d = abi.ExposureData(0.01, 1.5)
    lib.check(l.dt_hip_iop_exposure_process(0, C.byref(piece), C.byref(d), d_in, d_out), "exposure")
    mapped = l.dt_hip_map_image(0, d_out, 1, 0, w, h, 16)      # blocking: drains the stream first
    assert mapped == pin_out
    got = np.frombuffer((C.c_char * nbytes).from_address(mapped), np.float32).reshape(h, w, 4)
    want = (a - np.float32(0.01)) * np.float32(1.5)
    assert got.tobytes() == want.tobytes()
    assert l.dt_hip_unmap_mem_object(0, d_out, mapped) == abi.DT_HIP_SUCCESS
    assert l.dt_hip_map_buffer(0, d_out, 1, 0, 64, 16) == pin_out + 64
    # device-only memory has no host mapping; pageable host memory has no device view
    dev = l.dt_hip_alloc_device(0, w, h, 16)
    assert not l.dt_hip_map_image(0, dev, 1, 0, w, h, 16)
    assert l.dt_hip_unmap_mem_object(0, dev, None) != abi.DT_HIP_SUCCESS
    assert not l.dt_hip_alloc_device_use_host_pointer(0, w, h, 16, a.ctypes.data_as(C.c_void_p), 0)
    for m in (d_in, d_out, dev):
        l.dt_hip_release_mem_object(m)
    # releasing the views left the host memory with its owner
    assert l.dt_hip_is_pinned_memory(pin_in) == 1
    l.dt_hip_free_host_pinned(pin_in)
    l.dt_hip_free_host_pinned(pin_out)


def test_events_wait_and_flush():
    l = hc.hip()
    l.dt_hip_events_reset(0)
    l.dt_hip_events_enable(0, 1)
    piece = abi.Piece.make(256, 256)
    d = abi.ExposureData(0.0, 2.0)
    buf = lib.DeviceBuffer(0, 256 * 256 * 16)
    for _ in range(3):
        lib.check(l.dt_hip_iop_exposure_process(0, C.byref(piece), C.byref(d), buf.ptr, buf.ptr), "exposure")
    l.dt_hip_events_wait_for(0)
    tags, ms, cnt = (C.c_char_p * 8)(), (C.c_float * 8)(), (C.c_int * 8)()
    assert l.dt_hip_events_profiling(0, tags, ms, cnt, 8) == 1 and cnt[0] == 3 and tags[0] == b"exposure"
    assert l.dt_hip_events_flush(0, 1) == abi.DT_HIP_SUCCESS                  # reset: the records are gone
    assert l.dt_hip_events_profiling(0, tags, ms, cnt, 8) == 0
    l.dt_hip_events_enable(0, 0)
    assert l.dt_hip_events_flush(3, 0) != abi.DT_HIP_SUCCESS


def test_report_pipe_error_gives_up_on_the_fifth():
    """last: it switches the device path off for the process (opencl.c:1790-1802, DT_OPENCL_MAX_ERRORS 5)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, 'tests'); import hipcheck as hc; l = hc.hip();"
            "r = [l.dt_hip_report_pipe_error() for _ in range(6)]; print(r, l.dt_hip_is_enabled(), l.dt_hip_update_settings())")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=hc.__file__.rsplit("/tests/", 1)[0])
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().endswith("[1, 1, 1, 1, 2, 2] 0 0"), out.stdout
