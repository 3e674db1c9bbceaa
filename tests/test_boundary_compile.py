"""The drop-in boundary, proven the way SURVEY.md section 7 asks for it: the reference's OWN host files of the export path --
src/develop/pixelpipe_gpu.c (the device dispatch of one pipeline node, :191-744) and src/develop/tiling.c (the host
tilers, :842-1403) -- are compiled from where they lie, UNMODIFIED, against include/ansel_opencl_peer.h installed as
`common/opencl.h` (every dt_opencl_*() they call forwards to its dt_hip_*() peer), linked with libansel_hip.so, and run
on the device on a module written as INTEGRATION.md section 2 prescribes (a four-line process_cl() over
dt_hip_iop_exposure_process()).  The application's GUI-laden headers are replaced by tests/native/boundary/shim/ (the
slice of each struct the two files read; every GUI-free reference header is used as it is); the host-side services the
files call (cache lines, logging) are tests/native/boundary/boundary_stubs.c.  Nothing of the reference is copied.

not gpu: the two files compile and link (only where /root/reference exists; the built library travels to the GPU box).
gpu:     pixelpipe_process_on_GPU() and default_process_tiling_cl() produce the oracle's bytes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import checkers as ck
from ansel_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "native", "boundary")
SO = os.path.join(ROOT, "tests", "native", "libboundary.so")
REF = "/root/reference/src"
GLIB = ["-I/opt/conda/include/glib-2.0", "-I/opt/conda/lib/glib-2.0/include", "-I/opt/conda/include"]


def build():
    """compile the reference's two host files against the peer header; returns the command's stderr"""
    cmd = ["gcc", "-std=gnu11", "-O1", "-fPIC", "-shared", "-DHAVE_OPENCL", "-Werror=implicit-function-declaration",
           "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "include"), "-I" + REF] + GLIB + [
           os.path.join(REF, "develop", "pixelpipe_gpu.c"), os.path.join(REF, "develop", "tiling.c"),
           os.path.join(HERE, "boundary_stubs.c"), "-L" + os.path.join(ROOT, "ansel_amd"), "-lansel_hip", "-L/opt/conda/lib",
           "-lglib-2.0", "-Wl,-rpath,$ORIGIN/../../ansel_amd", "-Wl,-rpath,/opt/conda/lib", "-Wl,--no-undefined", "-lm", "-o", SO]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stderr


def test_reference_host_files_compile_unmodified_against_the_peer_header():
    if not os.path.isdir(REF):
        pytest.skip("the reference's sources are not on this machine (the GPU box): the library was built where they are")
    assert os.path.exists(os.path.join(ROOT, "ansel_amd", "libansel_hip.so")), "run __graft_entry__.build() first"
    build()
    syms = subprocess.run(["nm", "-D", "--defined-only", SO], capture_output=True, text=True).stdout
    for name in ("pixelpipe_process_on_GPU", "default_process_tiling_cl", "default_process_tiling", "default_tiling_callback"):
        assert " T " + name in syms, name  # the reference's own functions, now bound to the HIP runtime
    undefined = subprocess.run(["nm", "-D", "--undefined-only", SO], capture_output=True, text=True).stdout
    used = sorted({l.split()[-1] for l in undefined.splitlines() if " dt_hip_" in l})
    # what the two files reach through the peer header
    for name in ("dt_hip_alloc_device", "dt_hip_release_mem_object", "dt_hip_image_fits_device_reason", "dt_hip_finish",
                 "dt_hip_write_host_to_device_raw", "dt_hip_read_host_from_device_raw", "dt_hip_get_device_available",
                 "dt_hip_get_device_memalloc", "dt_hip_get_device_max_image_size", "dt_hip_enqueue_copy_image"):
        assert name in used, (name, used)
    names = [l.split()[-1] for l in undefined.splitlines() if l.split()]
    assert not [n for n in names if n.startswith("dt_opencl_") or n.startswith("cl")]  # no OpenCL, no unbound peer


def test_no_opencl_name_is_left_unbound():
    """every dt_opencl_* identifier the two host files use has a peer in include/ansel_opencl_peer.h"""
    if not os.path.isdir(REF):
        pytest.skip("needs the reference's sources")
    import re
    peer = open(os.path.join(ROOT, "include", "ansel_opencl_peer.h")).read()
    for f in ("pixelpipe_gpu.c", "tiling.c"):
        src = open(os.path.join(REF, "develop", f)).read()
        for name in sorted(set(re.findall(r"\bdt_opencl_[a-z_0-9]+", src))):
            assert re.search(r"\b%s\b" % name, peer), "%s uses %s, which the peer header lacks" % (f, name)


def _lib():
    assert os.path.exists(SO), "tests/native/libboundary.so missing: built by the CPU test where /root/reference exists"
    l = C.CDLL(SO)
    l.boundary_last_message.restype = C.c_char_p
    return l


@pytest.mark.gpu
def test_reference_pixelpipe_process_on_gpu_runs_the_module_on_the_device():
    l = _lib()
    w, h = 1504, 1000
    img = synth.rgba_image(w, h, seed=12, lo=-0.1, hi=1.4)
    out = np.zeros_like(img)
    flow, calls = C.c_int(0), C.c_int(0)
    rc = l.boundary_run_exposure_gpu(ck.ptr(img), ck.ptr(out), w, h, C.c_float(-0.01), C.c_float(1.7), C.byref(flow), C.byref(calls))
    assert rc == 0, (rc, l.boundary_last_message())
    assert calls.value == 1 and l.boundary_cpu_fallbacks() == 0 and l.boundary_unexpected_calls() == 0
    assert flow.value & (1 << 4) and not flow.value & (1 << 3)  # PIXELPIPE_FLOW_PROCESSED_ON_GPU, not _ON_CPU
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_exposure", abi.Piece.make(w, h), abi.ExposureData(-0.01, 1.7), img, want) == 0
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,budget_px", [(1504, 1000, 200_000), (1001, 777, 60_000), (640, 480, 10_000_000)])
def test_reference_default_process_tiling_cl_runs_the_module_tile_by_tile(w, h, budget_px):
    l = _lib()
    img = synth.rgba_image(w, h, seed=13, lo=-0.1, hi=1.4)
    out = np.zeros_like(img)
    calls = C.c_int(0)
    ok = l.boundary_run_exposure_tiled(ck.ptr(img), ck.ptr(out), w, h, C.c_float(0.02), C.c_float(0.8), C.c_size_t(budget_px * 16),
                                       C.byref(calls))
    assert ok == 1, l.boundary_last_message()
    if budget_px < w * h:
        assert calls.value > 1  # the reference's plan cut the frame
    else:
        assert calls.value == 1
    want = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_exposure", abi.Piece.make(w, h), abi.ExposureData(0.02, 0.8), img, want) == 0
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))  # a pointwise module: tiles = the frame
