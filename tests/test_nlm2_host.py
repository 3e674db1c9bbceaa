"""The gfx950 kernel for interior non-local-means chunks (ansel_amd/csrc/nlm2_body.h, launched as nlm_chunks_v2)
compiled for the HOST -- a workgroup is 1024 OS threads meeting at a barrier where the kernel has __syncthreads()
(tests/native/nlm2_host.cpp) -- against the oracle, bit for bit.  The same source runs on the device; this pins its
schedule (the work items of the term chains, the ring of the row recurrence, every table and window index, the
two-table pipeline) without a GPU.  The -m gpu tests then only have to show that the device executes it the same."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import checkers as ck
from ansel_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "native", "libnlm2_host.so")
SRC = os.path.join(ROOT, "tests", "native", "nlm2_host.cpp")
HDR = os.path.join(ROOT, "ansel_amd", "csrc", "nlm2_body.h")
HDR3 = os.path.join(ROOT, "ansel_amd", "csrc", "nlm3_body.h")
HDRT = os.path.join(ROOT, "ansel_amd", "csrc", "nlm_tail_body.h")


class NlmParams(C.Structure):  # oracle_nlm_params_t, oracle/src/nlmeans_core.h
    _fields_ = [("scattering", C.c_float), ("scale", C.c_float), ("luma", C.c_float), ("chroma", C.c_float),
                ("center_weight", C.c_float), ("sharpness", C.c_float), ("patch_radius", C.c_int),
                ("search_radius", C.c_int), ("norm", C.c_float * 4)]


@pytest.fixture(scope="module")
def host_kernel():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR3), os.path.getmtime(HDRT)):
        subprocess.check_call(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
                               "-I" + os.path.join(ROOT, "ansel_amd", "csrc"), SRC, "-o", SO])
    return C.CDLL(SO)


def _lab(w, h, seed):
    rng = np.random.default_rng(seed)
    rgb = synth.rgba_image(w, h, seed=seed, lo=0.0, hi=1.0)
    lab = np.zeros((h, w, 4), np.float32)
    lab[..., 0] = 100.0 * rgb[..., 1] + rng.normal(0, 1.5, (h, w))
    lab[..., 1] = 80.0 * (rgb[..., 0] - rgb[..., 1]) + rng.normal(0, 2.0, (h, w))
    lab[..., 2] = 80.0 * (rgb[..., 1] - rgb[..., 2]) + rng.normal(0, 2.0, (h, w))
    lab[..., 3] = rng.random((h, w))
    return np.ascontiguousarray(lab.astype(np.float32))


# (width, height, patch radius, search radius, scattering, luma, chroma, expected chunk, expected interior chunks)
CASES = [
    (170, 150, 2, 7, 0.0, 0.5, 1.0, (64, 51), 1),    # the module's defaults: 225 offsets
    (256, 207, 2, 2, 0.0, 0.5, 1.0, (72, 69), 2),    # the largest chunk: every accumulator slot of a thread in use
    (300, 250, 1, 3, 0.0, 1.0, 1.0, (64, 63), 6),    # patch radius 1 (3 chains per column, 3 segments), no blend
    (300, 250, 3, 2, 0.0, 0.3, 0.8, (64, 63), 6),    # patch radius 3 (7 chains, one segment)
    (300, 250, 2, 2, 0.9, 0.5, 1.0, (64, 63), 6),    # scattered offsets: shifts beyond the search radius
]


@pytest.mark.parametrize("w,h,P,K,scat,luma,chroma,chunk,n_interior", CASES)
def test_interior_chunk_kernel_on_the_host_equals_the_oracle(host_kernel, oracle_lib, w, h, P, K, scat, luma, chroma, chunk,
                                                             n_interior):
    o = oracle_lib
    img = _lab(w, h, 11 + P + K)
    p = NlmParams(scat, 1.0, luma, chroma, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert (cw, ch) == chunk
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm2_host_run(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(scat),
                                   C.c_float(p.sharpness), p.norm, C.c_float(luma), C.c_float(chroma), C.byref(seen))
    assert rc >= 0 and seen.value == n_interior  # rc: bit 0 = the tight layout ran, bit 1 = the four-table schedule
    written = ~np.isnan(got[..., 0])
    assert int(written.sum()) == n_interior * cw * ch  # interior chunks only, each pixel of them
    assert np.array_equal(got[written].view(np.uint32), want[written].view(np.uint32))


# denoise (profiled) in non-local-means mode: the weight with the centre pixel's term (center_weight >= 0, nlmeans_core.c:416-424);
# the module's defaults there are patch radius 1, search radius 7, central pixel weight 0.1
# (width, height, patch radius, search radius, scattering, center_weight, expected chunk, interior chunks)
CASES_CENTER = [
    (170, 150, 1, 7, 0.0, 0.1, (64, 51), 1),     # the module's defaults: 225 offsets
    (300, 250, 1, 3, 0.0, 0.0, (64, 63), 6),     # central pixel weight 0: the division by 1, the floor at -2
    (300, 250, 2, 2, 0.0, 1.0, (64, 63), 6),     # patch radius 2, the centre as heavy as the patch
    (256, 207, 3, 2, 0.9, 0.4, (72, 69), 2),     # patch radius 3, scattered offsets, the largest chunk (two tables)
]


@pytest.mark.parametrize("w,h,P,K,scat,cw_,chunk,n_interior", CASES_CENTER)
def test_interior_chunk_kernel_with_the_centre_pixel_term_equals_the_oracle(host_kernel, oracle_lib, w, h, P, K, scat, cw_, chunk, n_interior):
    o = oracle_lib
    rng = np.random.default_rng(77 + P + K)
    # what the module hands the core: a variance-stabilised frame, values around 0 .. a few units (denoiseprofile.c:1599-1680)
    img = np.ascontiguousarray((_lab(w, h, 21 + P + K) * np.float32(0.05) + rng.normal(0, 0.3, (h, w, 4))).astype(np.float32))
    p = NlmParams(scat, 1.0, 1.0, 1.0, cw_, 1.3, P, K, (C.c_float * 4)(1.0, 1.0, 1.0, 1.0))
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert (cw, ch) == chunk
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm2_host_run_center(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(scat),
                                          C.c_float(p.sharpness), p.norm, C.c_float(1.0), C.c_float(1.0), C.byref(seen), C.c_float(cw_))
    assert rc >= 0 and seen.value == n_interior
    written = ~np.isnan(got[..., 0])
    assert int(written.sum()) == n_interior * cw * ch
    assert np.array_equal(got[written].view(np.uint32), want[written].view(np.uint32))
    # the weights are not all the floor's: the centre term is exercised
    assert np.unique(want[written][..., 0]).size > 100


def test_configurations_outside_the_kernel_are_refused(host_kernel):
    img = np.zeros((100, 100, 4), np.float32)
    norm = (C.c_float * 4)(1, 1, 1, 1)
    args = (ck.ptr(img), ck.ptr(img), 100, 100, 64, 50)
    tail = (C.c_float(1.0), C.c_float(0.0), C.c_float(10.0), norm, C.c_float(1.0), C.c_float(1.0), None)
    assert host_kernel.nlm2_host_run(*args, 4, 2, *tail) == -1   # patch radius 4: the pipelined kernel's
    assert host_kernel.nlm2_host_run(*args, 2, 20, *tail) == -2  # 20 px of shift: the window does not fit the pitch


# ---- the third version (nlm3_body.h, launched as nlm_chunks_v3): wave roles, pixels that slide through registers ----
# (width, height, search radius, scattering, luma, chroma, expected chunk, interior chunks, does the body take it)
CASES3 = [
    (170, 150, 7, 0.0, 0.5, 1.0, (64, 51), 1, True),    # the module's defaults: 15 rows of 15 offsets; 64-column chunks
    (260, 168, 7, 0.0, 1.0, 1.0, (72, 56), 2, True),   # the 100 MP frame's chunk: 72 x 56, no blend
    (330, 168, 3, 0.0, 0.3, 0.8, (72, 56), 3, True),    # rows of 7 offsets (less than one ring turn of a C lane), 3 chunks
    (256, 207, 2, 0.0, 0.5, 1.0, (72, 69), 2, False),   # 69-row chunks: the second version's
    (300, 250, 2, 0.9, 0.5, 1.0, (64, 63), 6, False),   # scattered offsets are no rows of consecutive shifts
]


@pytest.mark.parametrize("w,h,K,scat,luma,chroma,chunk,n_interior,taken", CASES3)
def test_third_version_on_the_host_equals_the_oracle(host_kernel, oracle_lib, w, h, K, scat, luma, chroma, chunk, n_interior, taken):
    o = oracle_lib
    P = 2
    img = _lab(w, h, 5 + K)
    p = NlmParams(scat, 1.0, luma, chroma, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert (cw, ch) == chunk
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm3_host_run(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(scat),
                                   C.c_float(p.sharpness), p.norm, C.c_float(luma), C.c_float(chroma), C.byref(seen))
    assert rc == (1 if taken else 0)
    if not taken:
        return
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    assert seen.value == n_interior
    written = ~np.isnan(got[..., 0])
    assert int(written.sum()) == n_interior * cw * ch
    assert np.array_equal(got[written].view(np.uint32), want[written].view(np.uint32))


# the border ring with the BORDER body: frames whose every chunk is in the ring, partial last chunks (odd width, a last row
# of chunks lower than 10 rows keeps the first version's body and stays unwritten), offsets longer than a chunk is far
# from the edge
CASES3B = [
    (170, 150, 5, 0.5, 1.0),   # (search radius 5: rows of 11 offsets, more than one turn of a C lane's ring of ten, at half the
    (260, 168, 7, 1.0, 1.0),   #  module's 225 offsets -- the host harness is 1 024 OS threads on a barrier; one case keeps the 225)
    (151, 140, 3, 0.3, 0.8),   # 64 x 51 chunks, the last 23 columns (odd) and 38 rows
    (181, 140, 5, 0.5, 1.0),   # 72 x 51, the last 37 columns
    (173, 159, 5, 0.5, 0.9),   # 68 x 53, the last 37 columns
]


@pytest.mark.parametrize("w,h,K,luma,chroma", CASES3B)
def test_third_version_border_ring_on_the_host_equals_the_oracle(host_kernel, oracle_lib, w, h, K, luma, chroma):
    o = oracle_lib
    P = 2
    img = _lab(w, h, 11 + K)
    p = NlmParams(0.0, 1.0, luma, chroma, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm3_host_run_all(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(0.0),
                                       C.c_float(p.sharpness), p.norm, C.c_float(luma), C.c_float(chroma), C.byref(seen))
    if rc == 0:
        pytest.skip("chunk grid %d x %d is not the third version's" % (cw, ch))
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    written = ~np.isnan(got[..., 0])
    assert seen.value > 0 and int(written.sum()) > 0
    bad = written & (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    assert int(bad.sum()) == 0, "%d of %d written pixels differ; first at %s" % (int(bad.sum()), int(written.sum()), np.argwhere(bad)[:5].tolist())
    # every chunk with at least 10 rows is written
    nrows_last = h - (h - 1) // ch * ch
    expect = w * (h if nrows_last >= 10 else h - nrows_last)
    assert int(written.sum()) == expect


# ---- the fused variant (nlm3_body.h FUSED, launched as nlm_chunks_v4): three tables, the row recurrence inside the C role
#      (a DPP lane shift on the device, a per-wave exchange here).  The chunk grids of 57 - 64 rows -- the 45 MP and 60 MP
#      frames': 64 -- and, for the A/B against the third version, the ones that one takes too.
# (width, height, search radius, luma, chroma, expected chunk, interior chunks)
CASES4 = [
    (260, 192, 7, 0.5, 1.0, (72, 64), 2),    # the 45 MP / 60 MP frames' chunk: 72 x 64, the module's defaults
    (260, 171, 5, 1.0, 1.0, (72, 57), 2),    # 57 rows: one more than the third version's
    (170, 183, 3, 0.3, 0.8, (64, 61), 1),    # 64 x 61, rows of 7 offsets
    (260, 168, 5, 0.5, 1.0, (72, 56), 2),    # the 100 MP frame's chunk on the fused schedule
]


@pytest.mark.parametrize("w,h,K,luma,chroma,chunk,n_interior", CASES4)
def test_fused_variant_on_the_host_equals_the_oracle(host_kernel, oracle_lib, w, h, K, luma, chroma, chunk, n_interior):
    o = oracle_lib
    P = 2
    img = _lab(w, h, 7 + K)
    p = NlmParams(0.0, 1.0, luma, chroma, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert (cw, ch) == chunk
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm4_host_run(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(0.0),
                                   C.c_float(p.sharpness), p.norm, C.c_float(luma), C.c_float(chroma), C.byref(seen))
    assert rc == 1
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    assert seen.value == n_interior
    written = ~np.isnan(got[..., 0])
    assert int(written.sum()) == n_interior * cw * ch
    bad = written & (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    assert int(bad.sum()) == 0, "%d of %d written pixels differ; first at %s" % (int(bad.sum()), int(written.sum()), np.argwhere(bad)[:5].tolist())


def test_fused_variant_refuses_what_it_does_not_fit(host_kernel):
    img = np.zeros((207, 256, 4), np.float32)
    norm = (C.c_float * 4)(1, 1, 1, 1)
    tail = (C.c_float(1.0), C.c_float(0.0), C.c_float(10.0), norm, C.c_float(1.0), C.c_float(1.0), None)
    assert host_kernel.nlm4_host_run(ck.ptr(img), ck.ptr(img), 256, 207, 72, 69, 2, 2, *tail) == 0   # 69 rows
    assert host_kernel.nlm4_host_run(ck.ptr(img), ck.ptr(img), 256, 207, 72, 64, 3, 2, *tail) == 0   # patch radius 3


CASES4B = [
    (170, 128, 5, 0.5, 1.0),   # 64-row chunks, every chunk in the ring
    (181, 171, 3, 0.3, 0.8),   # 72 x 57, the last 37 columns
    (151, 140, 3, 1.0, 1.0),   # 64 x 51 chunks, the last 23 columns (odd) and 38 rows
]


@pytest.mark.parametrize("w,h,K,luma,chroma", CASES4B)
def test_fused_variant_border_ring_on_the_host_equals_the_oracle(host_kernel, oracle_lib, w, h, K, luma, chroma):
    o = oracle_lib
    P = 2
    img = _lab(w, h, 13 + K)
    p = NlmParams(0.0, 1.0, luma, chroma, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm4_host_run_all(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(0.0),
                                       C.c_float(p.sharpness), p.norm, C.c_float(luma), C.c_float(chroma), C.byref(seen))
    assert rc == 1, "chunk grid %d x %d" % (cw, ch)
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    written = ~np.isnan(got[..., 0])
    assert seen.value > 0 and int(written.sum()) > 0
    bad = written & (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    assert int(bad.sum()) == 0, "%d of %d written pixels differ; first at %s" % (int(bad.sum()), int(written.sum()), np.argwhere(bad)[:5].tolist())
    nrows_last = h - (h - 1) // ch * ch
    expect = w * (h if nrows_last >= 10 else h - nrows_last)
    assert int(written.sum()) == expect


# ---- tall chunk grids (65 - 69 rows: the 24 MP frame's 69, 42 MP's 68, 150 MP's 67; round 5): the fused body on a chunk's first
#      64 rows, exporting the column sums behind them, and nlm_tail_body.h on the rows that are left
# (width, height, search radius, luma, chroma, expected chunk, interior chunks)
CASES_TALL = [
    (256, 207, 2, 0.5, 1.0, (72, 69), 2),    # 69 rows: five tail rows, rows of 5 offsets
    (260, 207, 7, 1.0, 1.0, (72, 69), 2),    # ... with the module's 225 offsets, no blend
    (260, 204, 3, 0.5, 1.0, (72, 68), 2),    # 68 rows
    (170, 201, 3, 0.3, 0.8, (64, 67), 1),    # 67 rows, 64-column chunks (slots beyond the chunk in the tail's row batches)
    (260, 198, 3, 0.5, 1.0, (72, 66), 2),    # 66 rows
    (250, 195, 3, 0.5, 0.9, (68, 65), 2),    # 65 rows: ONE tail row; 68-column chunks
]


@pytest.mark.parametrize("w,h,K,luma,chroma,chunk,n_interior", CASES_TALL)
def test_tall_chunks_head_and_tail_on_the_host_equal_the_oracle(host_kernel, oracle_lib, w, h, K, luma, chroma, chunk, n_interior):
    o = oracle_lib
    P = 2
    img = _lab(w, h, 17 + K)
    p = NlmParams(0.0, 1.0, luma, chroma, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert (cw, ch) == chunk
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm_tall_host_run(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(0.0),
                                       C.c_float(p.sharpness), p.norm, C.c_float(luma), C.c_float(chroma), C.byref(seen))
    assert rc == 1
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    assert seen.value == n_interior
    written = ~np.isnan(got[..., 0])
    assert int(written.sum()) == n_interior * cw * ch, "head and tail together write every pixel of the interior chunks"
    bad = written & (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    assert int(bad.sum()) == 0, "%d of %d written pixels differ; first at %s" % (int(bad.sum()), int(written.sum()), np.argwhere(bad)[:5].tolist())


def test_tall_pair_refuses_what_it_does_not_fit(host_kernel):
    img = np.zeros((207, 256, 4), np.float32)
    norm = (C.c_float * 4)(1, 1, 1, 1)
    tail = (C.c_float(1.0), C.c_float(0.0), C.c_float(10.0), norm, C.c_float(1.0), C.c_float(1.0), None)
    assert host_kernel.nlm_tall_host_run(ck.ptr(img), ck.ptr(img), 256, 192, 72, 64, 2, 2, *tail) == 0   # 64 rows: the fused variant's own
    assert host_kernel.nlm_tall_host_run(ck.ptr(img), ck.ptr(img), 256, 207, 72, 69, 3, 2, *tail) == 0   # patch radius 3
    assert host_kernel.nlm_tall_host_run(ck.ptr(img), ck.ptr(img), 256, 207, 72, 69, 2, 2, C.c_float(1.0), C.c_float(0.9), *tail[2:]) == 0  # scattered


# ... and the outermost ring of a tall grid: the BORDER bodies of head and tail (frames whose every chunk is in the ring, partial
# last chunks: a last row of chunks of 65 - 68 rows has a tail of its own height, one of 10 - 64 rows has none, one lower than ten
# rows keeps the first version's body and stays unwritten)
CASES_TALL_B = [
    (170, 138, 3, 0.5, 1.0),   # 64 x 69, every chunk in the ring (49 offsets: the 225 of the module's defaults run on the interior
                               # case above and, on the device, in tests/test_gpu_nlmeans.py)
    (260, 207, 3, 1.0, 1.0),   # 72 x 69: three rows of chunks, the middle ones interior
    (181, 204, 3, 0.3, 0.8),   # 72 x 68, the last 37 columns
    (151, 274, 3, 0.5, 0.9),   # 64 x 69 (274 = 3 x 69 + 67): the last row of chunks 67 rows, the last 23 columns (odd)
    (200, 196, 2, 0.5, 1.0),   # 68 x 66 (196 = 2 x 66 + 64): the last row of chunks 64 rows -- a head without a tail
]


@pytest.mark.parametrize("w,h,K,luma,chroma", CASES_TALL_B)
def test_tall_chunks_border_ring_on_the_host_equals_the_oracle(host_kernel, oracle_lib, w, h, K, luma, chroma):
    o = oracle_lib
    P = 2
    img = _lab(w, h, 19 + K)
    p = NlmParams(0.0, 1.0, luma, chroma, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert 65 <= ch <= 69, (cw, ch)
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm_tall_host_run_all(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(0.0),
                                           C.c_float(p.sharpness), p.norm, C.c_float(luma), C.c_float(chroma), C.byref(seen))
    assert rc == 1, "chunk grid %d x %d" % (cw, ch)
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    written = ~np.isnan(got[..., 0])
    assert seen.value > 0 and int(written.sum()) > 0
    bad = written & (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    assert int(bad.sum()) == 0, "%d of %d written pixels differ; first at %s" % (int(bad.sum()), int(written.sum()), np.argwhere(bad)[:5].tolist())
    nrows_last = h - (h - 1) // ch * ch
    expect = w * (h if nrows_last >= 10 else h - nrows_last)
    assert int(written.sum()) == expect


# ---- round 6: patch radius 1 and the weight with the centre pixel's term (denoise (profiled)'s non-local-means mode, whose
#      defaults are patch radius 1, search radius 7, central pixel weight 0.1) on the third version's schedule and on the fused one:
#      nlm3_body.h P / CENTER.  Interior chunks and the outermost ring (BORDER bodies).
# (width, height, patch radius, search radius, center_weight (< 0: the plain weight), fused, ring too, expected chunk)
# (the host harness is 1 024 OS threads on a barrier: search radii kept small; the 225 offsets of the defaults run on the device,
#  tests/test_gpu_nlmeans.py)
CASES_R6 = [
    (260, 168, 1, 5, -1.0, False, False, (72, 56)),   # patch radius 1, the 100 MP frame's chunk: nine chains of <= 7 terms, rows of 11 offsets
    (260, 168, 1, 5, 0.1, False, False, (72, 56)),    # ... with the centre term: denoise (profiled)'s defaults but for the search radius
    (170, 150, 1, 3, 0.1, False, True, (64, 51)),     # 64 x 51 chunks, the ring with the centre term
    (260, 192, 1, 4, -1.0, True, False, (72, 64)),    # the fused schedule (the 45 / 60 MP frames' 64-row chunks), patch radius 1
    (170, 128, 1, 2, 0.4, True, True, (64, 64)),      # ... with the centre term, every chunk in the ring
    (260, 168, 2, 4, 1.0, False, False, (72, 56)),    # patch radius 2 with the centre as heavy as the patch
    (181, 171, 2, 2, 0.0, True, True, (72, 57)),      # fused, radius 2, central weight 0 (the division by 1, the floor at -2), ring
    (151, 140, 1, 3, -1.0, False, True, (64, 51)),    # plain weight, radius 1, ring with an odd last chunk width and a low last row
]


@pytest.mark.parametrize("w,h,P,K,cw_,fused,ring,chunk", CASES_R6)
def test_patch_radius_one_and_the_centre_term_on_the_third_version(host_kernel, oracle_lib, w, h, P, K, cw_, fused, ring, chunk):
    o = oracle_lib
    center = cw_ >= 0
    if center:
        rng = np.random.default_rng(91 + P + K)
        img = np.ascontiguousarray((_lab(w, h, 23 + P + K) * np.float32(0.05) + rng.normal(0, 0.3, (h, w, 4))).astype(np.float32))
        p = NlmParams(0.0, 1.0, 1.0, 1.0, cw_, 1.3, P, K, (C.c_float * 4)(1.0, 1.0, 1.0, 1.0))
    else:
        img = _lab(w, h, 19 + P + K)
        p = NlmParams(0.0, 1.0, 0.5, 1.0, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert (cw, ch) == chunk
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm3_host_run_ex(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(0.0),
                                      C.c_float(p.sharpness), p.norm, C.c_float(p.luma), C.c_float(p.chroma), C.byref(seen),
                                      C.c_float(cw_), int(ring), int(fused))
    assert rc == 1
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    written = ~np.isnan(got[..., 0])
    assert seen.value > 0 and int(written.sum()) > 0
    bad = written & (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    assert int(bad.sum()) == 0, "%d of %d written pixels differ; first at %s" % (int(bad.sum()), int(written.sum()), np.argwhere(bad)[:5].tolist())
    if ring:
        nrows_last = h - (h - 1) // ch * ch
        assert int(written.sum()) == w * (h if nrows_last >= 10 else h - nrows_last)
    else:
        assert int(written.sum()) == seen.value * cw * ch
    if center:
        assert np.unique(want[written][..., 0]).size > 100  # the weights are not all the floor's


# ... and on the tall chunk grids (65 - 69 rows: the 24 MP frame's 69): the fused head + nlm_tail_body.h with P / CENTER
# (width, height, patch radius, search radius, center_weight, ring too, expected chunk)
CASES_R6_TALL = [
    (260, 207, 1, 4, 0.1, False, (72, 69)),    # denoise (profiled)'s defaults (but for the search radius) on the 24 MP frame's chunk
    (170, 138, 1, 2, -1.0, True, (64, 69)),    # patch radius 1, plain weight, every chunk in the ring
    (170, 201, 2, 2, 0.5, True, (64, 67)),     # patch radius 2 with the centre term, 67-row chunks, ring
    (250, 195, 1, 3, 0.0, False, (68, 65)),    # ONE tail row, central weight 0
]


@pytest.mark.parametrize("w,h,P,K,cw_,ring,chunk", CASES_R6_TALL)
def test_patch_radius_one_and_the_centre_term_on_tall_chunks(host_kernel, oracle_lib, w, h, P, K, cw_, ring, chunk):
    o = oracle_lib
    center = cw_ >= 0
    if center:
        rng = np.random.default_rng(93 + P + K)
        img = np.ascontiguousarray((_lab(w, h, 27 + P + K) * np.float32(0.05) + rng.normal(0, 0.3, (h, w, 4))).astype(np.float32))
        p = NlmParams(0.0, 1.0, 1.0, 1.0, cw_, 1.3, P, K, (C.c_float * 4)(1.0, 1.0, 1.0, 1.0))
    else:
        img = _lab(w, h, 29 + P + K)
        p = NlmParams(0.0, 1.0, 0.5, 1.0, -1.0, 3000.0 / 51.0, P, K, (C.c_float * 4)(1 / 120.0 ** 2, 1 / 512.0 ** 2, 1 / 512.0 ** 2, 1.0))
    o.oracle_nlmeans_slice_height.restype = C.c_int
    o.oracle_nlmeans_slice_width.restype = C.c_int
    ch, cw = o.oracle_nlmeans_slice_height(h), o.oracle_nlmeans_slice_width(w)
    assert (cw, ch) == chunk
    got = np.full_like(img, np.nan)
    seen = C.c_int(0)
    rc = host_kernel.nlm_tall_host_run_ex(ck.ptr(img), ck.ptr(got), w, h, cw, ch, P, K, C.c_float(1.0), C.c_float(0.0),
                                          C.c_float(p.sharpness), p.norm, C.c_float(p.luma), C.c_float(p.chroma), C.byref(seen),
                                          C.c_float(cw_), int(ring))
    assert rc == 1
    want = np.zeros_like(img)
    o.oracle_nlmeans_core(ck.ptr(img), ck.ptr(want), w, h, C.byref(p))
    written = ~np.isnan(got[..., 0])
    assert seen.value > 0 and int(written.sum()) > 0
    bad = written & (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    assert int(bad.sum()) == 0, "%d of %d written pixels differ; first at %s" % (int(bad.sum()), int(written.sum()), np.argwhere(bad)[:5].tolist())
    if not ring:
        assert int(written.sum()) == seen.value * cw * ch
