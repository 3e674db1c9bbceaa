"""The gfx950 kernel that keeps an AMaZE tile on chip (ansel_amd/csrc/amaze_stream_body.h, launched as amaze_frame /
amaze_stream) compiled for the HOST (tests/native/amaze_host.cpp): a workgroup is 640 fibers run from barrier to barrier,
its LDS a heap block with a shadow that flags every race and every ring slot read after it was overwritten.  Against the
oracle, bit for bit, on every tile the kernel takes -- all but those of a frame's last tile row / column whose mirrored strips
overrun their plane or whose width is odd (amz::stream_tile_ok()), which belong to the first kernel.  The same
source runs on the device; this pins its schedule (lags, ring depths, the thread halves that share a phase, the votes a single
wave walks) without a GPU.  The -m gpu tests then only have to show that the device executes it the same."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import checkers as ck
from ansel_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "native", "libamaze_host.so")
SRC = os.path.join(ROOT, "tests", "native", "amaze_host.cpp")
HDR = os.path.join(ROOT, "ansel_amd", "csrc", "amaze_stream_body.h")
TS = 160


def build(so=SO, defines=()):
    if defines or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", *defines,
                               "-I" + os.path.join(ROOT, "ansel_amd", "csrc"), SRC, "-o", so])
    return so


@pytest.fixture(scope="module")
def host_kernel():
    lib = C.CDLL(build())
    lib.amaze_host_run.restype = C.c_int
    return lib


def textured(w, h, seed, gain=1.7):
    """a scene with fine checkerboards and stripes laid over parts of it: they switch the Nyquist branches on"""
    raw = synth.bayer_mosaic(w, h, seed=seed).astype(np.float32)
    cfa = ((raw - 512) / np.float32(synth.WHITE - 512) * np.float32(gain)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    a = (slice(h // 8, h // 2), slice(w // 6, 2 * w // 3))
    cfa[a] *= (0.55 + 0.45 * ((xx[a] + yy[a]) & 1)).astype(np.float32)
    b = (slice(h // 2, 7 * h // 8), slice(w // 4, 7 * w // 8))
    cfa[b] *= (0.6 + 0.4 * ((xx[b] >> 1) & 1)).astype(np.float32)
    c = (slice(h // 3, 2 * h // 3), slice(3 * w // 4, w - 8))
    cfa[c] *= (0.6 + 0.4 * (yy[c] & 1)).astype(np.float32)
    return cfa


def run(lib, cfa, filters, pm):
    h, w = cfa.shape
    got = np.full((h, w, 4), -7.0, np.float32)
    ns, na, err = C.c_int(), C.c_int(), C.create_string_buffer(512)
    ne = lib.amaze_host_run(ck.ptr(cfa), ck.ptr(got), w, h, C.c_uint32(filters), C.c_float(min(pm[:3])), C.byref(ns), C.byref(na), err, 512)
    return got, ne, err.value.decode(), ns.value, na.value


def oracle(cfa, filters, pm):
    h, w = cfa.shape
    piece = abi.Piece.make(w, h, filters=filters, channels=1, processed_maximum=pm)
    d = abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0)
    want = np.full((h, w, 4), -7.0, np.float32)
    assert ck.call(ck.oracle(), "oracle_demosaic", piece, d, cfa, want) == 0
    return want


def stream_tile_ok(w, h, top, left):
    """amz::stream_tile_ok(), restated"""
    bottom, right = min(top + TS, h + 16), min(left + TS, w + 16)
    rrmax = h - top if bottom > h else bottom - top
    ccmax = w - left if right > w else right - left
    if (rrmax < bottom - top and rrmax > TS - 16) or (ccmax < right - left and ccmax > TS - 16):
        return False
    return not (right - left < TS and (right - left) & 1)


def kept_pixels(w, h):
    """the pixels the tiles of the on-chip kernel write, and how many tiles those are / there are"""
    mask = np.zeros((h, w), bool)
    ok = n = 0
    for top in range(-16, h, TS - 32):
        for left in range(-16, w, TS - 32):
            n += 1
            if stream_tile_ok(w, h, top, left):
                ok += 1
                bottom, right = min(top + TS, h + 16), min(left + TS, w + 16)
                mask[max(top + 16, 0):max(min(bottom - 16, h), 0), max(left + 16, 0):max(min(right - 16, w), 0)] = True
    return mask, ok, n


def check(lib, cfa, filters, pm=(1.5, 1.0, 1.2, 1.0), all_tiles=None):
    h, w = cfa.shape
    got, ne, err, ns, na = run(lib, cfa, filters, pm)
    assert ne == 0, "%d schedule errors, the first: %s" % (ne, err)
    mask, ok, n = kept_pixels(w, h)
    assert (ns, na) == (ok, n)
    if all_tiles is not None:
        assert (ok == n) == all_tiles
    written = got[..., 0] != -7.0
    assert np.array_equal(written, mask)
    assert (got[..., 3] == -7.0).all()  # alpha is not written
    want = oracle(cfa, filters, pm)
    diff = (ck.ulp_diff(got, want) > 0) & written[..., None]
    assert int(diff.sum()) == 0, "%d values differ" % int(diff.sum())


@pytest.mark.parametrize("filters", [0x94949494, 0x49494949, 0x61616161, 0x16161616])
def test_streaming_tiles_equal_the_oracle(host_kernel, filters):
    """a frame of odd size: 9 of its 20 tiles (the mirrored top / left border and its corner, the bottom tile row with its
    mirrored strip among them); the last tile column is of odd width and stays with the first kernel.  Every CFA phase"""
    check(host_kernel, textured(517, 389, seed=906), filters, all_tiles=False)


@pytest.mark.parametrize("w,h,filters", [(600, 400, 0x94949494), (496, 432, 0x16161616), (288, 330, 0x49494949), (64, 64, 0x61616161),
                                         (160, 160, 0x94949494), (401, 333, 0x49494949), (305, 304, 0x61616161)])
def test_streaming_tiles_cut_by_the_frame(host_kernel, w, h, filters):
    """tiles the frame cuts on the right and at the bottom: shorter / narrower planes, the mirrored right and bottom strips and
    their corners.  The first four frames (the geometry of the 24 MP and 100 MP frames among them: a last column 144 or 32 wide,
    a last row 64 high) have EVERY tile on chip"""
    check(host_kernel, textured(w, h, seed=w + h), filters, all_tiles=(w, h) in ((600, 400), (496, 432), (288, 330), (64, 64), (160, 160)))


def test_streaming_tiles_many(host_kernel):
    """24 full tiles of a frame whose textures flag most of them for the Nyquist refinement -- the frames in which the last
    kept row of a tile depends on the bytes behind the second flag plane (profiles/r03_amaze_alias_probe.txt)"""
    check(host_kernel, textured(800, 600, seed=21), 0x49494949)


def test_streaming_tiles_highlights_and_flat_areas(host_kernel):
    """clipped highlights (the > clip_pt branches), constant planes (0 / 0 guarded by eps), zeros (the +-0 quotients of the
    exact binary32 replacement of the reference's binary64 divisions)"""
    w, h = 450, 330
    raw = synth.bayer_mosaic(w, h, seed=9).astype(np.float32)
    cfa = ((raw - 512) / np.float32(synth.WHITE - 512) * np.float32(3.0)).astype(np.float32)
    cfa[60:120, 80:200] = 1.0
    cfa[150:200, 30:90] = 0.0
    cfa[20:50, 220:260] = -0.05
    cfa = np.minimum(cfa, 1.0).astype(np.float32)
    check(host_kernel, cfa, synth.FILTERS_RGGB, pm=(1.0, 1.0, 1.0, 1.0))


def test_streaming_tiles_adversarial_values(host_kernel):
    """tiny values (their squares are denormals), huge values and negatives: every stage's selects and exact divisions on
    the same bits as the oracle.  (Photosites that are themselves near the smallest normal make the reference's exponent
    tricks, amaze.cc:77-121, wrap into NaNs; which of two NaN operands an addition hands on -- its sign ends up in the
    result through those tricks -- is the compiler's choice, on x86 and on the GPU: not a function of the input.)"""
    w, h = 320, 320
    rng = np.random.default_rng(5)
    cfa = textured(w, h, seed=4)
    cfa[40:80, 40:120] *= np.float32(1e-30)
    cfa[100:140, 150:260] *= np.float32(1e30)
    cfa[200:230, 30:200] *= -1.0
    cfa[rng.integers(0, h, 300), rng.integers(0, w, 300)] = 0.0
    check(host_kernel, cfa.astype(np.float32), 0x61616161)


def test_the_checker_sees_a_broken_schedule(tmp_path):
    """the harness is only worth something if it fails when the schedule is wrong: one row of lag less behind the
    Nyquist test (S6 reads flags S5 has not produced) must be reported as a stale ring slot"""
    src = open(HDR).read()
    assert "L_S6 = 13" in src
    hdr_dir = tmp_path / "csrc"
    hdr_dir.mkdir()
    (hdr_dir / "amaze_stream_body.h").write_text(src.replace("L_S6 = 13", "L_S6 = 12"))
    so = str(tmp_path / "libamaze_host_broken.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I" + str(hdr_dir), SRC, "-o", so])
    lib = C.CDLL(so)
    lib.amaze_host_run.restype = C.c_int
    _, ne, err, _, _ = run(lib, textured(300, 200, seed=3), 0x94949494, (1.5, 1.0, 1.2, 1.0))
    assert ne > 0 and ("stale ring slot" in err or "another thread" in err), err


@pytest.mark.parametrize("w,h,n", [(600, 400, 2), (496, 432, 3), (288, 330, 2)])
def test_streaming_tiles_on_row_bands(host_kernel, w, h, n):
    """the kernel on a row band (dt_hip_plan_bands() with the AMaZE demosaic: whole tile rows, 16 mosaic rows of either
    neighbour): the band's buffers hold its own rows and the halo only -- everything else is NaN here --, the tile rows are the
    frame's own, and the assembled bands are the oracle's frame"""
    cfa = textured(w, h, seed=w)
    filters, pm = 0x61616161, (1.5, 1.0, 1.2, 1.0)
    want = oracle(cfa, filters, pm)
    tile_rows = (h + 127) // 128
    assert tile_rows >= n
    lib = host_kernel
    lib.amaze_host_run_band.restype = C.c_int
    got = np.full((h, w, 4), -7.0, np.float32)
    for k in range(n):
        tv0, tv1 = k * tile_rows // n, (k + 1) * tile_rows // n
        row0, row1 = tv0 * 128, (tv1 * 128 if tv1 < tile_rows else h)
        top, bottom = (16 if tv0 else 0), min(16, h - row1)
        band_in = np.ascontiguousarray(cfa[row0 - top:row1 + bottom])
        band_out = np.full((row1 - row0, w, 4), -7.0, np.float32)
        ns, na, err = C.c_int(), C.c_int(), C.create_string_buffer(512)
        ne = lib.amaze_host_run_band(ck.ptr(band_in), ck.ptr(band_out), w, h, C.c_uint32(filters), C.c_float(min(pm[:3])),
                                     row0 - top, row0, row1, tv0, tv1, C.byref(ns), C.byref(na), err, 512)
        assert ne == 0, err.value.decode()
        assert ns.value == na.value  # every tile of these frames is one the kernel takes
        got[row0:row1] = band_out
    assert int((ck.ulp_diff(got[..., :3], want[..., :3]) > 0).sum()) == 0
    assert (got[..., 3] == -7.0).all()
