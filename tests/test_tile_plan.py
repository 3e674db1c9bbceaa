"""CPU: dt_hip_plan_tiles_ptp() (pure host function of libansel_hip.so) against the Python restatement of
src/develop/tiling.c:868-979 in tests/tile_plan.py, and the properties the tile loop relies on."""
import ctypes as C

import numpy as np
import pytest

import tile_plan as tp
from ansel_amd import abi, lib


def _tiling(factor_cl=2.0, maxbuf_cl=1.0, overhead=0, overlap=0, xalign=1, yalign=1):
    return abi.Tiling(factor_cl, factor_cl, maxbuf_cl, maxbuf_cl, overhead, overlap, xalign, yalign)


def _plan(w, h, ib, ob, t, filters, avail, memalloc=2 ** 40, mw=1 << 20, mh=1 << 20):
    pl = abi.TilePlan()
    rc = lib.load().dt_hip_plan_tiles_ptp(w, h, ib, ob, C.byref(t), filters, avail, memalloc, mw, mh, C.byref(pl))
    return rc, {k: getattr(pl, k) for k, _ in abi.TilePlan._fields_}


CASES = []
for (w, h) in [(6000, 4000), (11648, 8736), (4000, 6000), (512, 384), (333, 517), (64, 64)]:
    for (ib, ob) in [(16, 16), (4, 16), (2, 4)]:
        for overlap, xa in [(0, 1), (5, 2), (10, 2), (128, 1), (37, 3)]:
            for frac in (2.0, 0.7, 0.2, 0.03):
                CASES.append((w, h, ib, ob, overlap, xa, frac))


@pytest.mark.parametrize("w,h,ib,ob,overlap,xa,frac", CASES)
def test_plan_matches_the_restatement(w, h, ib, ob, overlap, xa, frac):
    t = _tiling(factor_cl=2.5, maxbuf_cl=1.0, overlap=overlap, xalign=xa, yalign=xa)
    avail = int(w * h * max(ib, ob) * 2.5 * frac)
    filters = 9 if xa == 3 else (0x94949494 if xa == 2 else 0)
    rc, got = _plan(w, h, ib, ob, t, filters, avail)
    want = tp.plan(w, h, ib, ob, t, filters, avail, 2 ** 40, 1 << 20, 1 << 20)
    if want["tiles_x"] * want["tiles_y"] > 10000:
        assert rc != abi.DT_HIP_SUCCESS
        return
    assert rc == abi.DT_HIP_SUCCESS and got == want
    if 3 * overlap <= min(w, h):  # else the reference switches to square tiles whatever the memory (:901-907)
        if frac >= 1.0:
            assert got["tiles_x"] == got["tiles_y"] == 1 and (got["width"], got["height"]) == (w, h)
        else:
            assert got["tiles_x"] * got["tiles_y"] > 1
    # every tile origin keeps the module's alignment (CFA phase), and the kept parts cover the frame
    assert got["tile_wd"] % xa == 0 or got["tiles_x"] == 1 or got["tile_wd"] == 1
    cover = np.zeros((h // 8 + 1, w // 8 + 1), bool) if w * h > 10 ** 6 else np.zeros((h, w), bool)
    s = 8 if w * h > 10 ** 6 else 1
    if got["tile_wd"] > 2 * got["overlap"] or got["tiles_x"] == 1:
        for x0, y0, wd, ht, ox, oy in tp.tiles(got, w, h):
            cover[(y0 + oy + s - 1) // s:(y0 + ht + s - 1) // s, (x0 + ox + s - 1) // s:(x0 + wd + s - 1) // s] = True
        assert cover[:(h + s - 1) // s, :(w + s - 1) // s].all()


def test_device_limits_bound_the_tile():
    t = _tiling()
    rc, p = _plan(6000, 4000, 16, 16, t, 0, 2 ** 40, memalloc=16 * 2000 * 1500)   # largest allocation: 2000 x 1500 px
    assert rc == abi.DT_HIP_SUCCESS and p["width"] * p["height"] * 16 <= 16 * 2000 * 1500 and p["tiles_x"] * p["tiles_y"] > 1
    rc, p = _plan(6000, 4000, 16, 16, t, 0, 2 ** 40, mw=1024, mh=1024)              # largest image side
    assert rc == abi.DT_HIP_SUCCESS and p["width"] <= 1024 and p["height"] <= 1024
    assert lib.load().dt_hip_plan_tiles_ptp(0, 10, 16, 16, C.byref(t), 0, 1, 1, 1, 1, C.byref(abi.TilePlan())) != abi.DT_HIP_SUCCESS
