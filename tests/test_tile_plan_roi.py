"""dt_hip_plan_tiles_roi() / dt_hip_tile_rois_finalscale() -- the tile grid and tile regions of
_default_process_tiling_cl_roi() (src/develop/tiling.c:1076-1390) for finalscale -- against the Python restatement
(tests/tile_plan_roi.py) and against what a tiling must satisfy whatever the arithmetic: the good parts partition the
output, every tile's input region lies inside the input and is the module's own answer for the tile's output region."""
import ctypes as C
import itertools

import pytest

import tile_plan_roi as tr
from ansel_amd import abi, lib

CASES = []
for (w, h), scale, frac in itertools.product([(6000, 4000), (4001, 2999), (1504, 1000), (9504, 6336)],
                                             [0.5, 0.3333, 0.25, 0.61, 0.125], [4.0, 0.5, 0.2, 0.07]):
    CASES.append((w, h, scale, frac))


def _rois(w, h, scale):
    roi_out = tr.R(0, 0, int(w * scale + 0.5), int(h * scale + 0.5), scale)
    roi_in = tr.modify_roi_in(roi_out)
    roi_in.width, roi_in.height = min(roi_in.width, w), min(roi_in.height, h)
    return roi_in, roi_out


def _c(r):
    return abi.Roi.make(r.x, r.y, r.width, r.height, r.scale)


@pytest.mark.parametrize("w,h,scale,frac", CASES)
def test_roi_tile_plan_and_regions(w, h, scale, frac):
    l = lib.load()
    roi_in, roi_out = _rois(w, h, scale)
    t = abi.Tiling(1.0 + scale * scale, 1.0 + scale * scale, 1.0, 1.0, 0, 4, 1, 1)  # default_tiling_callback() + the resampler's support
    avail = int(roi_in.width * roi_in.height * 16 * t.factor_cl * frac)
    want = tr.plan(roi_in, roi_out, 16, 16, t, 0, avail, avail, 1 << 30, 1 << 30)
    got = abi.TilePlanRoi()
    ci, co = _c(roi_in), _c(roi_out)
    rc = l.dt_hip_plan_tiles_roi(C.byref(ci), C.byref(co), 16, 16, C.byref(t), 0, avail, avail, 1 << 30, 1 << 30, C.byref(got))
    assert rc == 0
    assert {k: getattr(got, k) for k in want} == want
    if frac < 1.0:
        assert got.tiles_x * got.tiles_y > 1
    covered = 0
    for tx in range(got.tiles_x):
        for ty in range(got.tiles_y):
            a, b, g = abi.Roi(), abi.Roi(), abi.Roi()
            rc = l.dt_hip_tile_rois_finalscale(C.byref(got), C.byref(ci), C.byref(co), tx, ty, C.byref(a), C.byref(b), C.byref(g))
            exp = tr.tile_rois(want, roi_in, roi_out, tx, ty)
            if exp is None:
                assert rc == abi.DT_HIP_TILE_EMPTY
                continue
            assert rc == 0
            for cr, pr in zip((a, b, g), exp):
                assert (cr.x, cr.y, cr.width, cr.height, cr.scale) == pr.tup()
            # whatever the arithmetic: the good part starts on the grid, lies inside the full output region, ...
            assert (g.x, g.y) == (tx * got.tile_wd, ty * got.tile_ht)
            assert b.x <= g.x and b.y <= g.y and b.x + b.width >= g.x + g.width and b.y + b.height >= g.y + g.height
            # ... whose input region is the module's own answer for it, inside the frame's input
            own = tr.modify_roi_in(tr.R(b.x, b.y, b.width, b.height, b.scale))
            assert a.x >= roi_in.x and a.y >= roi_in.y and a.x + a.width <= roi_in.x + roi_in.width
            assert a.y + a.height <= roi_in.y + roi_in.height
            assert (a.x, a.y) == (max(own.x, roi_in.x), max(own.y, roi_in.y))
            covered += g.width * g.height
    assert covered == roi_out.width * roi_out.height  # the good parts partition the output
