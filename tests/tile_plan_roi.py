"""TEST INFRASTRUCTURE: the tile grid and the per-tile regions of _default_process_tiling_cl_roi()
(src/develop/tiling.c:1076-1390) with finalscale's modify_roi_in() (src/iop/finalscale.c:76-107) restated in Python --
binary32 arithmetic where the C code has it, C's truncating int conversions -- to check dt_hip_plan_tiles_roi() /
dt_hip_tile_rois_finalscale() and to drive the oracle over the same tiles as dt_hip_default_process_tiling_roi()."""
import math

import numpy as np

f32 = np.float32


def _lcm(a, b):
    return a * b // math.gcd(a, b)


def _cmod(n, a):  # C's %: sign of the dividend
    return int(math.fmod(n, a))


def align_up(n, a):
    return n + a - _cmod(n, a)


def align_down(n, a):
    return n - _cmod(n, a)


def align_close(n, a):
    off = _cmod(n, a)
    return n + ((a - off) if off > a // 2 else -off)


def roundf(x):
    """roundf() of a value converted to binary32: half away from zero"""
    x = float(f32(x))
    return int(math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


class R:
    def __init__(self, x, y, w, h, scale):
        self.x, self.y, self.width, self.height, self.scale = int(x), int(y), int(w), int(h), float(scale)

    def copy(self):
        return R(self.x, self.y, self.width, self.height, self.scale)

    def tup(self):
        return (self.x, self.y, self.width, self.height, self.scale)


def modify_roi_in(roi_out):
    """finalscale.c:76-107, the full-resolution pipeline; scale is a double in dt_iop_roi_t"""
    ri = roi_out.copy()
    if ri.scale > float(f32(1.0)):
        ri.x = roundf(float(f32(ri.x)) / roi_out.scale)
        ri.y = roundf(float(f32(ri.y)) / roi_out.scale)
        ri.width = roundf(roi_out.width / roi_out.scale)
        ri.height = roundf(roi_out.height / roi_out.scale)
        ri.scale = 1.0
    else:
        ri.width = roundf(roi_out.width / roi_out.scale)
        ri.height = roundf(roi_out.height / roi_out.scale)
        ri.scale = 1.0
        resample = f32(roi_out.scale / ri.scale)
        ri.x = roundf(f32(ri.x) / resample)   # int / float -> binary32 division
        ri.y = roundf(f32(ri.y) / resample)
    return ri


def plan(roi_in, roi_out, in_bpp, out_bpp, t, filters, available, memalloc, max_w, max_h):
    max_bpp = max(in_bpp, out_bpp)
    fullscale = max(f32(roi_in.scale / roi_out.scale),
                    np.sqrt((f32(roi_in.width) * f32(roi_in.height)) / (f32(roi_out.width) * f32(roi_out.height))))
    delta = int(np.ceil(fullscale))
    inacc = 5 * delta
    factor = max(f32(t.factor_cl), f32(1))
    single = min(max((f32(available) - f32(t.overhead)) / factor, f32(0)), f32(memalloc))
    maxbuf = max(f32(t.maxbuf_cl), f32(1))
    width = min(max(roi_in.width, roi_out.width), max_w)
    height = min(max(roi_in.height, roi_out.height), max_h)
    al = _lcm(_lcm(t.xalign, t.yalign), 4 if filters != 9 else 1)

    def foot(w, h):
        return f32(w) * f32(h) * f32(max_bpp) * maxbuf

    if foot(width, height) > single:
        scale = single / foot(width, height)
        if width < height and scale >= f32(0.333):
            height = align_down(int(np.floor(f32(height) * scale)), al)
        elif height <= width and scale >= f32(0.333):
            width = align_down(int(np.floor(f32(width) * scale)), al)
        else:
            r = np.sqrt(scale)
            width, height = align_down(int(np.floor(f32(width) * r)), al), align_down(int(np.floor(f32(height) * r)), al)
    if 3 * t.overlap > width or 3 * t.overlap > height:
        width = height = align_down(int(np.floor(np.sqrt(f32(width) * f32(height)))), al)
    overlap_in = align_up(t.overlap, al)
    overlap_out = int(np.ceil(f32(overlap_in) / fullscale))
    while foot(width, height) > single:
        if width <= al and height <= al:
            break
        if width < height and height > al:
            height -= al
        elif width > al:
            width -= al
        else:
            height -= al
    if width < max(roi_in.width, roi_out.width):
        width = max(al, align_down(width, al))
    if height < max(roi_in.height, roi_out.height):
        height = max(al, align_down(height, al))

    def count(n_in, n_out, size):
        if n_in > n_out:
            return int(np.ceil(f32(n_in) / f32(max(size - 2 * overlap_in - inacc, 1)))) if size < n_in else 1
        return int(np.ceil(f32(n_out) / f32(max(size - 2 * overlap_out, 1)))) if size < n_out else 1

    tiles_x, tiles_y = count(roi_in.width, roi_out.width, width), count(roi_in.height, roi_out.height, height)
    tile_wd = align_up(roi_out.width // tiles_x if roi_out.width % tiles_x == 0 else roi_out.width // tiles_x + 1, al)
    tile_ht = align_up(roi_out.height // tiles_y if roi_out.height % tiles_y == 0 else roi_out.height // tiles_y + 1, al)
    return dict(width=width, height=height, tile_wd=tile_wd, tile_ht=tile_ht, tiles_x=tiles_x, tiles_y=tiles_y,
                overlap_in=overlap_in, overlap_out=overlap_out, delta=delta, xyalign=al)


def _clamp_into(r, outer):
    r.x = max(r.x, outer.x)
    r.y = max(r.y, outer.y)
    r.width = min(r.width, outer.width + outer.x - r.x)
    r.height = min(r.height, outer.height + outer.y - r.y)


def tile_rois(p, roi_in, roi_out, tx, ty):
    """:1228-1300 -> (iroi_full, oroi_full, oroi_good) or None for a tile without output pixels"""
    tw, th, al, delta, ov = p["tile_wd"], p["tile_ht"], p["xyalign"], p["delta"], p["overlap_in"]
    wd = roi_out.width - tx * tw if (tx + 1) * tw > roi_out.width else tw
    ht = roi_out.height - ty * th if (ty + 1) * th > roi_out.height else th
    if wd <= 0 or ht <= 0:
        return None
    oroi_good = R(roi_out.x + tx * tw, roi_out.y + ty * th, wd, ht, roi_out.scale)
    iroi_good = modify_roi_in(oroi_good)
    _clamp_into(iroi_good, roi_in)
    nx = max(align_close(iroi_good.x - ov - delta, al), roi_in.x)
    ny = max(align_close(iroi_good.y - ov - delta, al), roi_in.y)
    nw = min(align_up(iroi_good.width + ov + delta + (iroi_good.x - nx), al), roi_in.width + roi_in.x - nx)
    nh = min(align_up(iroi_good.height + ov + delta + (iroi_good.y - ny), al), roi_in.height + roi_in.y - ny)
    iroi_full = R(nx, ny, nw, nh, iroi_good.scale)
    oroi_full = oroi_good.copy()
    probe = modify_roi_in(oroi_full)
    it = 10
    while (abs(probe.x - iroi_full.x) > delta or abs(probe.y - iroi_full.y) > delta or abs(probe.width - iroi_full.width) > delta
           or abs(probe.height - iroi_full.height) > delta) and it > 0:
        k = oroi_full.scale / iroi_full.scale
        oroi_full.x = int(oroi_full.x + (iroi_full.x - probe.x) * k)           # int += double: truncation toward zero
        oroi_full.y = int(oroi_full.y + (iroi_full.y - probe.y) * k)
        oroi_full.width = int(oroi_full.width + (iroi_full.width - probe.width) * k)
        oroi_full.height = int(oroi_full.height + (iroi_full.height - probe.height) * k)
        probe = modify_roi_in(oroi_full)
        it -= 1
    assert it > 0
    oroi_full.x = min(oroi_full.x, oroi_good.x)
    oroi_full.y = min(oroi_full.y, oroi_good.y)
    oroi_full.width = max(oroi_full.width, oroi_good.x + oroi_good.width - oroi_full.x)
    oroi_full.height = max(oroi_full.height, oroi_good.y + oroi_good.height - oroi_full.y)
    _clamp_into(oroi_full, roi_out)
    iroi_full = modify_roi_in(oroi_full)
    _clamp_into(iroi_full, roi_in)
    return iroi_full, oroi_full, oroi_good
