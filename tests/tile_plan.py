"""TEST INFRASTRUCTURE: the tile plan and tile loop of _default_process_tiling_cl_ptp() (src/develop/tiling.c:842-1067)
restated in Python -- binary32 arithmetic where the C code has it -- to check dt_hip_plan_tiles_ptp() and to drive the
oracle over the same tiles as dt_hip_default_process_tiling_ptp()."""
import copy
import math

import numpy as np

f32 = np.float32


def _lcm(a, b):
    return a * b // math.gcd(a, b)


def plan(roi_w, roi_h, in_bpp, out_bpp, t, filters, available, memalloc, max_w, max_h):
    """:868-979; t = abi.Tiling.  Returns dict(width, height, tile_wd, tile_ht, tiles_x, tiles_y, overlap)"""
    max_bpp = max(in_bpp, out_bpp)
    factor = max(f32(t.factor_cl), f32(1))
    single = min(max((f32(available) - f32(t.overhead)) / factor, f32(0)), f32(memalloc))
    maxbuf = max(f32(t.maxbuf_cl), f32(1))
    width, height = min(roi_w, max_w), min(roi_h, max_h)

    def foot(w, h):
        return f32(w) * f32(h) * f32(max_bpp) * maxbuf

    if foot(width, height) > single:
        scale = single / foot(width, height)
        if width < height and scale >= f32(0.333):
            height = int(np.floor(f32(height) * scale))
        elif height <= width and scale >= f32(0.333):
            width = int(np.floor(f32(width) * scale))
        else:
            r = np.sqrt(scale)
            width, height = int(np.floor(f32(width) * r)), int(np.floor(f32(height) * r))
    if 3 * t.overlap > width or 3 * t.overlap > height:
        width = height = int(np.floor(np.sqrt(f32(width) * f32(height))))
    xyalign = _lcm(t.xalign, t.yalign)
    walign = _lcm(xyalign, 4 if filters != 9 else 1)
    halign = xyalign
    if width < roi_w:
        width = (width // walign) * walign
    if height < roi_h:
        height = (height // halign) * halign
    while foot(width, height) > single:
        if width <= walign and height <= halign:
            break
        if width < height and height > halign:
            height -= halign
        elif width > walign:
            width -= walign
        else:
            height -= halign
    if width < roi_w:
        width = max(walign, width - width % walign)
    if height < roi_h:
        height = max(halign, height - height % halign)
    overlap = (t.overlap // xyalign + 1) * xyalign if t.overlap % xyalign else t.overlap
    tile_wd = width - 2 * overlap if width - 2 * overlap > 0 else 1
    tile_ht = height - 2 * overlap if height - 2 * overlap > 0 else 1
    tiles_x = int(np.ceil(f32(roi_w) / f32(tile_wd))) if width < roi_w else 1
    tiles_y = int(np.ceil(f32(roi_h) / f32(tile_ht))) if height < roi_h else 1
    return dict(width=width, height=height, tile_wd=tile_wd, tile_ht=tile_ht, tiles_x=tiles_x, tiles_y=tiles_y,
                overlap=overlap)


def tiles(p, roi_w, roi_h):
    """the tile loop of :981-1040: yields (x0, y0, wd, ht, ox, oy) -- tile origin and size in the frame, offset of the
    part that is kept inside the tile"""
    for tx in range(p["tiles_x"]):
        for ty in range(p["tiles_y"]):
            wd = roi_w - tx * p["tile_wd"] if tx * p["tile_wd"] + p["width"] > roi_w else p["width"]
            ht = roi_h - ty * p["tile_ht"] if ty * p["tile_ht"] + p["height"] > roi_h else p["height"]
            if (wd <= 2 * p["overlap"] and tx > 0) or (ht <= 2 * p["overlap"] and ty > 0):
                continue
            yield (tx * p["tile_wd"], ty * p["tile_ht"], wd, ht, p["overlap"] if tx > 0 else 0, p["overlap"] if ty > 0 else 0)


def run_tiled(call, piece, p, src, out):
    """call(piece_tile, in_tile, out_tile) per tile; src / out are full frames (h, w[, c])"""
    h, w = src.shape[:2]
    for x0, y0, wd, ht, ox, oy in tiles(p, w, h):
        pt = copy.deepcopy(piece)
        pt.roi_in.x += x0
        pt.roi_in.y += y0
        pt.roi_out.x += x0
        pt.roi_out.y += y0
        pt.roi_in.width = pt.roi_out.width = wd
        pt.roi_in.height = pt.roi_out.height = ht
        o = np.zeros((ht, wd) + out.shape[2:], out.dtype)
        call(pt, np.ascontiguousarray(src[y0:y0 + ht, x0:x0 + wd]), o)
        out[y0 + oy:y0 + ht, x0 + ox:x0 + wd] = o[oy:, ox:]
