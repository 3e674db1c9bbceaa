"""TEST INFRASTRUCTURE: an oracle-backed stand-in for ansel_amd.tiled.HipBandEngine, so that the
multi-rank driver (planner, count all-reduce, halo send/recv) can be exercised on CPU with gloo.

It follows the same begin / exchange / finish protocol on numpy buffers.  The demosaic of a band is
taken from the oracle's whole-frame RCD run on a frame that is zero outside the band's rows + halo:
the band's output rows only depend on the frame's tile rows [tile_row0, tile_row1), whose input
rows are exactly band + halo, so the rows are those of the real whole-frame result."""
import copy
import ctypes as C

import numpy as np
import torch

import checkers as ck
from ansel_amd import abi
from ansel_amd import lib as hiplib
from ansel_amd.tiled import BandRequest, BandWork

CFA_OPS = ("rawprepare", "temperature", "highlights")
STENCIL_OPS = ("denoiseprofile", "diffuse", "nlmeans")


def halo_rows(node):
    """dt_hip_band_halo_rows(): a pure host function of libansel_hip.so"""
    l = hiplib.load()
    return l.dt_hip_band_halo_rows(node.op.encode(), C.byref(node.piece), C.cast(C.byref(node.data), C.c_void_p),
                                   C.sizeof(node.data))


def _band_piece(piece, band):
    p = copy.deepcopy(piece)
    p.roi_in.y += band.row0
    p.roi_in.height = band.rows
    p.roi_out.y += band.row0
    p.roi_out.height = band.rows
    return p


class OracleBandEngine:
    def __init__(self, nodes, width, height):
        self.nodes = nodes
        self.w, self.h = width, height
        self.l = ck.oracle()
        assert self.l is not None, "oracle/liboracle.so missing: make -C oracle oracle"

    def begin(self, band, raw_band, width):
        src = np.ascontiguousarray(raw_band)
        rows, w = band.rows, self.w
        count = None
        unclipped = None
        for n in self.nodes:
            if n.op not in CFA_OPS:
                break
            dst = np.zeros((rows, w), np.float32)
            if n.op == "rawprepare":
                p = copy.deepcopy(n.piece)
                d = copy.deepcopy(n.data)
                p.roi_out.y += d.y + band.row0
                d.y = 0
                p.roi_in.height = rows
                p.roi_out.height = rows
                assert ck.call(self.l, "oracle_rawprepare", p, d, src, dst) == 0
            elif n.op == "temperature":
                assert ck.call(self.l, "oracle_temperature", _band_piece(n.piece, band), n.data, src, dst) == 0
            else:
                # highlights.c:717-733 with the bypass decision deferred to finish()
                pm = [np.float32(v) if v > 0 else np.float32(1) for v in list(n.piece.processed_maximum)[:3]]
                clip = np.float32(n.data.clip) * min(pm)
                unclipped = src.copy()
                count = torch.tensor([int((src > clip).sum())], dtype=torch.int64)
                dst = np.where(clip < src, clip, src).astype(np.float32)
            src = dst
        halo = torch.zeros((band.halo_top + rows + band.halo_bottom, w), dtype=torch.float32)
        halo[band.halo_top:band.halo_top + rows] = torch.from_numpy(src)
        return BandWork(halo, count, unclipped)

    def resolve(self, band, work):
        if work.count is not None and int(work.count.item()) < 25:
            work.halo.numpy()[band.halo_top:band.halo_top + band.rows] = work.token

    def finish(self, band, work, out_band):
        """resumable like dt_hip_pipe_band_finish(): a BandRequest to serve, or None when the band is done"""
        if getattr(work, "walk", None) is None:
            work.walk = self._walk(band, work, out_band)
        try:
            work.req = next(work.walk)
            return work.req
        except StopIteration:
            return None

    def relay(self, band, work):
        work.relay_frame = work.req.relay
        work.relay_frame.numpy().reshape(self.h, self.w, 4)[band.row0:band.row0 + band.rows] = work.relay_rows

    def _stencil(self, n, band, src):
        """a stencil module on a band.  The module input of the band + the halo rows the driver fetched go into
        a frame that is zero elsewhere, the oracle runs on that frame and the band's rows are cut out: they only
        depend on rows within the halo, so they are the rows of the real whole-frame result -- if and only if the
        halo dt_hip_band_halo_rows() asks for is large enough, which is what this checks on CPU.  The profiled
        wavelets also need frame-wide sums; this stand-in gets them the blunt way, by all-reducing the module
        input itself (own rows filled, zero elsewhere: the sum is the frame, exactly)."""
        rows, w, h = band.rows, self.w, self.h
        hr = halo_rows(n)
        if hr < 0:
            return src
        top, bottom = min(hr, band.row0), min(hr, h - band.row0 - rows)
        hb = torch.zeros((top + rows + bottom, w * 4), dtype=torch.float32)
        hb[top:top + rows] = torch.from_numpy(src.reshape(rows, w * 4))
        if top or bottom:
            yield BandRequest(hb, hr, None)
        frame = np.zeros((h, w, 4), np.float32)
        frame[band.row0 - top:band.row0 + rows + bottom] = hb.numpy().reshape(-1, w, 4)
        if n.op == "denoiseprofile" and n.data.mode == abi.DT_HIP_DENOISEPROFILE_WAVELETS and rows < h:
            f64 = torch.zeros((h, w, 4), dtype=torch.float64)
            f64[band.row0:band.row0 + rows] = torch.from_numpy(src.astype(np.float64))
            f64 = f64.reshape(-1)
            yield BandRequest(None, 0, f64, sum_planes=1)  # one table of [frame rows][4 w] doubles: the all-gather path
            frame = f64.numpy().reshape(h, w, 4).astype(np.float32)
        full = np.zeros((h, w, 4), np.float32)
        assert ck.call(self.l, "oracle_" + n.op, n.piece, n.data, frame, full) == 0, n.op
        return np.ascontiguousarray(full[band.row0:band.row0 + rows])

    def _walk(self, band, work, out_band):
        rows, w, h = band.rows, self.w, self.h
        cfa = work.halo.numpy()
        frame = np.zeros((h, w), np.float32)
        r0 = band.row0 - band.halo_top
        frame[r0:r0 + cfa.shape[0]] = cfa
        src = held = None
        for n in self.nodes:
            if n.op in CFA_OPS:
                continue
            if n.op == "demosaic":
                full = np.zeros((h, w, 4), np.float32)
                assert ck.call(self.l, "oracle_demosaic", n.piece, n.data, frame, full) == 0
                src = np.ascontiguousarray(full[band.row0:band.row0 + rows])
            elif n.op == "export_u16":
                self.l.oracle_export_convert_u16(w, rows, ck.ptr(src), ck.ptr(out_band))
                return
            elif n.op == "blend":
                # dt_develop_blend_process(): blend(input of the module, output of the module) in place
                assert ck.call(self.l, "oracle_develop_blend", _band_piece(n.piece, band), n.data,
                               np.ascontiguousarray(held), src) == 0
            elif n.op == "bilat":
                # the relay stop, the blunt way: the frame of module inputs is what travels from band to band (each
                # adds its rows), the oracle runs on the frame the last band broadcasts
                work.relay_rows = src
                yield BandRequest(None, 0, None, torch.zeros((h * w * 4,), dtype=torch.float32))
                frame_in = work.relay_frame.numpy().reshape(h, w, 4)
                full = np.zeros((h, w, 4), np.float32)
                assert ck.call(self.l, "oracle_bilat", n.piece, n.data, np.ascontiguousarray(frame_in), full) == 0
                held = src
                src = np.ascontiguousarray(full[band.row0:band.row0 + rows])
            elif n.op in STENCIL_OPS:
                held = src
                src = yield from self._stencil(n, band, src)
            else:
                dst = np.zeros((rows, w, 4), np.float32)
                assert ck.call(self.l, "oracle_" + n.op, _band_piece(n.piece, band), n.data, src, dst) == 0, n.op
                held = src
                src = dst
        out_band[...] = src


def whole_frame(nodes, raw, w, h):
    """the unsplit oracle chain"""
    l = ck.oracle()
    src = raw
    prev = None
    for n in nodes:
        if n.op == "blend":
            assert ck.call(l, "oracle_develop_blend", n.piece, n.data, np.ascontiguousarray(prev), src) == 0
            continue
        if n.op == "export_u16":
            out = np.zeros((h, w, 4), np.uint16)
            l.oracle_export_convert_u16(w, h, ck.ptr(src), ck.ptr(out))
            return out
        dst = np.zeros((h, w) if n.op in CFA_OPS else (h, w, 4), np.float32)
        assert ck.call(l, "oracle_" + n.op, n.piece, n.data, np.ascontiguousarray(src), dst) == 0, n.op
        prev = src
        src = dst
    return src


def test_frame(w, h, n_over_top, n_over_bottom, seed=3):
    """a mosaic without clipped photosites except n_over_top in the top half and n_over_bottom in the
    bottom half (to steer the highlights bypass across bands)"""
    from ansel_amd import synth
    rng = np.random.default_rng(seed)
    raw = rng.integers(synth.BLACK + 50, synth.BLACK + 3000, size=(h, w)).astype(np.uint16)
    ys = rng.integers(8, h // 2 - 8, size=n_over_top)
    xs = rng.integers(8, w - 8, size=n_over_top)
    raw[ys, xs] = 65535
    ys = rng.integers(h // 2 + 8, h - 8, size=n_over_bottom)
    xs = rng.integers(8, w - 8, size=n_over_bottom)
    raw[ys, xs] = 65535
    return raw
