"""CPU: the multi-GPU row-band path (ansel_amd/tiled.py, include/ansel_hip.h section 3b) without a GPU.

  * dt_hip_plan_bands() is a pure host function of libansel_hip.so: partition + halo properties
  * world_size 2 over gloo: the real driver (begin -> all-reduce + send/recv -> finish) with the
    oracle-backed engine of tests/band_engine.py; the gathered bands must equal the unsplit
    oracle chain bit for bit, including the highlights bypass that depends on the count of the
    WHOLE frame (20 clipped photosites: 10 per band -> bypass; 40: 20 per band -> clip)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import band_engine as be  # noqa: E402
import checkers as ck  # noqa: E402
from ansel_amd import abi, filmic, lib, params, pipe, tiled  # noqa: E402

needs_oracle = pytest.mark.skipif(ck.oracle() is None, reason="oracle/liboracle.so not built")


@pytest.mark.parametrize("w,h", [(11648, 8736), (9504, 6336), (640, 480), (400, 300)])
@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_rcd_bands_partition_the_frame_on_tile_rows(w, h, n):
    num_vertical = 1 + (h - 19) // 94
    if num_vertical < n:
        with pytest.raises(Exception):
            tiled.plan_bands(w, h, n)
        return
    bands = tiled.plan_bands(w, h, n)
    assert bands[0].row0 == 0 and bands[-1].row0 + bands[-1].rows == h
    assert bands[0].tile_row0 == 0 and bands[-1].tile_row1 == num_vertical
    for k, b in enumerate(bands):
        if k:
            assert b.row0 == bands[k - 1].row0 + bands[k - 1].rows
            assert b.tile_row0 == bands[k - 1].tile_row1
            assert b.row0 == 94 * b.tile_row0 + 9 and b.halo_top == 9
        else:
            assert b.halo_top == 0
        # the band's buffer holds every input row its tile rows read (rcd.c:296-300)
        assert b.row0 - b.halo_top == 94 * b.tile_row0
        assert b.row0 + b.rows + b.halo_bottom == min(94 * (b.tile_row1 - 1) + 112, h)
        assert b.rows >= 9  # a neighbour's halo never spans more than one band


@pytest.mark.parametrize("w,h,n", [(1504, 1000, 2), (1504, 1000, 5), (600, 400, 3), (11648, 8736, 8), (500, 144, 2)])
def test_amaze_bands_partition_the_frame_on_tile_rows(w, h, n):
    """with the AMaZE demosaic a band owns whole 128-row tile rows of the frame's own grid and asks either neighbour for the 16
    mosaic rows its tiles read beyond them (fewer where the frame ends first)"""
    bands = tiled.plan_bands(w, h, n, abi.DT_HIP_DEMOSAIC_AMAZE)
    tile_rows = (h + 127) // 128
    assert bands[0].row0 == 0 and bands[0].halo_top == 0 and bands[-1].halo_bottom == 0
    assert bands[-1].row0 + bands[-1].rows == h and bands[0].tile_row0 == 0 and bands[-1].tile_row1 == tile_rows
    for k, b in enumerate(bands):
        assert b.rows > 0 and b.row0 == 128 * b.tile_row0 and b.tile_row1 > b.tile_row0
        if k:
            assert b.row0 == bands[k - 1].row0 + bands[k - 1].rows and b.tile_row0 == bands[k - 1].tile_row1 and b.halo_top == 16
        if k + 1 < n:
            assert b.halo_bottom == min(16, h - b.row0 - b.rows)
        # what the band's tiles read of the mosaic is in its own rows + halo
        first = max(128 * b.tile_row0 - 16, 0)
        last = min(128 * b.tile_row1 + 16, h)
        assert b.row0 - b.halo_top <= first and b.row0 + b.rows + b.halo_bottom >= last


def test_amaze_bands_need_a_tile_row_each():
    with pytest.raises(lib.AnselHipError):
        tiled.plan_bands(600, 400, 5, abi.DT_HIP_DEMOSAIC_AMAZE)  # 4 tile rows


@pytest.mark.parametrize("w,h", [(517, 389), (6024, 4000), (600, 392)])
def test_amaze_frames_with_tiles_of_the_first_kernel_have_no_band_plan(w, h):
    """only the on-chip AMaZE kernel walks a band; a frame that keeps tiles in the first kernel's body -- a last tile column of
    odd width, width or height 1..15 past a multiple of 128 (a mirrored strip running past its plane) -- is refused by the PLAN,
    before any band has run a stage (it used to surface inside dt_hip_pipe_band_finish())"""
    with pytest.raises(lib.AnselHipError, match="no band mode"):
        tiled.plan_bands(w, h, 2, abi.DT_HIP_DEMOSAIC_AMAZE)


def test_bands_without_demosaic_are_two_row_aligned():
    bands = tiled.plan_bands(640, 486, 4, demosaic_method=-1)
    assert sum(b.rows for b in bands) == 486
    assert all(b.row0 % 2 == 0 and b.halo_top == 0 and b.halo_bottom == 0 for b in bands)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _nodes(w, h, lut, demosaic_method=abi.DT_HIP_DEMOSAIC_RCD):
    coeffs = params.unbounded_coeffs(lut)
    return pipe.light_pipe_nodes(w, h, lut.ctypes.data, float(lut[0]), coeffs, with_filmic=True,
                                 filmic=filmic.default_data(), demosaic_method=demosaic_method)


def _rank_main(rank, world, port, w, h, n_top, n_bottom, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lut = params.srgb_encode_lut()
        nodes = _nodes(w, h, lut)
        raw = be.test_frame(w, h, n_top, n_bottom)
        bands = tiled.plan_bands(w, h, world, tiled.pipe_demosaic_method(nodes))
        b = bands[rank]
        engine = be.OracleBandEngine(nodes, w, h)
        out = np.zeros((b.rows, w, 4), np.uint16)
        tiled.process_band(engine, bands, rank, raw[b.row0:b.row0 + b.rows], out, w, dist=dist)
        np.save(os.path.join(outdir, "band%d.npy" % rank), out)
    finally:
        dist.destroy_process_group()


@needs_oracle
@pytest.mark.parametrize("n_top,n_bottom", [(10, 10), (20, 20)])
def test_two_ranks_over_gloo_equal_the_unsplit_frame(tmp_path, n_top, n_bottom):
    import torch.multiprocessing as mp
    w, h, world = 256, 400, 2
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, w, h, n_top, n_bottom, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("band%d.npy" % r)) for r in range(world)], axis=0)
    lut = params.srgb_encode_lut()
    nodes = _nodes(w, h, lut)
    raw = be.test_frame(w, h, n_top, n_bottom)
    want = be.whole_frame(nodes, raw, w, h)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    # the two cases really differ in the bypass decision: each band alone sees < 25 clipped photosites
    clipped_frame = int((raw == 65535).sum())
    assert (clipped_frame < 25) == (n_top + n_bottom < 25)


@needs_oracle
@pytest.mark.parametrize("n_top,n_bottom", [(4, 30), (12, 8)])
def test_local_protocol_matches_the_unsplit_frame_for_three_bands(n_top, n_bottom):
    """same protocol, all bands in one process (the data motion the single-GPU test uses)"""
    w, h, n = 256, 500, 3
    lut = params.srgb_encode_lut()
    nodes = _nodes(w, h, lut)
    raw = be.test_frame(w, h, n_top, n_bottom)
    bands = tiled.plan_bands(w, h, n)
    engine = be.OracleBandEngine(nodes, w, h)
    outs = [np.zeros((b.rows, w, 4), np.uint16) for b in bands]
    tiled.process_bands_locally(engine, bands, [raw[b.row0:b.row0 + b.rows] for b in bands], outs, w)
    assert np.array_equal(np.concatenate(outs, axis=0), be.whole_frame(nodes, raw, w, h))


@needs_oracle
@pytest.mark.parametrize("w,h,n", [(256, 500, 3), (300, 330, 2)])
def test_local_protocol_with_the_amaze_demosaic(w, h, n):
    """the same with AMaZE: bands of whole 128-row tile rows, 16 halo rows; each band demosaics a frame that is zero beyond
    its rows and halo, so the assembled frame equals the unsplit one only if that halo is what its tiles read"""
    lut = params.srgb_encode_lut()
    nodes = _nodes(w, h, lut, abi.DT_HIP_DEMOSAIC_AMAZE)
    raw = be.test_frame(w, h, 12, 8)
    bands = tiled.plan_bands(w, h, n, tiled.pipe_demosaic_method(nodes))
    assert bands[1].halo_top == 16
    engine = be.OracleBandEngine(nodes, w, h)
    outs = [np.zeros((b.rows, w, 4), np.uint16) for b in bands]
    tiled.process_bands_locally(engine, bands, [raw[b.row0:b.row0 + b.rows] for b in bands], outs, w)
    assert np.array_equal(np.concatenate(outs, axis=0), be.whole_frame(nodes, raw, w, h))


# ---- the stencil modules of the full pipe on row bands (config 4 of BASELINE.json) ----------------------------
def _full_nodes(w, h, lut, which):
    coeffs = params.unbounded_coeffs(lut)
    nodes = pipe.denoise_pipe_nodes(w, h, lut.ctypes.data, float(lut[0]), coeffs, filmic=filmic.default_data(),
                                    diffuse_iterations=2, with_nlmeans=True, with_bilat=which in ("bilat", "everything"))
    drop = {"wavelets": ("diffuse", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "diffuse": ("denoiseprofile", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "diffuse_inpaint": ("denoiseprofile", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "nlmeans": ("denoiseprofile", "diffuse"),
            "dn_nlmeans": ("diffuse", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "bilat": ("denoiseprofile", "diffuse", "nlmeans"), "blended": (), "all": (), "everything": ()}[which]
    nodes = [n for n in nodes if n.op not in drop]
    if which == "blended":
        # blends on a pointwise module (uniform) and on two stencil modules (parametric masks, tone curve); the last
        # node of the RGBA part is itself blended
        nodes = [n for n in nodes if n.op not in ("nlmeans", "rgb_to_lab", "lab_to_rgb")]
        out = []
        for n in nodes:
            out.append(n)
            if n.op == "exposure":
                out.append(pipe.Node("blend", abi.BlendData.uniform(params.WORK_IN, 60.0, abi.BLEND_MULTIPLY, 0.5), n.piece))
            if n.op in ("denoiseprofile", "diffuse"):
                d = abi.BlendData.uniform(params.WORK_IN, 80.0)
                d.channel(abi.BLENDIF_GRAY_in, 0.02, 0.15, 0.6, 0.9, boost=1.0)
                d.channel(abi.BLENDIF_Jz_in, 0.05, 0.2, 1.0, 1.0, boost=-4.0)
                d.channel(abi.BLENDIF_hz_out, 0.1, 0.3, 0.8, 0.95)
                d.contrast, d.brightness = 0.3, -0.2
                out.append(pipe.Node("blend", d, n.piece))
        return out
    if which == "diffuse_inpaint":
        # threshold > 0: the inpainting noise is keyed on the pixel's position in the FRAME
        for n in nodes:
            if n.op == "diffuse":
                n.data = params.diffuse("inpaint_highlights", iterations=2, threshold=0.05)
    if which == "dn_nlmeans":
        for n in nodes:
            if n.op == "denoiseprofile":
                n.data = params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS)
    return nodes


def test_halo_rows_of_the_stencil_modules():
    """dt_hip_band_halo_rows(): pure host function"""
    lut = params.srgb_encode_lut()
    by_op = {n.op: n for n in _full_nodes(752, 2000, lut, "all")}
    # the profiled wavelets take 2 rows of their input (scale 0 reads +-2) and then 2 * 2^k rows of their own coarse
    # plane before scale k: no row is computed twice
    assert be.halo_rows(by_op["denoiseprofile"]) == 2
    # patch radius 2 + 1 + search radius 7, + the 60-row-ish chunk the band boundary may cut (nlmeans_core.c:264-295)
    assert 10 + 50 <= be.halo_rows(by_op["nlmeans"]) <= 10 + 70
    it, scales = 2, oracle_scales(by_op["diffuse"])
    assert be.halo_rows(by_op["diffuse"]) == it * 3 * ((1 << scales) - 1)
    assert be.halo_rows(by_op["colorin"]) == 0
    # a frame the wavelets pass through untouched (denoiseprofile.c:1325-1329)
    small = {n.op: n for n in _full_nodes(100, 2000, lut, "all")}  # 100 columns < 2 x the coarsest dilation 64
    assert be.halo_rows(small["denoiseprofile"]) == -1


def oracle_scales(node):
    import ctypes as C
    l = ck.oracle()
    if l is None:
        pytest.skip("oracle/liboracle.so not built")
    l.oracle_diffuse_scales.restype = C.c_int
    return l.oracle_diffuse_scales(C.byref(node.piece), C.byref(node.data))


def _full_rank_main(rank, world, port, w, h, which, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "8"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lut = params.srgb_encode_lut()
        nodes = _full_nodes(w, h, lut, which)
        raw = be.test_frame(w, h, 10, 10)
        bands = tiled.plan_bands(w, h, world, tiled.pipe_demosaic_method(nodes))
        b = bands[rank]
        engine = be.OracleBandEngine(nodes, w, h)
        out = np.zeros((b.rows, w, 4), np.uint16)
        tiled.process_band(engine, bands, rank, raw[b.row0:b.row0 + b.rows], out, w, dist=dist)
        np.save(os.path.join(outdir, "band%d.npy" % rank), out)
    finally:
        dist.destroy_process_group()


@needs_oracle
@pytest.mark.parametrize("which", ["wavelets", "diffuse", "diffuse_inpaint", "nlmeans", "dn_nlmeans", "blended", "all", "bilat",
                                   "everything"])
def test_two_ranks_full_pipe_over_gloo_equal_the_unsplit_frame(tmp_path, which):
    """halo send/recv of float4 rows + the all-reduce of the wavelets, and the halo sizes themselves: a band cut
    from a frame that is zero beyond the halo equals the rows of the real frame only if the halo is enough"""
    import torch.multiprocessing as mp
    w, h, world = 128, 400, 2
    port = _free_port()
    mp.spawn(_full_rank_main, args=(world, port, w, h, which, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("band%d.npy" % r)) for r in range(world)], axis=0)
    lut = params.srgb_encode_lut()
    want = be.whole_frame(_full_nodes(w, h, lut, which), be.test_frame(w, h, 10, 10), w, h)
    assert np.array_equal(got, want)


@needs_oracle
def test_three_ranks_full_pipe_over_gloo_equal_the_unsplit_frame(tmp_path):
    """world size 3: a middle band with two neighbours -- halo rows from both sides, its segment of the wavelets' table of
    partial sums gathered from and to two peers, a relay turn that neither starts nor ends the chain"""
    import torch.multiprocessing as mp
    w, h, world = 128, 640, 3
    port = _free_port()
    mp.spawn(_full_rank_main, args=(world, port, w, h, "everything", str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("band%d.npy" % r)) for r in range(world)], axis=0)
    lut = params.srgb_encode_lut()
    want = be.whole_frame(_full_nodes(w, h, lut, "everything"), be.test_frame(w, h, 10, 10), w, h)
    assert np.array_equal(got, want)


def test_the_sums_table_is_gathered_not_reduced():
    """gather_sums(): every band's own row segment of every plane reaches every band; what lies outside a band's rows
    in ITS table is never read (poisoned here), which an all-reduce would have added in"""
    import torch

    class _Dist:  # a one-process stand-in: rank r's call sees all three stages
        def __init__(self, stages):
            self.stages = stages

        def all_gather(self, parts, stage, group=None):
            for p, s in zip(parts, self.stages):
                p.copy_(s)
    bands = tiled.plan_bands(128, 640, 3)
    height, planes, per_row = 640, 2, 8
    truth = torch.arange(planes * height * per_row, dtype=torch.float64).view(planes, -1)
    longest = max(b.rows for b in bands) * per_row
    stages = []
    for b in bands:
        st = torch.zeros((planes, longest), dtype=torch.float64)
        st[:, :b.rows * per_row] = truth[:, b.row0 * per_row:(b.row0 + b.rows) * per_row]
        stages.append(st)
    for r, b in enumerate(bands):
        mine = torch.full((planes, height * per_row), float("nan"), dtype=torch.float64)
        mine[:, b.row0 * per_row:(b.row0 + b.rows) * per_row] = truth[:, b.row0 * per_row:(b.row0 + b.rows) * per_row]
        req = tiled.BandRequest(None, 0, mine.view(-1), sum_planes=planes)
        tiled.gather_sums(req, bands, r, _Dist(stages))
        assert torch.equal(mine, truth)


@needs_oracle
@pytest.mark.parametrize("which", ["all", "everything"])
def test_local_protocol_full_pipe_three_bands(which):
    w, h, n = 128, 480, 3
    lut = params.srgb_encode_lut()
    nodes = _full_nodes(w, h, lut, which)
    raw = be.test_frame(w, h, 4, 30)
    bands = tiled.plan_bands(w, h, n)
    engine = be.OracleBandEngine(nodes, w, h)
    outs = [np.zeros((b.rows, w, 4), np.uint16) for b in bands]
    tiled.process_bands_locally(engine, bands, [raw[b.row0:b.row0 + b.rows] for b in bands], outs, w)
    assert np.array_equal(np.concatenate(outs, axis=0), be.whole_frame(nodes, raw, w, h))


def test_a_neighbour_thinner_than_the_halo_is_refused():
    class _Dist:
        pass
    bands = tiled.plan_bands(128, 400, 4)
    req = tiled.BandRequest(halo=object(), h=200, sums=None)
    with pytest.raises(Exception, match="fewer than"):
        tiled.serve_request(req, bands, 1, dist=_Dist())
