"""-m gpu: the row-band path on the device (include/ansel_hip.h section 3b, ansel_amd/tiled.py).

All bands of a frame run one after the other on the one GPU of the test box, through the same
begin / resolve / finish entry points a multi-GPU job uses, with the two collectives replaced by
tensor copies.  The assembled bands must equal the unsplit executor run bit for bit -- including
the RCD tile grid (a band runs the frame's own tile rows) and the highlights bypass that depends
on the clipped count of the whole frame."""
import os

import numpy as np
import pytest

import band_engine as be
import hipcheck as hc
from ansel_amd import filmic, lib, params, pipe, synth, tiled

pytestmark = pytest.mark.gpu


def _setup():
    hc.hip()
    import torch
    lut = params.srgb_encode_lut()
    d_lut = torch.from_numpy(lut).to("cuda:0")
    return torch, lut, d_lut


def _nodes(w, h, d_lut, lut, demosaic_method=None):
    from ansel_amd import abi
    return pipe.light_pipe_nodes(w, h, d_lut.data_ptr(), float(lut[0]), params.unbounded_coeffs(lut),
                                 with_filmic=True, filmic=filmic.default_data(),
                                 demosaic_method=abi.DT_HIP_DEMOSAIC_RCD if demosaic_method is None else demosaic_method)


def _whole(torch, nodes, raw, w, h, fusion):
    p = pipe.DevicePipe(0, nodes, fusion=fusion)
    d_in = torch.from_numpy(raw.view(np.int16)).to("cuda:0")
    d_out = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda:0")
    p.process(d_in.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    p.close()
    return d_out.cpu().numpy().view(np.uint16)


def _banded(torch, nodes, raw, w, h, n, fusion):
    p = pipe.DevicePipe(0, nodes, fusion=fusion)
    engine = tiled.HipBandEngine(p, "cuda:0")
    bands = tiled.plan_bands(w, h, n, tiled.pipe_demosaic_method(nodes))
    ins = [torch.from_numpy(np.ascontiguousarray(raw[b.row0:b.row0 + b.rows]).view(np.int16)).to("cuda:0") for b in bands]
    outs = [torch.zeros((b.rows, w, 4), dtype=torch.int16, device="cuda:0") for b in bands]
    tiled.process_bands_locally(engine, bands, [t.data_ptr() for t in ins], [t.data_ptr() for t in outs], w)
    torch.cuda.synchronize()
    p.close()
    return np.concatenate([t.cpu().numpy().view(np.uint16) for t in outs], axis=0)


@pytest.mark.parametrize("fusion", [True, False])
# 402 x 640: a width that is not a multiple of 4 puts the own rows of bands k > 0 (behind 9 halo rows) off the 16-byte grid
@pytest.mark.parametrize("w,h,n", [(1504, 1000, 2), (1504, 1000, 5), (400, 300, 3), (752, 2000, 8), (402, 640, 2), (402, 640, 3)])
def test_bands_equal_the_unsplit_frame(w, h, n, fusion):
    torch, lut, d_lut = _setup()
    nodes = _nodes(w, h, d_lut, lut)
    raw = synth.bayer_mosaic(w, h, seed=5)
    assert np.array_equal(_banded(torch, nodes, raw, w, h, n, fusion), _whole(torch, nodes, raw, w, h, fusion))


@pytest.mark.parametrize("w,h,n", [(1504, 1000, 2), (1504, 1000, 5), (600, 400, 3), (752, 2000, 8)])
def test_bands_with_the_amaze_demosaic_equal_the_unsplit_frame(w, h, n):
    """AMaZE on row bands: whole tile rows of the frame's own 128-row grid per band, 16 mosaic rows of either neighbour; the
    on-chip kernel indexes the band's buffers with frame rows"""
    from ansel_amd import abi
    torch, lut, d_lut = _setup()
    nodes = _nodes(w, h, d_lut, lut, abi.DT_HIP_DEMOSAIC_AMAZE)
    raw = synth.bayer_mosaic(w, h, seed=7)
    whole = _whole(torch, nodes, raw, w, h, True)
    assert np.array_equal(_banded(torch, nodes, raw, w, h, n, True), whole)
    assert np.array_equal(_c_driver(torch, nodes, raw, w, h, n), whole)


def test_amaze_bands_of_a_frame_with_tiles_of_the_first_kind_are_refused():
    """a frame of odd width keeps its last tile column in the first kernel, which has no row-band mode"""
    from ansel_amd import abi
    torch, lut, d_lut = _setup()
    w, h = 517, 389
    nodes = _nodes(w, h, d_lut, lut, abi.DT_HIP_DEMOSAIC_AMAZE)
    raw = synth.bayer_mosaic(w, h, seed=7)
    with pytest.raises(lib.AnselHipError, match="no band mode"):  # dt_hip_plan_bands() says so, before any band runs a stage
        _banded(torch, nodes, raw, w, h, 2, True)


@pytest.mark.parametrize("fusion", [True, False])
@pytest.mark.parametrize("n_top,n_bottom", [(10, 10), (20, 20), (0, 3), (24, 0), (13, 12)])
def test_highlights_bypass_is_decided_on_the_whole_frame(n_top, n_bottom, fusion):
    torch, lut, d_lut = _setup()
    w, h = 512, 600
    nodes = _nodes(w, h, d_lut, lut)
    raw = be.test_frame(w, h, n_top, n_bottom)
    whole = _whole(torch, nodes, raw, w, h, fusion)
    for n in (2, 3):
        assert np.array_equal(_banded(torch, nodes, raw, w, h, n, fusion), whole), n


def test_bands_equal_the_oracle():
    """and the unsplit oracle chain (CPU) is what both produce"""
    torch, lut, d_lut = _setup()
    w, h = 400, 300
    raw = be.test_frame(w, h, 9, 9)
    dev = _banded(torch, _nodes(w, h, d_lut, lut), raw, w, h, 2, True)
    host_nodes = pipe.light_pipe_nodes(w, h, lut.ctypes.data, float(lut[0]), params.unbounded_coeffs(lut),
                                       with_filmic=True, filmic=filmic.default_data())
    assert np.array_equal(dev, be.whole_frame(host_nodes, raw, w, h))


def test_rccl_accepts_runtime_owned_buffers():
    """world_size 1 over RCCL: the collectives run on views of dt_hip-owned memory"""
    torch, lut, d_lut = _setup()
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        w, h = 400, 300
        nodes = _nodes(w, h, d_lut, lut)
        raw = be.test_frame(w, h, 9, 9)
        p = pipe.DevicePipe(0, nodes, fusion=True)
        engine = tiled.HipBandEngine(p, "cuda:0")
        bands = tiled.plan_bands(w, h, 1)
        d_in = torch.from_numpy(raw.view(np.int16)).to("cuda:0")
        d_out = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda:0")
        work = engine.begin(bands[0], d_in.data_ptr(), w)
        before = int(work.count.item())
        dist.all_reduce(work.count)  # what sum_clipped() issues for N > 1
        assert int(work.count.item()) == before == 18
        engine.resolve(bands[0], work)
        engine.finish(bands[0], work, d_out.data_ptr())
        torch.cuda.synchronize()
        p.close()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint16), _whole(torch, nodes, raw, w, h, True))
        # the wavelets' table of partial sums is binary64 in runtime-owned memory: what serve_request() all-reduces
        vals = np.random.default_rng(1).random(4096)
        buf = lib.DeviceBuffer.from_numpy(0, vals)
        view = tiled.device_view(buf.ptr, (4096,), "<f8", torch.device("cuda:0"))
        dist.all_reduce(view, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert np.array_equal(buf.to_numpy((4096,), np.float64), vals)
        buf.release()
    finally:
        dist.destroy_process_group()


# ---- the stencil modules of the full pipe on row bands (config 4 of BASELINE.json) ----------------------------
def _full_nodes(w, h, d_lut, lut, which):
    from ansel_amd import abi
    nodes = pipe.denoise_pipe_nodes(w, h, d_lut.data_ptr(), float(lut[0]), params.unbounded_coeffs(lut),
                                    filmic=filmic.default_data(), diffuse_iterations=2, with_nlmeans=True,
                                    with_bilat=which in ("bilat", "bilat_fine", "everything"))
    drop = {"wavelets": ("diffuse", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "diffuse": ("denoiseprofile", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "diffuse_inpaint": ("denoiseprofile", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "nlmeans": ("denoiseprofile", "diffuse"),
            "dn_nlmeans": ("diffuse", "nlmeans", "rgb_to_lab", "lab_to_rgb"),
            "bilat": ("denoiseprofile", "diffuse", "nlmeans"), "bilat_fine": ("denoiseprofile", "diffuse", "nlmeans"),
            "blended": (), "all": (), "everything": ()}[which]
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
    if which == "bilat_fine":
        # a fine grid: many grid rows per band, band borders inside grid cells
        for n in nodes:
            if n.op == "bilat":
                n.data = abi.BilatData.bilateral(sigma_s=7.0, sigma_r=9.0, detail=-0.6)
    if which == "dn_nlmeans":
        for n in nodes:
            if n.op == "denoiseprofile":
                n.data = params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS)
    return nodes


@pytest.mark.parametrize("which", ["wavelets", "diffuse", "diffuse_inpaint", "nlmeans", "dn_nlmeans", "blended", "all", "bilat",
                                   "bilat_fine", "everything"])
@pytest.mark.parametrize("w,h,n", [(752, 2000, 2), (752, 2000, 5), (400, 640, 2), (400, 640, 1)])
def test_full_pipe_bands_equal_the_unsplit_frame(w, h, n, which):
    """denoise (profiled) wavelets / non-local means, diffuse-or-sharpen and nlmeans on row bands: halo rows
    from the neighbours, the wavelets' thresholds from the frame-wide sums, local contrast's bilateral grid relayed
    from band to band -- bit-identical to the unsplit run"""
    torch, lut, d_lut = _setup()
    nodes = _full_nodes(w, h, d_lut, lut, which)
    raw = synth.bayer_mosaic(w, h, seed=7)
    whole = _whole(torch, nodes, raw, w, h, True)
    banded = _banded(torch, nodes, raw, w, h, n, True)
    assert np.array_equal(banded, whole)
    assert whole.std() > 100  # a real picture came out


def test_full_pipe_bands_equal_the_oracle():
    torch, lut, d_lut = _setup()
    w, h = 256, 480
    raw = be.test_frame(w, h, 9, 9)
    dev = _banded(torch, _full_nodes(w, h, d_lut, lut, "all"), raw, w, h, 3, True)
    host = pipe.denoise_pipe_nodes(w, h, lut.ctypes.data, float(lut[0]), params.unbounded_coeffs(lut),
                                   filmic=filmic.default_data(), diffuse_iterations=2, with_nlmeans=True, with_bilat=False)
    assert np.array_equal(dev, be.whole_frame(host, raw, w, h))


def test_blended_pipe_closed_by_a_blend_on_bands():
    """a pipe whose LAST node is a blend: the module in front of it writes the band's output buffer, the blend works
    in place there"""
    from ansel_amd import abi
    torch, lut, d_lut = _setup()
    w, h = 400, 640
    rgb = abi.Piece.make(w, h, channels=4)
    nodes = [pipe.Node("diffuse", params.diffuse("lens_deblur_soft", iterations=2), rgb),
             pipe.Node("blend", abi.BlendData.uniform(params.WORK_IN, 35.0, abi.BLEND_AVERAGE), rgb)]
    img = synth.rgba_image(w, h, seed=3)
    d_in = torch.from_numpy(img).to("cuda:0")
    p = pipe.DevicePipe(0, nodes)
    whole = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    p.process(d_in.data_ptr(), whole.data_ptr())
    engine = tiled.HipBandEngine(p, "cuda:0")
    bands = tiled.plan_bands(w, h, 3, -1)
    outs = [torch.zeros((b.rows, w, 4), dtype=torch.float32, device="cuda:0") for b in bands]
    tiled.process_bands_locally(engine, bands, [d_in[b.row0:b.row0 + b.rows].data_ptr() for b in bands],
                                [t.data_ptr() for t in outs], w)
    torch.cuda.synchronize()
    p.close()
    assert torch.equal(torch.cat(outs, dim=0), whole)
    # and a mask blur is refused on bands
    d = abi.BlendData.uniform(params.WORK_IN, 50.0)
    d.blur_radius = 3.0
    p = pipe.DevicePipe(0, [nodes[0], pipe.Node("blend", d, rgb)])
    engine = tiled.HipBandEngine(p, "cuda:0")
    with pytest.raises(lib.AnselHipError, match="mask blur"):
        engine.begin(bands[0], d_in.data_ptr(), w)
    p.close()
    # and so is mask feathering (the guided filter's tile grid is the frame's)
    d = abi.BlendData.uniform(params.WORK_IN, 50.0)
    d.feathering_radius, d.feathering_guide = 3.0, abi.MASK_GUIDE_OUT_AFTER_BLUR
    p = pipe.DevicePipe(0, [nodes[0], pipe.Node("blend", d, rgb)])
    engine = tiled.HipBandEngine(p, "cuda:0")
    with pytest.raises(lib.AnselHipError, match="feathering"):
        engine.begin(bands[0], d_in.data_ptr(), w)
    p.close()


def test_blend_with_a_form_mask_on_bands():
    """a host-rendered form mask (drawn shapes + a parametric condition) in band mode: the plane is the frame's, every
    band reads its rows"""
    import blend_cases
    from ansel_amd import abi
    torch, lut, d_lut = _setup()
    w, h = 400, 640
    rgb = abi.Piece.make(w, h, channels=4)
    form = torch.from_numpy(blend_cases.form_plane(w, h)).to("cuda:0")
    d = dict(blend_cases.form_cases(abi.BLEND_CS_RGB_SCENE))["drawn-c0"]
    d.form_mask = form.data_ptr()
    nodes = [pipe.Node("diffuse", params.diffuse("lens_deblur_soft", iterations=1), rgb), pipe.Node("blend", d, rgb)]
    img = synth.rgba_image(w, h, seed=3)
    d_in = torch.from_numpy(img).to("cuda:0")
    p = pipe.DevicePipe(0, nodes)
    whole = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    p.process(d_in.data_ptr(), whole.data_ptr())
    engine = tiled.HipBandEngine(p, "cuda:0")
    bands = tiled.plan_bands(w, h, 3, -1)
    outs = [torch.zeros((b.rows, w, 4), dtype=torch.float32, device="cuda:0") for b in bands]
    tiled.process_bands_locally(engine, bands, [d_in[b.row0:b.row0 + b.rows].data_ptr() for b in bands],
                                [t.data_ptr() for t in outs], w)
    torch.cuda.synchronize()
    p.close()
    # bit patterns: the test plane carries a NaN, which the blend hands through
    assert torch.equal(torch.cat(outs, dim=0).view(torch.int32), whole.view(torch.int32))
    assert not torch.equal(whole.view(torch.int32), d_in.view(torch.int32))


@pytest.mark.parametrize("which", ["all", "everything"])
def test_band_abort_frees_what_a_stopped_walk_holds(which):
    """dt_hip_pipe_band_abort(): a band given up at any stop of the walk (an error on another rank, a cancelled export)
    returns every buffer it holds to the pool"""
    import ctypes as C
    torch, lut, d_lut = _setup()
    l = lib.load()
    w, h = 400, 640
    nodes = _full_nodes(w, h, d_lut, lut, which)
    raw = synth.bayer_mosaic(w, h, seed=7)
    p = pipe.DevicePipe(0, nodes)
    engine = tiled.HipBandEngine(p, "cuda:0")
    bands = tiled.plan_bands(w, h, 2, tiled.pipe_demosaic_method(nodes))
    d_in = torch.from_numpy(np.ascontiguousarray(raw[:bands[0].rows]).view(np.int16)).to("cuda:0")
    d_out = torch.zeros((bands[0].rows, w, 4), dtype=torch.int16, device="cuda:0")
    cur, peak = C.c_size_t(0), C.c_size_t(0)
    torch.cuda.synchronize()
    l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
    base = cur.value
    for stops in range(0, 14):
        work = engine.begin(bands[0], d_in.data_ptr(), w)
        engine.resolve(bands[0], work)
        done = False
        for _ in range(stops):
            if engine.finish(bands[0], work, d_out.data_ptr()) is None:
                done = True
                break
        if not done:
            l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
            assert cur.value > base                      # the stopped walk holds buffers
            engine.abort(work)
        l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
        assert cur.value == base, (stops, cur.value - base)
        engine.abort(work)                               # a second abort (or one after the walk ended) is a no-op
        if done:
            break
    assert done and stops >= 8                           # mosaic halo aside: wavelets 2 + 6 scales + sums, diffuse, nlmeans
    p.close()


def _laplacian_nodes(w, h, d_lut, lut):
    """local contrast in its local-laplacian mode: a pyramid over the frame, no row-band implementation"""
    from ansel_amd import abi
    nodes = pipe.denoise_pipe_nodes(w, h, d_lut.data_ptr(), float(lut[0]), params.unbounded_coeffs(lut),
                                    filmic=filmic.default_data(), with_nlmeans=True, with_bilat=True)
    for n in nodes:
        if n.op == "bilat":
            n.data = abi.BilatData.local_laplacian()
    return nodes


def test_modules_without_a_band_mode_are_refused():
    torch, lut, d_lut = _setup()
    w, h = 400, 640
    nodes = _laplacian_nodes(w, h, d_lut, lut)
    p = pipe.DevicePipe(0, nodes)
    engine = tiled.HipBandEngine(p, "cuda:0")
    bands = tiled.plan_bands(w, h, 2)
    d_in = torch.zeros((bands[0].rows, w), dtype=torch.int16, device="cuda:0")
    with pytest.raises(lib.AnselHipError, match="bilateral-grid mode only"):
        engine.begin(bands[0], d_in.data_ptr(), w)
    p.close()


# ---- the walk driven from inside the library: dt_hip_pipe_process_bands() (one C process, a host thread per band) ----
def _c_driver(torch, nodes, raw, w, h, n):
    import ctypes as C
    from ansel_amd import abi
    l = lib.load()
    pipes = [pipe.DevicePipe(0, nodes, fusion=True) for _ in range(n)]  # every band on the one device of the test box
    bands = tiled.plan_bands(w, h, n, tiled.pipe_demosaic_method(nodes))
    ins = [torch.from_numpy(np.ascontiguousarray(raw[b.row0:b.row0 + b.rows]).view(np.int16)).to("cuda:0") for b in bands]
    outs = [torch.zeros((b.rows, w, 4), dtype=torch.int16, device="cuda:0") for b in bands]
    torch.cuda.synchronize()
    rc = l.dt_hip_pipe_process_bands((C.c_void_p * n)(*[p.handle for p in pipes]), n, (abi.Band * n)(*bands),
                                     (C.c_void_p * n)(*[t.data_ptr() for t in ins]),
                                     (C.c_void_p * n)(*[t.data_ptr() for t in outs]))
    lib.check(rc, "dt_hip_pipe_process_bands")
    for p in pipes:
        p.close()
    return np.concatenate([t.cpu().numpy().view(np.uint16) for t in outs], axis=0)


@pytest.mark.parametrize("w,h,n", [(1504, 1000, 2), (1504, 1000, 5), (402, 640, 3), (752, 2000, 8)])
def test_c_driver_light_pipe_equals_the_unsplit_frame(w, h, n):
    torch, lut, d_lut = _setup()
    nodes = _nodes(w, h, d_lut, lut)
    raw = synth.bayer_mosaic(w, h, seed=5)
    assert np.array_equal(_c_driver(torch, nodes, raw, w, h, n), _whole(torch, nodes, raw, w, h, True))


@pytest.mark.parametrize("n_top,n_bottom", [(10, 10), (0, 3), (13, 12)])
def test_c_driver_decides_the_highlights_bypass_on_the_whole_frame(n_top, n_bottom):
    torch, lut, d_lut = _setup()
    w, h = 512, 600
    nodes = _nodes(w, h, d_lut, lut)
    raw = be.test_frame(w, h, n_top, n_bottom)
    assert np.array_equal(_c_driver(torch, nodes, raw, w, h, 3), _whole(torch, nodes, raw, w, h, True))


@pytest.mark.parametrize("which", ["wavelets", "diffuse", "nlmeans", "blended", "all", "bilat", "bilat_fine", "everything"])
@pytest.mark.parametrize("w,h,n", [(752, 2000, 5), (400, 640, 2)])
def test_c_driver_full_pipe_equals_the_unsplit_frame(w, h, n, which):
    """halo pulls between the bands' buffers and the all-gather of the wavelets' partial sums, done by the library"""
    torch, lut, d_lut = _setup()
    nodes = _full_nodes(w, h, d_lut, lut, which)
    raw = synth.bayer_mosaic(w, h, seed=7)
    assert np.array_equal(_c_driver(torch, nodes, raw, w, h, n), _whole(torch, nodes, raw, w, h, True))


def test_c_driver_reports_the_band_that_failed():
    torch, lut, d_lut = _setup()
    w, h = 400, 640
    nodes = _laplacian_nodes(w, h, d_lut, lut)
    raw = synth.bayer_mosaic(w, h, seed=7)
    with pytest.raises(lib.AnselHipError, match="bilateral-grid mode only"):
        _c_driver(torch, nodes, raw, w, h, 2)
    # and the pool is back where it was
    import ctypes as C
    cur, peak = C.c_size_t(0), C.c_size_t(0)
    l = lib.load()
    l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
    before = cur.value
    with pytest.raises(lib.AnselHipError):
        _c_driver(torch, nodes, raw, w, h, 2)
    l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
    assert cur.value == before


def test_c_driver_statistics_and_the_peer_self_test():
    """what a walk moved between devices (nothing here: one device), and the self-test of the calls a multi-device walk
    rests on -- peer access, a device-to-device copy ordered by a cross-device event -- on the devices that are there"""
    import ctypes as C
    from ansel_amd import abi
    torch, lut, d_lut = _setup()
    w, h = 752, 2000
    nodes = _full_nodes(w, h, d_lut, lut, "everything")
    raw = synth.bayer_mosaic(w, h, seed=7)
    _c_driver(torch, nodes, raw, w, h, 4)
    l = lib.load()
    st = abi.BandStats()
    l.dt_hip_pipe_bands_stats(C.byref(st))
    assert (st.bands, st.devices, st.peer_copies, st.peer_bytes, st.pairs_without_peer_access) == (4, 1, 0, 0, 0)
    assert st.exchange_stops >= 10  # seven wavelet scales, the sums, diffuse, non-local means, the bilateral relay
    ndev = torch.cuda.device_count()
    devs = (C.c_int * ndev)(*range(ndev))
    rc = l.dt_hip_peer_selftest(devs, ndev)
    assert rc == 0, l.dt_hip_last_error()
    assert l.dt_hip_peer_selftest(devs, 0) == abi.DT_HIP_INVALID_ARG
