"""-m gpu: the export pipe end to end on the GPU -- module-by-module through the per-module C-ABI,
through the C++ executor (dt_hip_pipe_*) with fusion off and on -- must give the same bytes, and
those bytes must be what the CPU checkers produce for the same chain."""
import ctypes as C

import numpy as np
import pytest

import checkers as ck
import hipcheck as hc
from ansel_amd import abi, filmic, lib, params, pipe, synth

pytestmark = pytest.mark.gpu


def _setup(w, h, seed=1):
    hc.hip()
    raw = synth.bayer_mosaic(w, h, seed=seed)
    lut = params.srgb_encode_lut()
    d_lut = lib.DeviceBuffer.from_numpy(0, lut)
    coeffs = params.unbounded_coeffs(lut)
    return raw, lut, d_lut, coeffs


def _nodes(w, h, lut_ptr, lut, coeffs, **kw):
    return pipe.light_pipe_nodes(w, h, lut_ptr, float(lut[0]), coeffs, with_filmic=True,
                                 filmic=filmic.default_data(), **kw)


_GLUE = {"rgb_to_lab": "dt_hip_transform_rgb_to_lab", "lab_to_rgb": "dt_hip_transform_lab_to_rgb"}


def _run_chain_modulewise(nodes, raw, w, h):
    sizes = {"rawprepare": 4, "temperature": 4, "highlights": 4, "export_u16": 8}
    bufs = [lib.DeviceBuffer.from_numpy(0, raw)]
    for n in nodes:
        bufs.append(lib.DeviceBuffer(0, w * h * sizes.get(n.op, 16)))
    pipe.run_nodes(0, nodes, [b.ptr for b in bufs])
    assert lib.load().dt_hip_finish(0) == 1
    out = bufs[-1].to_numpy((h, w, 4), np.uint16)
    for b in bufs:
        b.release()
    return out


def _run_executor(nodes, raw, w, h, fusion):
    din = lib.DeviceBuffer.from_numpy(0, raw)
    dout = lib.DeviceBuffer(0, w * h * 8)
    p = pipe.DevicePipe(0, nodes, fusion=fusion)
    groups = p.num_groups
    p.process(din.ptr, dout.ptr)
    assert lib.load().dt_hip_finish(0) == 1
    out = dout.to_numpy((h, w, 4), np.uint16)
    p.close()
    din.release()
    dout.release()
    return out, groups


def _run_cpu(which, nodes, raw, w, h):
    l = ck.ref() if which == "ref" else ck.oracle()
    prefix = "ref_" if which == "ref" else "oracle_"
    src = raw
    for n in nodes:
        if n.op == "export_u16":
            out = np.zeros((h, w, 4), np.uint16)
            getattr(l, prefix + "export_convert_u16")(w, h, ck.ptr(src), ck.ptr(out))
            return out
        dst = np.zeros((h, w) if n.op in ("rawprepare", "temperature", "highlights") else (h, w, 4), np.float32)
        assert ck.call(l, prefix + n.op, n.piece, n.data, np.ascontiguousarray(src), dst) == 0, n.op
        src = dst


@pytest.mark.parametrize("w,h", [(1504, 1000), (400, 300)])
def test_fused_pipe_equals_modulewise_and_cpu(w, h):
    raw, lut, d_lut, coeffs = _setup(w, h)
    dev_nodes = _nodes(w, h, d_lut.ptr, lut, coeffs)
    modulewise = _run_chain_modulewise(dev_nodes, raw, w, h)
    unfused, g0 = _run_executor(dev_nodes, raw, w, h, fusion=False)
    fused, g1 = _run_executor(dev_nodes, raw, w, h, fusion=True)
    assert g0 == len(dev_nodes) and g1 == 3, (g0, g1)  # raw chain, demosaic, rgb chain
    assert np.array_equal(modulewise, unfused)
    assert np.array_equal(modulewise, fused)
    host_nodes = _nodes(w, h, lut.ctypes.data, lut, coeffs)
    mask = np.zeros((h, w), np.uint8)
    ck.oracle().oracle_rcd_stale_mask(ck.ptr(mask), w, h, C.c_uint32(synth.FILTERS_RGGB))
    stale = np.zeros((h, w), bool)
    stale[:, w - 9:w - 6] = mask[:, w - 9:w - 6] != 0
    for which in hc.checkers_available():
        cpu = _run_cpu(which, host_nodes, raw, w, h)
        diff = (cpu != fused).any(axis=-1)
        if which == "ref":
            diff &= ~stale  # reference reads stale scratch on <= 3 columns
        assert not diff.any(), "%s: %d pixels differ" % (which, int(diff.sum()))


@pytest.mark.parametrize("version,preserve_color", [(0, 0), (1, 1), (2, 3)])
def test_pipe_with_a_2019_colour_science_runs_filmic_in_its_own_launch(version, preserve_color):
    """old edits (filmic colour sciences v1..v3): the planner keeps the module out of the fused run"""
    w, h = 400, 300
    raw, lut, d_lut, coeffs = _setup(w, h, seed=9)
    fd = filmic.commit(filmic.UserParams.defaults(version=version, preserve_color=preserve_color, saturation=10.0))
    dev_nodes = pipe.light_pipe_nodes(w, h, d_lut.ptr, float(lut[0]), coeffs, with_filmic=True, filmic=fd)
    modulewise = _run_chain_modulewise(dev_nodes, raw, w, h)
    fused, groups = _run_executor(dev_nodes, raw, w, h, fusion=True)
    assert groups == 5, groups  # raw chain | demosaic | exposure..calibration | filmic | colorout, u16
    assert np.array_equal(modulewise, fused)
    host_nodes = pipe.light_pipe_nodes(w, h, lut.ctypes.data, float(lut[0]), coeffs, with_filmic=True, filmic=fd)
    cpu = _run_cpu("oracle", host_nodes, raw, w, h)
    assert np.array_equal(cpu, fused)


def test_executor_falls_back_to_single_launches_for_unfusable_geometry():
    """width not a multiple of 4: the CFA group is not fusable; results must still be exact"""
    w, h = 402, 301
    raw, lut, d_lut, coeffs = _setup(w, h, seed=5)
    nodes = _nodes(w, h, d_lut.ptr, lut, coeffs)
    modulewise = _run_chain_modulewise(nodes, raw, w, h)
    fused, groups = _run_executor(nodes, raw, w, h, fusion=True)
    assert groups == 5  # rawprepare, temperature, highlights, demosaic, rgb chain
    assert np.array_equal(modulewise, fused)


def test_executor_bypass_of_highlight_clipping_in_fused_raw_chain():
    """fewer than 25 clipped photosites: the fused CFA group must copy them through unclipped"""
    w, h = 400, 300
    raw, lut, d_lut, coeffs = _setup(w, h, seed=7)
    raw = np.minimum(raw, 6000).astype(np.uint16)
    raw[10, 10] = 16383
    raw[20, 31] = 16383
    nodes = _nodes(w, h, d_lut.ptr, lut, coeffs)
    modulewise = _run_chain_modulewise(nodes, raw, w, h)
    fused, _ = _run_executor(nodes, raw, w, h, fusion=True)
    assert np.array_equal(modulewise, fused)


def test_executor_rejects_unknown_module():
    l = hc.hip()
    p = l.dt_hip_pipe_new(0)
    piece = abi.Piece.make(8, 8)
    assert l.dt_hip_pipe_add_node(p, b"lens", C.byref(piece), None, 0) == abi.DT_HIP_INVALID_ARG
    d = abi.ExposureData(0.0, 1.0)
    assert l.dt_hip_pipe_add_node(p, b"exposure", C.byref(piece), C.cast(C.byref(d), C.c_void_p), 4) == abi.DT_HIP_INVALID_ARG
    l.dt_hip_pipe_free(p)


def test_denoise_pipe_executor_equals_modulewise_and_oracle():
    """config 3 as far as it runs on device: + denoise (profiled) wavelets + diffuse or sharpen.  The
    stencil modules run as their own launches between the fused pointwise groups."""
    w, h = 640, 400
    raw, lut, d_lut, coeffs = _setup(w, h, seed=4)
    nodes = pipe.denoise_pipe_nodes(w, h, d_lut.ptr, float(lut[0]), coeffs, filmic=filmic.default_data())
    modulewise = _run_chain_modulewise(nodes, raw, w, h)
    fused, groups = _run_executor(nodes, raw, w, h, fusion=True)
    unfused, g0 = _run_executor(nodes, raw, w, h, fusion=False)
    assert g0 == len(nodes) and groups == 6, (g0, groups)  # raw | rcd | denoise | exp,colorin,calib | diffuse | filmic,colorout,u16
    assert np.array_equal(modulewise, fused) and np.array_equal(modulewise, unfused)
    host_nodes = pipe.denoise_pipe_nodes(w, h, lut.ctypes.data, float(lut[0]), coeffs, filmic=filmic.default_data())
    assert np.array_equal(fused, _run_cpu("oracle", host_nodes, raw, w, h))


def test_full_config3_pipe_with_nlmeans():
    """... + RGB -> Lab, denoise (non-local means), local contrast (bilateral grid), Lab -> RGB between
    diffuse and filmic: every module of BASELINE.json config 3"""
    w, h = 300, 220
    raw, lut, d_lut, coeffs = _setup(w, h, seed=6)
    nodes = pipe.denoise_pipe_nodes(w, h, d_lut.ptr, float(lut[0]), coeffs, filmic=filmic.default_data(), with_nlmeans=True)
    fused, groups = _run_executor(nodes, raw, w, h, fusion=True)
    assert groups == 9, groups  # "lab_to_rgb" is the first stage of the fused run that follows it
    assert np.array_equal(fused, _run_chain_modulewise(nodes, raw, w, h))
    host_nodes = pipe.denoise_pipe_nodes(w, h, lut.ctypes.data, float(lut[0]), coeffs, filmic=filmic.default_data(),
                                         with_nlmeans=True)
    assert np.array_equal(fused, _run_cpu("oracle", host_nodes, raw, w, h))


def _blended_nodes(w, h, lut_ptr, lut, coeffs):
    """the light pipe with a uniform 60 % multiply blend on exposure and a parametric mask (grey + Jz in, hue
    out, tone curve) on color calibration"""
    nodes = _nodes(w, h, lut_ptr, lut, coeffs)
    rgb = abi.Piece.make(w, h, channels=4, processed_maximum=synth.WB_COEFFS)
    out = []
    for n in nodes:
        out.append(n)
        if n.op == "exposure":
            out.append(pipe.Node("blend", abi.BlendData.uniform(params.WORK_IN, 60.0, abi.BLEND_MULTIPLY, 0.5), rgb))
        if n.op == "channelmixerrgb":
            d = abi.BlendData.uniform(params.WORK_IN, 80.0)
            d.channel(abi.BLENDIF_GRAY_in, 0.02, 0.15, 0.6, 0.9, boost=1.0)
            d.channel(abi.BLENDIF_Jz_in, 0.05, 0.2, 1.0, 1.0, boost=-4.0)
            d.channel(abi.BLENDIF_hz_out, 0.1, 0.3, 0.8, 0.95)
            d.contrast, d.brightness = 0.3, -0.2
            out.append(pipe.Node("blend", d, rgb))
    return out


def _run_cpu_blended(nodes, raw, w, h):
    """the oracle chain with the blend stage: blend(input of the module, output of the module) in place"""
    o = ck.oracle()
    src, prev = raw, None
    for n in nodes:
        if n.op == "export_u16":
            out = np.zeros((h, w, 4), np.uint16)
            o.oracle_export_convert_u16(w, h, ck.ptr(src), ck.ptr(out))
            return out
        if n.op == "blend":
            assert ck.call(o, "oracle_develop_blend", n.piece, n.data, np.ascontiguousarray(prev), src) == 0
            continue
        dst = np.zeros((h, w) if n.op in ("rawprepare", "temperature", "highlights") else (h, w, 4), np.float32)
        assert ck.call(o, "oracle_" + n.op, n.piece, n.data, np.ascontiguousarray(src), dst) == 0, n.op
        prev, src = src, dst


def test_executor_runs_the_blend_stage():
    """a blended module is not fused with its neighbours (its input and output both exist as buffers), the
    blend runs in place in its output, and the result is the oracle's"""
    w, h = 640, 400
    raw, lut, d_lut, coeffs = _setup(w, h, seed=9)
    nodes = _blended_nodes(w, h, d_lut.ptr, lut, coeffs)
    fused, g1 = _run_executor(nodes, raw, w, h, fusion=True)
    unfused, g0 = _run_executor(nodes, raw, w, h, fusion=False)
    # raw chain | rcd | exposure | blend | colorin | calibration | blend | filmic, colorout, u16
    assert g0 == len(nodes) and g1 == 8, (g0, g1)
    assert np.array_equal(fused, unfused)
    host_nodes = _blended_nodes(w, h, lut.ctypes.data, lut, coeffs)
    cpu = _run_cpu_blended(host_nodes, raw, w, h)
    diff = (cpu != fused).any(axis=-1)
    assert not diff.any(), "%d pixels differ" % int(diff.sum())
    plain, _ = _run_executor(_nodes(w, h, d_lut.ptr, lut, coeffs), raw, w, h, fusion=True)
    assert not np.array_equal(plain, fused)


def test_executor_blend_as_last_node_and_misplaced():
    w, h = 64, 48
    hc.hip()
    img = synth.rgba_image(w, h, seed=2, lo=0.0, hi=1.5)
    rgb = abi.Piece.make(w, h, channels=4)
    nodes = [pipe.Node("exposure", abi.ExposureData(0.01, 1.7), rgb),
             pipe.Node("blend", abi.BlendData.uniform(params.WORK_IN, 35.0, abi.BLEND_AVERAGE), rgb)]
    din = lib.DeviceBuffer.from_numpy(0, img)
    dout = lib.DeviceBuffer(0, w * h * 16)
    p = pipe.DevicePipe(0, nodes, fusion=True)
    p.process(din.ptr, dout.ptr)
    assert lib.load().dt_hip_finish(0) == 1
    got = dout.to_numpy((h, w, 4), np.float32)
    p.close()
    o = ck.oracle()
    mid = np.zeros_like(img)
    assert ck.call(o, "oracle_exposure", rgb, nodes[0].data, img, mid) == 0
    assert ck.call(o, "oracle_develop_blend", rgb, nodes[1].data, img, mid) == 0
    assert int((ck.ulp_diff(got, mid) > 0).sum()) == 0
    # a blend node with no module in front of it is an error, not a no-op
    p = pipe.DevicePipe(0, nodes[1:], fusion=True)
    with pytest.raises(lib.AnselHipError):
        p.process(din.ptr, dout.ptr)
    p.close()


def test_very_large_frame_fused_equals_modulewise():
    """201 MP (16384 x 12288): 3.2 GB per float4 plane, byte offsets beyond 2^31 in every RGBA kernel; the fused
    executor, the unfused executor and the per-module entry points must export the same bytes"""
    w, h = 16384, 12288
    hc.hip()
    raw = synth.bayer_mosaic_tiled(w, h, seed=3)
    lut = params.srgb_encode_lut()
    d_lut = lib.DeviceBuffer.from_numpy(0, lut)
    coeffs = params.unbounded_coeffs(lut)
    nodes = _nodes(w, h, d_lut.ptr, lut, coeffs)
    fused, g1 = _run_executor(nodes, raw, w, h, fusion=True)
    assert g1 == 3
    # the last rows are where a 32-bit offset would have wrapped
    assert fused[-64:].any() and fused[:64].any()
    unfused, _ = _run_executor(nodes, raw, w, h, fusion=False)
    assert np.array_equal(fused, unfused)
    del unfused
    modulewise = _run_chain_modulewise(nodes, raw, w, h)
    assert np.array_equal(fused, modulewise)


def test_executor_returns_every_intermediate_to_the_pool():
    """dt_hip_pipe_process() on the full pipe (stencil modules, Lab glue, two blends) and on a pipe that fails in the
    middle: the runtime's allocation counter is back at its baseline afterwards"""
    import ctypes as C
    import torch
    from ansel_amd import filmic
    l = hc.hip()
    w, h = 400, 300
    lut = params.srgb_encode_lut()
    d_lut = torch.from_numpy(lut).to("cuda:0")
    coeffs = params.unbounded_coeffs(lut)
    nodes = pipe.denoise_pipe_nodes(w, h, d_lut.data_ptr(), float(lut[0]), coeffs, filmic=filmic.default_data(),
                                    diffuse_iterations=2, with_nlmeans=True, with_bilat=True)
    out = []
    for n in nodes:
        out.append(n)
        if n.op in ("exposure", "diffuse"):
            out.append(pipe.Node("blend", abi.BlendData.uniform(params.WORK_IN, 60.0, abi.BLEND_MULTIPLY, 0.5), n.piece))
    raw = torch.from_numpy(synth.bayer_mosaic(w, h, seed=2).view(np.int16)).to("cuda:0")
    res = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda:0")
    cur, peak = C.c_size_t(0), C.c_size_t(0)
    torch.cuda.synchronize()
    l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
    base = cur.value
    p = pipe.DevicePipe(0, out)
    for _ in range(2):
        p.process(raw.data_ptr(), res.data_ptr())
    torch.cuda.synchronize()
    l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
    assert cur.value == base and peak.value > base
    p.close()
    # a module that refuses its parameters half way down the pipe (diffuse with iscale 0)
    bad = [n for n in nodes]
    for k, n in enumerate(bad):
        if n.op == "diffuse":
            d = params.diffuse("lens_deblur_soft", iterations=1)
            d.iscale = 0.0
            bad[k] = pipe.Node("diffuse", d, n.piece)
    p = pipe.DevicePipe(0, bad)
    with pytest.raises(lib.AnselHipError):
        p.process(raw.data_ptr(), res.data_ptr())
    torch.cuda.synchronize()
    l.dt_hip_memory_statistics(0, C.byref(cur), C.byref(peak))
    assert cur.value == base
    p.close()


def test_lab_glue_is_fused_into_the_pointwise_runs():
    """"lab_to_rgb" in front of a fusable run and "rgb_to_lab" behind one are stages of that run's kernel: one launch, the
    same bits as the module-by-module chain"""
    w, h = 320, 200
    lut = params.srgb_encode_lut()
    rgb = abi.Piece.make(w, h, channels=4, processed_maximum=synth.WB_COEFFS)
    img = synth.rgba_image(w, h, seed=9, lo=0.0, hi=1.0)
    lab_in = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_rgb_to_lab", rgb, abi.LabData.make(params.WORK_IN), img, lab_in) == 0
    nodes = [pipe.Node("lab_to_rgb", abi.LabData.make(params.WORK_OUT), rgb),
             pipe.Node("exposure", abi.ExposureData(-0.000244140625, 1.6245047), rgb),
             pipe.Node("rgb_to_lab", abi.LabData.make(params.WORK_IN), rgb)]
    outs = {}
    for fusion in (True, False):
        din = lib.DeviceBuffer.from_numpy(0, lab_in)
        dout = lib.DeviceBuffer(0, w * h * 16)
        p = pipe.DevicePipe(0, nodes, fusion=fusion)
        assert p.num_groups == (1 if fusion else 3)
        p.process(din.ptr, dout.ptr)
        assert lib.load().dt_hip_finish(0) == 1
        outs[fusion] = dout.to_numpy((h, w, 4), np.float32)
        p.close()
    assert np.array_equal(outs[True].view(np.uint32), outs[False].view(np.uint32))
    o = ck.oracle()
    a, b, c = np.zeros_like(img), np.zeros_like(img), np.zeros_like(img)
    assert ck.call(o, "oracle_lab_to_rgb", rgb, nodes[0].data, lab_in, a) == 0
    assert ck.call(o, "oracle_exposure", rgb, nodes[1].data, a, b) == 0
    assert ck.call(o, "oracle_rgb_to_lab", rgb, nodes[2].data, b, c) == 0
    assert np.array_equal(outs[True].view(np.uint32), c.view(np.uint32))


def test_lab_glue_with_tone_curves_runs_in_its_own_launch():
    """a work profile with tone curves: the conversions are not stages of the fused run; same bits as the oracle chain"""
    w, h = 320, 200
    enc, dec = params.srgb_encode_lut(), params.srgb_decode_lut()
    d_enc, d_dec = lib.DeviceBuffer.from_numpy(0, enc), lib.DeviceBuffer.from_numpy(0, dec)
    ce, cd = params.unbounded_coeffs(enc), params.unbounded_coeffs(dec)
    rgb = abi.Piece.make(w, h, channels=4, processed_maximum=synth.WB_COEFFS)
    img = synth.rgba_image(w, h, seed=9, lo=0.0, hi=1.0)
    lab_in = np.zeros_like(img)
    assert ck.call(ck.oracle(), "oracle_rgb_to_lab", rgb, abi.LabData.make(params.WORK_IN), img, lab_in) == 0

    def nodes(dev):
        return [pipe.Node("lab_to_rgb", abi.LabData.make(params.WORK_OUT, [(d_enc.ptr if dev else enc.ctypes.data, float(enc[0]), ce)] * 3), rgb),
                pipe.Node("exposure", abi.ExposureData(-0.000244140625, 1.6245047), rgb),
                pipe.Node("rgb_to_lab", abi.LabData.make(params.WORK_IN, [(d_dec.ptr if dev else dec.ctypes.data, float(dec[0]), cd)] * 3), rgb)]
    din = lib.DeviceBuffer.from_numpy(0, lab_in)
    dout = lib.DeviceBuffer(0, w * h * 16)
    p = pipe.DevicePipe(0, nodes(True), fusion=True)
    assert p.num_groups == 3
    p.process(din.ptr, dout.ptr)
    assert lib.load().dt_hip_finish(0) == 1
    got = dout.to_numpy((h, w, 4), np.float32)
    p.close()
    o = ck.oracle()
    host = nodes(False)
    a, b, c = np.zeros_like(img), np.zeros_like(img), np.zeros_like(img)
    assert ck.call(o, "oracle_lab_to_rgb", rgb, host[0].data, lab_in, a) == 0
    assert ck.call(o, "oracle_exposure", rgb, host[1].data, a, b) == 0
    assert ck.call(o, "oracle_rgb_to_lab", rgb, host[2].data, b, c) == 0
    assert np.array_equal(got.view(np.uint32), c.view(np.uint32))


def test_detailmask_stage_and_a_details_threshold_in_a_pipe():
    """the hidden stage writes its plane while the frame goes through; a blend further down refines its mask with it"""
    w, h = 320, 200
    hc.hip()
    img = synth.rgba_image(w, h, seed=6, lo=0.0, hi=1.5)
    rgb = abi.Piece.make(w, h, channels=4)
    d_plane = lib.DeviceBuffer(0, w * h * 4)
    h_plane = ck.aligned_empty((h, w), np.float32)

    def nodes(plane_ptr):
        bd = abi.BlendData.uniform(params.WORK_IN, 75.0).channel(abi.BLENDIF_GRAY_in, 0.05, 0.2, 0.6, 0.9, boost=1.0)
        bd.details = 0.3
        bd.detail_mask = plane_ptr
        return [pipe.Node("detailmask", abi.DetailmaskData.make((2.0, 1.0, 1.5), plane_ptr), rgb),
                pipe.Node("exposure", abi.ExposureData(0.01, 1.7), rgb),
                pipe.Node("blend", bd, rgb)]
    din = lib.DeviceBuffer.from_numpy(0, img)
    dout = lib.DeviceBuffer(0, w * h * 16)
    p = pipe.DevicePipe(0, nodes(d_plane.ptr), fusion=True)
    p.process(din.ptr, dout.ptr)
    assert lib.load().dt_hip_finish(0) == 1
    got = dout.to_numpy((h, w, 4), np.float32)
    p.close()
    o = ck.oracle()
    host = nodes(h_plane.ctypes.data)
    a, b = np.zeros_like(img), np.zeros_like(img)
    assert ck.call(o, "oracle_detailmask", rgb, host[0].data, img, a) == 0
    assert ck.call(o, "oracle_exposure", rgb, host[1].data, a, b) == 0
    assert ck.call(o, "oracle_develop_blend", rgb, host[2].data, a, b) == 0
    assert int((ck.ulp_diff(got, b) > 0).sum()) == 0
    assert int((ck.ulp_diff(d_plane.to_numpy((h, w), np.float32), h_plane) > 0).sum()) == 0
