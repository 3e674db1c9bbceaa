"""-m gpu: the scanline packing of the format writers on the device (dt_hip_export_pack_rows, the "export_rows" pipe
node; src/imageio/format/tiff.c:293-360): `layers` of the 4 samples of every pixel, packed.  It moves bytes, so the
checker is numpy slicing."""
import ctypes as C

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import abi, filmic, lib, params, pipe, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(131, 67), (1, 1), (1, 50), (40, 3), (1504, 1000)])
@pytest.mark.parametrize("dtype,bpp", [(np.uint8, 8), (np.uint16, 16), (np.float32, 32)])
@pytest.mark.parametrize("layers", [3, 1])
def test_pack_rows(w, h, dtype, bpp, layers):
    l = hc.hip()
    rng = np.random.default_rng(w * 7 + h + bpp + layers)
    if dtype == np.float32:
        a = rng.standard_normal((h, w, 4)).astype(np.float32)
        a[0, 0, 0] = np.nan
    else:
        a = rng.integers(0, np.iinfo(dtype).max + 1, size=(h, w, 4)).astype(dtype)
    d_in = lib.DeviceBuffer.from_numpy(0, a)
    d_out = lib.DeviceBuffer(0, h * w * layers * a.itemsize)
    lib.check(l.dt_hip_export_pack_rows(0, w, h, bpp, layers, d_in.ptr, d_out.ptr), "pack_rows")
    got = d_out.to_numpy((h, w, layers), dtype)
    assert got.tobytes() == np.ascontiguousarray(a[..., :layers]).tobytes()


def test_pack_rows_refuses_what_no_writer_asks_for():
    l = hc.hip()
    buf = lib.DeviceBuffer(0, 1024)
    assert l.dt_hip_export_pack_rows(0, 4, 4, 16, 2, buf.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG
    assert l.dt_hip_export_pack_rows(0, 4, 4, 12, 3, buf.ptr, buf.ptr) == abi.DT_HIP_INVALID_ARG
    assert l.dt_hip_export_pack_rows(0, 0, 4, 16, 3, buf.ptr, buf.ptr) == abi.DT_HIP_SUCCESS


@pytest.mark.parametrize("fusion", [True, False])
@pytest.mark.parametrize("w,h", [(400, 300), (1504, 1000), (333, 517)])
def test_pipe_exports_scanlines(w, h, fusion):
    """the light pipe closed by export_u16 + export_rows: fused, the chain stores the scanlines itself"""
    hc.hip()
    import torch
    lut = params.srgb_encode_lut()
    d_lut = torch.from_numpy(lut).to("cuda:0")
    nodes = pipe.light_pipe_nodes(w, h, d_lut.data_ptr(), float(lut[0]), params.unbounded_coeffs(lut),
                                  with_filmic=True, filmic=filmic.default_data())
    raw = synth.bayer_mosaic(w, h, seed=11)
    d_in = torch.from_numpy(raw.view(np.int16)).to("cuda:0")
    p = pipe.DevicePipe(0, nodes, fusion=fusion)
    rgba = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda:0")
    p.process(d_in.data_ptr(), rgba.data_ptr())
    groups = p.num_groups
    p.close()
    rows_nodes = nodes + [pipe.Node("export_rows", abi.ExportRowsData(16, 3), nodes[-1].piece)]
    p = pipe.DevicePipe(0, rows_nodes, fusion=fusion)
    rows = torch.zeros((h, w, 3), dtype=torch.int16, device="cuda:0")
    p.process(d_in.data_ptr(), rows.data_ptr())
    torch.cuda.synchronize()
    assert p.num_groups == (groups if fusion else groups + 1)  # fused: no extra launch
    p.close()
    assert torch.equal(rows, rgba[..., :3])
    assert rows.cpu().numpy().view(np.uint16).std() > 100


def test_pipe_exports_8_bit_scanlines():
    """the 8-bit writers (jpeg, png8): ... -> colorout -> export_u8 -> export_rows{8, 3} == the u8 conversion of the
    float pipe output (imageio_core.c:706-727), alpha dropped"""
    hc.hip()
    import torch
    w, h = 400, 300
    lut = params.srgb_encode_lut()
    d_lut = torch.from_numpy(lut).to("cuda:0")
    nodes = pipe.light_pipe_nodes(w, h, d_lut.data_ptr(), float(lut[0]), params.unbounded_coeffs(lut),
                                  with_filmic=True, filmic=filmic.default_data())
    float_nodes = nodes[:-1]                                       # without export_u16: float4 out
    raw = synth.bayer_mosaic(w, h, seed=12)
    d_in = torch.from_numpy(raw.view(np.int16)).to("cuda:0")
    p = pipe.DevicePipe(0, float_nodes)
    f = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    p.process(d_in.data_ptr(), f.data_ptr())
    p.close()
    l = lib.load()
    u8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    lib.check(l.dt_hip_export_convert_u8(0, w, h, f.data_ptr(), u8.data_ptr()), "u8")
    rows_nodes = float_nodes + [pipe.Node("export_u8", None, nodes[-1].piece),
                                pipe.Node("export_rows", abi.ExportRowsData(8, 3), nodes[-1].piece)]
    p = pipe.DevicePipe(0, rows_nodes)
    rows = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    p.process(d_in.data_ptr(), rows.data_ptr())
    torch.cuda.synchronize()
    p.close()
    assert torch.equal(rows, u8[..., :3]) and rows.float().std() > 10
