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


def _nodes(w, h, d_lut, lut):
    return pipe.light_pipe_nodes(w, h, d_lut.data_ptr(), float(lut[0]), params.unbounded_coeffs(lut),
                                 with_filmic=True, filmic=filmic.default_data())


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
@pytest.mark.parametrize("w,h,n", [(1504, 1000, 2), (1504, 1000, 5), (400, 300, 3), (752, 2000, 8)])
def test_bands_equal_the_unsplit_frame(w, h, n, fusion):
    torch, lut, d_lut = _setup()
    nodes = _nodes(w, h, d_lut, lut)
    raw = synth.bayer_mosaic(w, h, seed=5)
    assert np.array_equal(_banded(torch, nodes, raw, w, h, n, fusion), _whole(torch, nodes, raw, w, h, fusion))


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
    finally:
        dist.destroy_process_group()
