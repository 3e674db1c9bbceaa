"""-m gpu: examples/export_pipe.c -- the C-ABI driven from plain C (init, pinned upload, executor, read-back) --
must export the same bytes as the Python-driven executor on the same synthetic mosaic."""
import os
import subprocess

import numpy as np
import pytest

import hipcheck as hc
from ansel_amd import abi, lib, pipe

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lcg_mosaic(w, h):
    s, out = 0x9E3779B97F4A7C15, np.empty(w * h, np.uint16)
    for k in range(w * h):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[k] = 512 + ((s >> 33) % 15000)
    return out.reshape(h, w)


def _fnv1a(b):
    h = 0xcbf29ce484222325
    for x in b.tobytes():
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_example_exports_the_same_bytes():
    exe = os.path.join(ROOT, "examples", "export_pipe")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    w, h = 208, 150
    out = subprocess.run([exe, str(w), str(h)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    fields = dict(f.split("=") for f in out.stdout.split() if "=" in f)
    assert fields["groups"] == "3"  # raw chain | rcd | exposure + u16
    # the same nodes through the ctypes binding
    hc.hip()
    wb = (2.1, 1.0, 1.6, 1.0)
    raw = _lcg_mosaic(w, h)
    p_raw = abi.Piece.make(w, h, filters=0x94949494, channels=1, datatype=abi.DT_HIP_TYPE_UINT16, processed_maximum=(1, 1, 1, 1))
    p_cfa = abi.Piece.make(w, h, filters=0x94949494, channels=1, processed_maximum=(1, 1, 1, 1))
    p_wb = abi.Piece.make(w, h, filters=0x94949494, channels=1, processed_maximum=wb)
    p_rgb = abi.Piece.make(w, h, channels=4, processed_maximum=wb)
    nodes = [pipe.Node("rawprepare", abi.RawprepareData(0, 0, 0, 0, abi.f4(512, 512, 512, 512), abi.f4(15871, 15871, 15871, 15871)), p_raw),
             pipe.Node("temperature", abi.TemperatureData(abi.f4(*wb)), p_cfa),
             pipe.Node("highlights", abi.HighlightsData(abi.DT_HIP_HIGHLIGHTS_CLIP, 1.0), p_wb),
             pipe.Node("demosaic", abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD, 0.0), p_wb),
             pipe.Node("exposure", abi.ExposureData(-0.000244140625, float(np.float32(1.6245047))), p_rgb),
             pipe.Node("export_u16", None, p_rgb)]
    din = lib.DeviceBuffer.from_numpy(0, raw)
    dout = lib.DeviceBuffer(0, w * h * 8)
    p = pipe.DevicePipe(0, nodes, fusion=True)
    p.process(din.ptr, dout.ptr)
    assert lib.load().dt_hip_finish(0) == 1
    got = dout.to_numpy((h, w, 4), np.uint16)
    p.close()
    assert "%016x" % _fnv1a(got) == fields["fnv1a"]


@pytest.mark.parametrize("bands", [2, 5])
def test_c_example_in_row_bands_exports_the_same_bytes(bands):
    """examples/export_pipe W H N: the frame cut into N row bands walked by dt_hip_pipe_process_bands() from plain C"""
    exe = os.path.join(ROOT, "examples", "export_pipe")
    w, h = 400, 900
    whole = subprocess.run([exe, str(w), str(h)], capture_output=True, text=True, timeout=120)
    split = subprocess.run([exe, str(w), str(h), str(bands)], capture_output=True, text=True, timeout=120)
    assert whole.returncode == 0 and split.returncode == 0, whole.stderr + split.stderr
    f0 = dict(f.split("=") for f in whole.stdout.split() if "=" in f)
    f1 = dict(f.split("=") for f in split.stdout.split() if "=" in f)
    assert f1["bands"] == str(bands) and f0["fnv1a"] == f1["fnv1a"]


def test_c_example_batch_with_a_writer_exports_the_same_bytes_every_frame():
    """examples/export_pipe W H -N: N frames through dt_hip_batch_* with the encoder as a writer callback, from plain C"""
    exe = os.path.join(ROOT, "examples", "export_pipe")
    w, h = 400, 300
    whole = subprocess.run([exe, str(w), str(h)], capture_output=True, text=True, timeout=120)
    batch = subprocess.run([exe, str(w), str(h), "-7"], capture_output=True, text=True, timeout=120)
    assert whole.returncode == 0 and batch.returncode == 0, whole.stderr + batch.stderr
    f0 = dict(f.split("=") for f in whole.stdout.split() if "=" in f)
    f1 = dict(f.split("=") for f in batch.stdout.split() if "=" in f)
    assert f1["written"] == "7" and f1["all_equal"] == "1" and f0["fnv1a"] == f1["fnv1a"]
