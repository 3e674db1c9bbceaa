"""CPU: the scale bookkeeping bench.py reports with (ansel_amd/modinfo.py) equals what the oracle
derives, over the frame sizes of BASELINE.json and a few odd ones."""
import ctypes as C

import pytest

import checkers as ck
from ansel_amd import abi, modinfo, params

SIZES = [(6000, 4000), (8256, 5504), (9504, 6336), (11648, 8736), (300, 200), (97, 1500), (64, 64)]


@pytest.mark.parametrize("w,h", SIZES)
def test_denoiseprofile_bands(w, h, oracle_lib):
    for scale in (1.0, 0.5, 0.23):
        piece = abi.Piece.make(w, h, roi_in=abi.Roi.make(0, 0, w, h, scale), roi_out=abi.Roi.make(0, 0, w, h, scale))
        d = params.denoiseprofile()
        assert modinfo.denoiseprofile_bands(piece) == oracle_lib.oracle_denoiseprofile_bands(C.byref(piece), C.byref(d))


@pytest.mark.parametrize("preset,over", [("default", {}), ("lens_deblur_soft", {}), ("fast_local_contrast", {}),
                                         ("default", dict(radius=1)), ("default", dict(radius=2048, radius_center=1024))])
def test_diffuse_scales(preset, over, oracle_lib):
    for scale in (1.0, 0.5, 0.1):
        piece = abi.Piece.make(100, 100, roi_in=abi.Roi.make(0, 0, 100, 100, scale), roi_out=abi.Roi.make(0, 0, 100, 100, scale))
        d = params.diffuse(preset, **over)
        assert modinfo.diffuse_scales(piece, d) == oracle_lib.oracle_diffuse_scales(C.byref(piece), C.byref(d))
