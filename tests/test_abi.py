"""not gpu: the C-ABI library loads, exports every symbol include/ansel_hip.h declares, its struct
layouts match the ctypes mirror, and the product path fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from ansel_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ansel_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dt_hip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    l = lib.load()
    missing = [s for s in _declared_symbols() if not hasattr(l, s)]
    assert not missing, missing
    assert not l._ansel_missing, l._ansel_missing


def test_exports_beyond_the_header_are_the_test_hooks_and_nothing_else():
    """what the product library exports and include/ansel_hip.h does not declare: the device-side self-test entry points of
    testhooks.hip (dt_hip_test_*: devmath / in-range arithmetic on plain arrays, and dt_hip_test_dispatch(), which sends a
    launch to a fallback kernel so that a test can compare two kernels on one frame).  INTEGRATION.md says they are for the
    tests only; nothing else may leak out of the library"""
    import subprocess
    so = os.path.join(ROOT, "ansel_amd", "libansel_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    extra = sorted(e for e in exported - set(_declared_symbols()) if not e.startswith("_"))
    not_hooks = [e for e in extra if not e.startswith("dt_hip_test_")]
    assert not not_hooks, not_hooks
    assert "dt_hip_test_dispatch" in extra


def test_measuring_build_compiles():
    """every translation unit with -DANSEL_HIP_MEASURING (the A/B switches and superseded kernels tools/ load through
    ANSEL_HIP_LIB): no test runs it, so at least it must build -- objects that are up to date are not rebuilt"""
    from ansel_amd import build
    path = build.build(verbose=False, measuring=True)
    assert os.path.exists(path)
    import subprocess
    syms = subprocess.run(["nm", "-D", path], capture_output=True, text=True, check=True).stdout
    assert " U getenv" in syms  # the measuring build reads its switches from the environment; the product must not:
    prod = subprocess.run(["nm", "-D", os.path.join(ROOT, "ansel_amd", "libansel_hip.so")], capture_output=True, text=True, check=True).stdout
    assert " U getenv" not in prod


def test_ctypes_mirror_covers_the_header():
    declared = set(_declared_symbols())
    mirrored = set(lib.load()._ansel_protos)
    assert declared <= mirrored | {"dt_hip_abi_sizeof"}, sorted(declared - mirrored)


@pytest.mark.parametrize("name,ctype", [
    ("roi", abi.Roi), ("piece", abi.Piece), ("tiling", abi.Tiling), ("rawprepare", abi.RawprepareData),
    ("temperature", abi.TemperatureData), ("highlights", abi.HighlightsData), ("demosaic", abi.DemosaicData),
    ("exposure", abi.ExposureData), ("conversion", abi.Conversion), ("channelmixerrgb", abi.ChannelmixerrgbData),
    ("filmic_spline", abi.FilmicSpline), ("filmicrgb", abi.FilmicrgbData), ("diffuse", abi.DiffuseData), ("denoiseprofile", abi.DenoiseprofileData), ("nlmeans", abi.NlmeansData), ("lab", abi.LabData), ("bilat", abi.BilatData), ("finalscale", abi.FinalscaleData), ("blend", abi.BlendData), ("detailmask", abi.DetailmaskData), ("export_rows", abi.ExportRowsData), ("tile_plan", abi.TilePlan), ("band", abi.Band),
    ("band_state", abi.BandState), ("band_stats", abi.BandStats)])
def test_struct_sizes_match_the_compiled_library(name, ctype):
    l = lib.load()
    assert l.dt_hip_abi_sizeof(name.encode()) == C.sizeof(ctype), name


def test_no_cpu_fallback():
    """without a GPU the product refuses to run instead of computing on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(lib.AnselHipError):
        lib.init()
    l = lib.load()
    piece = abi.Piece.make(8, 8)
    d = abi.ExposureData(0.0, 1.0)
    buf = (C.c_float * 256)()
    rc = l.dt_hip_iop_exposure_process(0, C.byref(piece), C.byref(d), C.addressof(buf), C.addressof(buf))
    assert rc != abi.DT_HIP_SUCCESS


def test_product_does_not_touch_the_oracle():
    """nothing under ansel_amd/ or include/ may import, link or name the oracle"""
    bad = []
    for base in ("ansel_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "_obj" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"liboracle|libansel_ref|import\s+checkers|oracle_\w+\s*\(|ref_\w+process", text):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_product_library_reads_no_environment_on_a_launch_path():
    """the A/B switches of the kernels' development (superseded kernel versions, roles switched off, clock reads) exist in
    the MEASURING build only (ansel_amd/build.py --measuring, -DANSEL_HIP_MEASURING): the product library does not import
    getenv, names no ANSEL_* variable, and does not carry the superseded kernels"""
    import subprocess
    so = os.path.join(ROOT, "ansel_amd", "libansel_hip.so")
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", so]).decode()
    assert "getenv" not in undefined
    blob = open(so, "rb").read()
    assert b"ANSEL_HIP_" not in blob and b"ANSEL_NLM" not in blob
    for gone in (b"rcd_tiles_v1", b"nlm_chunks_v2_timed", b"bilat_lightness", b"18bspline_decomposeILi", b"12dn_decomposeEPK"):
        assert gone not in blob, gone
    # every getenv() in the sources sits behind the measuring build's macro
    for f in sorted(os.listdir(os.path.join(ROOT, "ansel_amd", "csrc"))):
        if not f.endswith((".hip", ".cpp", ".h")):
            continue
        lines = open(os.path.join(ROOT, "ansel_amd", "csrc", f)).read().split("\n")
        depth = []
        for ln in lines:
            s = ln.strip()
            if s.startswith("#ifdef ANSEL_HIP_MEASURING"):
                depth.append(True)
            elif s.startswith(("#if", "#ifdef", "#ifndef")):
                depth.append(False)
            elif s.startswith("#endif") and depth:
                depth.pop()
            elif re.search(r"(?<![_\w])getenv\s*\(", s.split("//")[0]) and not any(depth):
                raise AssertionError("%s: getenv() outside #ifdef ANSEL_HIP_MEASURING: %s" % (f, s))
