"""What the compiler made of the shipped kernels, read from the code objects' metadata (tools/kernel_resources.py):
no scratch memory in the hot kernels of the export pipe (round 4's review found 12 - 24 bytes with folded spills in the
first wavelet scale while DESIGN.md said "no spills"), register counts that keep the occupancy DESIGN.md quotes, and the
kernarg offsets the kernels read their by-value parameter blocks at (hip_common.h kernarg_at())."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

OBJ = os.path.join(ROOT, "ansel_amd", "csrc", "_obj")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.isdir(OBJ) or not any(f.endswith(".o") for f in os.listdir(OBJ)):
        pytest.skip("no objects under ansel_amd/csrc/_obj (run __graft_entry__.build())")
    rows = kr.table(OBJ)
    assert len(rows) > 100
    return rows


# the kernels of the metric's workload (bench.py's full pipe, every launch group) and of the other BASELINE configurations
HOT = [r"^void dn_decompose_strip<", r"^void diffuse_pde_strip<(true|false), \d+, (true|false)>", r"bspline_decompose_strip", r"^void nlm_chunks_v3<9, [67], [12], (true|false)>",
       r"^void nlm_chunks_v4<9, 7, (true|false), [12], (true|false)>", r"^void nlm_tail<", r"^void ansel::rgb_chain<", r"^void dn_finish_chain<", r"rcd_tiles", r"raw_chain",
       r"bilat_(zcells|splat2|blur_yz|blur_x|slice)", r"dn_band_(sums|threshold)", r"^void filmic_kernel<", r"^void apply_matrix<", r"^void (rgb_to_lab|lab_to_rgb)<", r"^void channelmixerrgb<"]


def test_hot_kernels_use_no_scratch(kernels):
    hot = [k for k in kernels if any(re.search(p, k["demangled"]) for p in HOT)]
    unmatched = [p for p in HOT if not any(re.search(p, k["demangled"]) for k in kernels)]
    assert not unmatched, "patterns that name no kernel any more (a renamed kernel drops out of the guard silently): %s" % unmatched
    assert len(hot) >= 60, "the patterns must find the pipe's kernels: %d" % len(hot)
    bad = ["%s: %d B scratch, %d VGPR + %d SGPR spills" % (k["demangled"][:80], k["scratch"], k["vgpr_spills"], k["sgpr_spills"])
           for k in hot if k["scratch"] != 0 or k["vgpr_spills"] != 0]
    assert not bad, "\n".join(bad)


def test_occupancy_bounds_design_quotes(kernels):
    by = {}
    for k in kernels:
        by.setdefault(re.sub(r"\(.*", "", k["demangled"]), k)
    # eight workgroups of four waves a CU: <= 64 VGPRs for every wavelet launch a frame's seven scales take
    for name, k in by.items():
        if name.startswith("void dn_decompose_strip<") and not name.startswith("void dn_decompose_strip<true, 0"):
            assert k["vgpr"] <= 64, (name, k["vgpr"])
        # four waves a SIMD: <= 128 for the preset modes of the diffusion PDE and for the sixteen-wave non-local-means workgroups
        if re.match(r"void diffuse_pde_strip<(true|false), \d+, (true|false)>", name) or name.startswith("void nlm_chunks_v"):
            assert k["vgpr"] <= 128, (name, k["vgpr"])


def test_the_pde_has_its_dma_instantiations(kernels):
    """every preset mode of the diffusion PDE exists with its rows fetched by LDS-DMA (the chain's path at dilations <= 16) and
    without (the larger dilations, the stored-detail path)"""
    names = {re.sub(r"\(.*", "", k["demangled"]) for k in kernels}
    modes = {re.match(r"void diffuse_pde_strip<true, (-?\d+), true>", n).group(1) for n in names
             if re.match(r"void diffuse_pde_strip<true, -?\d+, true>", n)}
    assert len(modes) == 9, modes  # eight preset combinations + the run-time kinds
    for m in modes:
        assert "void diffuse_pde_strip<true, %s, false>" % m in names and "void diffuse_pde_strip<false, %s, false>" % m in names


def test_the_pde_dma_kernel_fetches_its_rows_without_destination_registers():
    """diffuse_pde_strip<true, MODE, true> takes its support rows by LDS-DMA (global_load_lds_dwordx4, inline assembly with M0 saved
    and restored in the statement): six per row step (three planes, main + halo piece) in each of the three unrolled row steps and
    in the prologue's three rows, and NO register-destination global load in the kernel but the luminance mask's byte"""
    import subprocess
    import tempfile
    obj = os.path.join(OBJ, "diffuse.o")
    if not os.path.exists(obj):
        pytest.skip("no diffuse.o")
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        subprocess.run([kr.LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], check=True)
        subprocess.run([kr.LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        text = subprocess.run([kr.LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
    # the bench's preset: orders 1 and 3 along the isophotes (PDE_MODE_DEBLUR = 253)
    m = re.search(r"<_ZN\S*diffuse_pde_stripILb1ELi253ELb1E\S*>:\n(.*?)\n\n", text, re.S)
    assert m, "diffuse_pde_strip<true, 253, true> not found in diffuse.o"
    body = m.group(1)
    assert body.count("global_load_lds_dwordx4") == 36, body.count("global_load_lds_dwordx4")
    assert len(re.findall(r"s_mov_b32 m0, s\d+", body)) >= 36 + 12  # a destination per piece + M0 restored per statement
    assert "global_load_dwordx4" not in body and "global_load_dwordx2" not in body
    assert body.count("global_store_dwordx4") >= 2


def test_kernarg_offsets_match_what_the_kernels_read(kernels):
    """kernarg_at<T>(offset) reads a by-value argument in place; the offset each kernel computes (and static_asserts) must be
    where the code object says the argument is"""
    want = {r"^void filmic_kernel<": (3, 24), r"^void blend_kernel<": (3, 24), r"^void blend_mask_kernel<": (3, 24),
            r"^void nlm_chunks_v3<": (2, 16), r"^void nlm_chunks_v4<": (2, 16), r"^void ansel::rgb_chain<": (3, 24)}
    seen = set()
    for k in kernels:
        for pat, (index, offset) in want.items():
            if re.search(pat, k["demangled"]):
                seen.add(pat)
                off, size, kind = k["args"][index]
                assert kind == "by_value" and off == offset and size > 16, (k["demangled"][:60], k["args"][:5])
    assert seen == set(want)
