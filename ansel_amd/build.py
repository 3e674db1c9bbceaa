"""Build libansel_hip.so (the product) for gfx950 with hipcc, in-tree.

    python -m ansel_amd.build [--force] [--measuring]

--measuring builds ansel_amd/libansel_hip_measuring.so instead: every translation unit with -DANSEL_HIP_MEASURING, which
compiles the superseded kernel versions and the A/B switches of the kernels' development (read from the environment:
ANSEL_HIP_* / ANSEL_NLM2_*, hip_common.h measuring_env()) back in.  tools/ load it through ANSEL_HIP_LIB; the product
library has neither and reads no environment variable on a launch path.

hipcc cross-compiles without a GPU.  Each translation unit becomes an object under
ansel_amd/csrc/_obj/ (rebuilt only when it or a header is newer), then everything is linked
into ansel_amd/libansel_hip.so, which is what ansel_amd.lib loads and what travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libansel_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: one IEEE operation per source operation, nothing fused that the reference's
# strict CPU build does not fuse (explicit fmaf() stays a single v_fma_f32).
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
          "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

# per-file extra flags
EXTRA = {
    # memory-bound float4 taps: the one kernel that gains from the SLP vectoriser
    "diffuse_bspline.hip": ["-fslp-vectorize"],
    # rcd_demosaic() runs with FTZ/DAZ set (src/iop/demosaic/rcd.c:300)
    # -fno-slp-vectorize: the SLP vectoriser pairs binary32 operations into v_pk_*_f32, which issue at half rate on
    # gfx950 (no gain over two scalar operations) and need their operands in adjacent registers (560 v_mov in this
    # kernel)
    "demosaic_rcd.hip": ["-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize"],
    # (nlmeans.hip with -fslp-vectorize, round 6: 1 885 v_pk_add / v_pk_mul_f32 in nlm_chunks_v3 -- one issue slot for two of the A1 role's
    # column pairs -- and 22.1 against 21.5 - 22.0 ms at 100 MP: nothing, with spills in every other instantiation)
}


def _sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") or f.endswith(".cpp"):
            out.append(f)
    return out


def _headers_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return max(m, os.path.getmtime(os.path.abspath(__file__)))


def build(force=False, verbose=True, measuring=False):
    obj_dir = OBJ + "_measuring" if measuring else OBJ
    lib_path = os.path.join(HERE, "libansel_hip_measuring.so") if measuring else LIB
    common = COMMON + (["-DANSEL_HIP_MEASURING"] if measuring else [])
    os.makedirs(obj_dir, exist_ok=True)
    hm = _headers_mtime()
    objs = []
    procs = []
    for f in _sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(obj_dir, f.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hm):
            continue
        cmd = [HIPCC] + common + EXTRA.get(f, []) + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print("[ansel_amd.build] hipcc", f, flush=True)
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for f, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(out.decode(errors="replace"))
            sys.stderr.write("[ansel_amd.build] FAILED: %s\n" % f)
        elif verbose and out.strip():
            sys.stdout.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_path]
        if verbose:
            print("[ansel_amd.build] link", os.path.relpath(lib_path, ROOT), flush=True)
        subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, measuring="--measuring" in sys.argv)
