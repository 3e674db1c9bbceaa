#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in the built library, read from the code objects' metadata.

    python tools/kernel_resources.py [--obj-dir ansel_amd/csrc/_obj] [--filter substring] [--json]

Each translation unit's object under ansel_amd/csrc/_obj carries its gfx950 code object in `.hip_fatbin`; its note
section lists, per kernel, .vgpr_count / .sgpr_count / .private_segment_fixed_size (scratch bytes per lane) /
.vgpr_spill_count / .sgpr_spill_count / .group_segment_fixed_size.  tests/test_kernel_resources.py asserts on this
table (no scratch in the shipped hot kernels); DESIGN.md quotes it.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    if not names:
        return []
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return out[:len(names)]


def kernels_of_object(obj):
    """[{name, vgpr, sgpr, scratch, vgpr_spills, sgpr_spills, lds}] of one .o (or .so with a single bundle)"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return []
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    # the note's payload is a YAML document: amdhsa.kernels is the list of kernel descriptors
    start = notes.find("---")
    end = notes.find("\n...", start)
    if start < 0:
        return []
    import yaml
    doc = yaml.safe_load(notes[start + 3:end if end > 0 else None])
    out = []
    for k in doc.get("amdhsa.kernels", []):
        out.append({"name": k[".name"], "vgpr": k.get(".vgpr_count", -1), "sgpr": k.get(".sgpr_count", -1),
                    "agpr": k.get(".agpr_count", 0), "scratch": k.get(".private_segment_fixed_size", -1),
                    "vgpr_spills": k.get(".vgpr_spill_count", -1), "sgpr_spills": k.get(".sgpr_spill_count", -1),
                    "lds": k.get(".group_segment_fixed_size", -1),
                    # the explicit arguments in declaration order: (offset, size, kind)
                    "args": [(a[".offset"], a[".size"], a[".value_kind"]) for a in k.get(".args", [])
                             if not a[".value_kind"].startswith("hidden_")]})
    return out


def table(obj_dir):
    rows = []
    for f in sorted(os.listdir(obj_dir)):
        if f.endswith(".o"):
            for k in kernels_of_object(os.path.join(obj_dir, f)):
                k["unit"] = f[:-2]
                rows.append(k)
    for k, d in zip(rows, demangle([k["name"] for k in rows])):
        k["demangled"] = re.sub(r"\(anonymous namespace\)::", "", d)
    return rows


def main():
    obj_dir = os.path.join(ROOT, "ansel_amd", "csrc", "_obj")
    flt = None
    a = sys.argv[1:]
    if "--obj-dir" in a:
        obj_dir = a[a.index("--obj-dir") + 1]
    if "--filter" in a:
        flt = a[a.index("--filter") + 1]
    rows = table(obj_dir)
    if flt:
        rows = [k for k in rows if flt in k["demangled"]]
    if "--json" in a:
        print(json.dumps(rows, indent=1))
        return
    print("%-22s %5s %5s %7s %6s %6s %7s  %s" % ("unit", "vgpr", "sgpr", "scratch", "vspill", "sspill", "lds", "kernel"))
    for k in rows:
        short = re.sub(r"\(.*", "", k["demangled"])
        print("%-22s %5d %5d %7d %6d %6d %7d  %s" % (k["unit"], k.get("vgpr", -1), k.get("sgpr", -1), k.get("scratch", -1),
                                                     k.get("vgpr_spills", -1), k.get("sgpr_spills", -1), k.get("lds", -1), short))


if __name__ == "__main__":
    main()
