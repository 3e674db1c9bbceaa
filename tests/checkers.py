"""Loaders for the two CPU checkers.  TEST INFRASTRUCTURE: only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this module.

  ref()     oracle/_ref/libansel_ref.so   the reference's own code (strict IEEE build)
  oracle()  oracle/liboracle.so           the committed C restatement

Both expose <prefix>_<op>(const dt_hip_piece_t*, const dt_hip_<op>_data_t*, in, out) on host
buffers with the C-ABI structs of include/ansel_hip.h."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


_cache = {}


def ref(fast=False):
    key = "ref_fast" if fast else "ref"
    if key not in _cache:
        _cache[key] = _load(os.path.join(ROOT, "oracle", "_ref",
                                         "libansel_ref_fast.so" if fast else "libansel_ref.so"))
    return _cache[key]


def oracle():
    if "oracle" not in _cache:
        _cache["oracle"] = _load(os.path.join(ROOT, "oracle", "liboracle.so"))
    return _cache["oracle"]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def aligned_empty(shape, dtype, align=64):
    """numpy only guarantees 16-byte alignment; the reference's loops carry `aligned(...:64)` clauses
    (pixelpipe buffers are 64-byte aligned there), so unaligned buffers fault in oracle/_ref"""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.empty(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def call(lib, name, piece, data, inp, out):
    fn = getattr(lib, name)
    fn.restype = C.c_int
    src = inp
    if inp.ctypes.data % 64:
        src = aligned_empty(inp.shape, inp.dtype)
        src[...] = inp
    dst = out
    if out.ctypes.data % 64:
        dst = aligned_empty(out.shape, out.dtype)
        dst[...] = out
    rc = fn(C.byref(piece), C.byref(data) if data is not None else None, ptr(src), ptr(dst))
    if dst is not out:
        out[...] = dst
    return rc


def ulp_diff(a, b):
    """per-element distance in units of last place between two float32 arrays; NaN==NaN -> 0,
    +0 vs -0 -> 0 (the sign of zero is compared separately where it matters)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    d[both_nan] = 0
    one_nan = np.isnan(a) ^ np.isnan(b)
    d[one_nan] = 1 << 40
    return d
