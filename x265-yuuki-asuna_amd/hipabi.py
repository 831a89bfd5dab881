"""ctypes binding of libx265hip.so (the C ABI declared in include/x265hip.h).

PyTorch is used only as plumbing: device memory (tensors), streams and torch.distributed.
Every wrapper passes raw device pointers + the current torch stream through the C ABI - the
same entry points a C++ host (x265 itself) would call.  There is no CPU fallback: if the
library is missing or no gfx950 device is present, calls raise.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libx265hip.so")

CMP_SAD, CMP_SATD, CMP_SA8D, CMP_SSE_PP, CMP_PSY_COST = range(5)

_lib = None


class X265HipError(RuntimeError):
    pass


class MEParams(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int), ("range", ctypes.c_int),
        ("fenc", ctypes.c_void_p), ("fenc_stride", ctypes.c_ssize_t),
        ("fref", ctypes.c_void_p), ("fref_stride", ctypes.c_ssize_t),
        ("surf", ctypes.c_void_p), ("best", ctypes.c_void_p),
        ("cost_x", ctypes.c_void_p), ("cost_y", ctypes.c_void_p),
    ]


def lib() -> ctypes.CDLL:
    """Load libx265hip.so (built in-tree by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise X265HipError(f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback for the product path")
        L = ctypes.CDLL(LIB_PATH)
        L.x265hip_version.restype = ctypes.c_char_p
        L.x265hip_last_error.restype = ctypes.c_char_p
        L.x265hip_table_calls.restype = ctypes.c_uint64 if hasattr(L, "x265hip_table_calls") else None
        L.x265hip_me_fullsearch.argtypes = [ctypes.POINTER(MEParams), ctypes.c_void_p]
        L.x265hip_me_best_reset.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.x265hip_pixelcmp_batch.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc < 0:
        raise X265HipError(f"{what} failed ({rc}): {lib().x265hip_last_error().decode()}")
    return rc


def exported_symbols():
    """Names include/x265hip.h declares; used by the CPU-only ABI test."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "x265hip.h")
    return sorted(set(re.findall(r"\b(x265hip_[a-z0-9_]+)\s*\(", open(hdr).read())))


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def me_fullsearch(depth, width, height, rng, fenc, fenc_stride, fref, fref_stride,
                  surf=None, best=None, cost_x=None, cost_y=None,
                  fenc_off=0, fref_off=0, stream=None):
    """fenc/fref: torch tensors holding the planes; *_off = element offset of pixel (0,0)."""
    es = 1 if depth == 8 else 2
    p = MEParams()
    p.depth, p.width, p.height, p.range = depth, width, height, rng
    p.fenc, p.fenc_stride = fenc.data_ptr() + fenc_off * es, fenc_stride
    p.fref, p.fref_stride = fref.data_ptr() + fref_off * es, fref_stride
    p.surf, p.best = _p(surf), _p(best)
    p.cost_x, p.cost_y = _p(cost_x), _p(cost_y)
    s = current_stream() if stream is None else stream
    check(lib().x265hip_me_fullsearch(ctypes.byref(p), s), "x265hip_me_fullsearch")


def me_best_reset(best, stream=None):
    s = current_stream() if stream is None else stream
    check(lib().x265hip_me_best_reset(best.data_ptr(), best.numel(), s), "x265hip_me_best_reset")


def pixelcmp_batch(kind, depth, w, h, a, a_stride, b, b_stride, njobs, out,
                   a_off=None, a_step=0, b_off=None, b_step=0, a_base=0, b_base=0, stream=None):
    es = 1 if depth == 8 else 2
    s = current_stream() if stream is None else stream
    check(lib().x265hip_pixelcmp_batch(kind, depth, w, h,
                                       a.data_ptr() + a_base * es, a_stride, _p(a_off), a_step,
                                       b.data_ptr() + b_base * es, b_stride, _p(b_off), b_step,
                                       njobs, out.data_ptr(), s), "x265hip_pixelcmp_batch")
