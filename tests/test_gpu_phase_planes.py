"""GPU parity: x265hip_phase_planes (every fractional phase of a reference plane in one pass) and its host-pointer consumer
x265hip_phase_cache vs the oracle's restatement, which applies the oracle's interpolation primitives - pinned against the real
reference table - block by block the way MotionEstimate::subpelCompare (motion.cpp:1571-1664) does."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
F = importlib.import_module("x265-yuuki-asuna_amd.frames")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _interior(a):
    return a[:, 8:-8, 8:-8]


def _planes(depth, seed, w=192, h=128):
    """Padded Y / Cb / Cr buffers of a synthetic picture with full-range noise mixed in (clipping and rounding paths)."""
    rng = np.random.default_rng(seed)
    y, cb, cr = F.synth_clip(w, h, 1, depth=depth, seed=seed)[0]
    pmax = (1 << depth) - 1
    y = y.copy(); y[::7, ::5] = pmax; y[3::11, 2::3] = 0
    ybuf, stride, _, w64, h64 = F.pad_plane(y)
    (cbuf, sc, _), (rbuf, _, _) = F.pad_chroma(cb, w64, h64), F.pad_chroma(cr, w64, h64)
    cbuf = cbuf.copy(); cbuf.reshape(-1)[::13] = rng.integers(0, pmax + 1, cbuf.reshape(-1)[::13].size)
    return ybuf.reshape(-1), stride, ybuf.size // stride, cbuf.reshape(-1), rbuf.reshape(-1), sc, cbuf.size // sc


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_phase_planes_match_oracle(depth):
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    ybuf, stride, rows, cbuf, rbuf, sc, rows_c = _planes(depth, 90 + depth)
    es = ybuf.itemsize
    for src, st, rw, chroma in ((ybuf, stride, rows, False), (cbuf, sc, rows_c, True)):
        guard_lo, guard_hi = 4 * st * es + 64, 8 * st * es
        d_src = torch.zeros(guard_lo + src.nbytes + guard_hi, dtype=torch.uint8, device=dev)
        d_src[guard_lo:guard_lo + src.nbytes] = torch.from_numpy(src.view(np.uint8)).to(dev)
        nph = 63 if chroma else 15
        d_dst = torch.zeros(nph * src.nbytes, dtype=torch.uint8, device=dev)
        A.phase_planes(depth, d_src, guard_lo, d_dst, st, rw, chroma=chroma)
        torch.cuda.synchronize()
        got = d_dst.cpu().numpy().view(src.dtype).reshape(nph, rw, st)
        want = O.phase_planes(depth, src, st, rw, chroma=chroma)
        for ph in range(nph):
            g, e = _interior(got[ph:ph + 1]), _interior(want[ph:ph + 1])
            assert np.array_equal(g, e), f"depth {depth} {'chroma' if chroma else 'luma'} phase {ph + 1}: {np.count_nonzero(g != e)} samples differ"
        assert len({_interior(got[ph:ph + 1]).tobytes() for ph in range(nph)}) == nph, "two phases produced the same plane"


class CacheParams(ctypes.Structure):
    """x265hip_phase_cache_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("stride", ctypes.c_ssize_t), ("rows", ctypes.c_int), ("stride_c", ctypes.c_ssize_t), ("rows_c", ctypes.c_int),
                ("slots", ctypes.c_int)]


@pytest.mark.parametrize("depth", [8, 10])
def test_phase_cache_serves_host_planes(depth):
    """The host-pointer consumer: submit two pictures into two slots, wait for the flags, compare the pinned planes; resubmitting a slot
    bumps its generation and the old flags no longer match."""
    O = _oracle()
    L = A.lib()
    pics = [_planes(depth, 120 + depth + k) for k in range(2)]
    _, stride, rows, _, _, sc, rows_c = pics[0]
    h = ctypes.c_void_p()
    L.x265hip_phase_cache_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(CacheParams)]
    A.check(L.x265hip_phase_cache_create(ctypes.byref(h), ctypes.byref(CacheParams(depth, stride, rows, sc, rows_c, 2))), "x265hip_phase_cache_create")
    L.x265hip_phase_cache_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.x265hip_phase_cache_planes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.x265hip_phase_cache_planes.restype = ctypes.c_void_p
    L.x265hip_phase_cache_ready.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.x265hip_phase_cache_ready.restype = ctypes.POINTER(ctypes.c_int)
    L.x265hip_phase_cache_destroy.argtypes = [ctypes.c_void_p]
    try:
        gens = [L.x265hip_phase_cache_submit(h, k, p[0].ctypes.data, p[3].ctypes.data, p[4].ctypes.data) for k, p in enumerate(pics)]
        assert all(g > 0 for g in gens)
        for k, p in enumerate(pics):
            rdy = L.x265hip_phase_cache_ready(h, k)
            t0 = time.time()
            while (rdy[0] != gens[k] or rdy[1] != gens[k]) and time.time() - t0 < 60:
                time.sleep(0.005)
            assert rdy[0] == gens[k] and rdy[1] == gens[k], "the planes never arrived"
            for plane, (src, st, rw, chroma) in enumerate(((p[0], stride, rows, False), (p[3], sc, rows_c, True), (p[4], sc, rows_c, True))):
                nph = 63 if chroma else 15
                raw = (ctypes.c_uint8 * (nph * src.nbytes)).from_address(L.x265hip_phase_cache_planes(h, k, plane))
                got = np.frombuffer(raw, dtype=src.dtype).reshape(nph, rw, st)
                assert np.array_equal(_interior(got), _interior(O.phase_planes(depth, src, st, rw, chroma=chroma))), f"slot {k} plane {plane}"
        g2 = L.x265hip_phase_cache_submit(h, 0, pics[1][0].ctypes.data, pics[1][3].ctypes.data, pics[1][4].ctypes.data)
        assert g2 == gens[0] + 1
        assert L.x265hip_phase_cache_submit(h, 5, pics[0][0].ctypes.data, None, None) < 0          # bad slot / missing chroma: refused
    finally:
        L.x265hip_phase_cache_destroy(h)


def test_phase_planes_at_4k():
    """BASELINE configs[2] size: the 15 luma and 63 Cb phase planes of a 3840x2160 picture (4032 x 2336 / 2112 x 1168 buffers), whole
    planes against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    ybuf, stride, rows, cbuf, _, sc, rows_c = _planes(8, 77, w=3840, h=2160)
    assert (stride, rows, sc, rows_c) == (4032, 2336, 2112, 1168)
    for src, st, rw, chroma in ((ybuf, stride, rows, False), (cbuf, sc, rows_c, True)):
        guard_lo, guard_hi = 4 * st + 64, 8 * st
        d_src = torch.zeros(guard_lo + src.nbytes + guard_hi, dtype=torch.uint8, device=dev)
        d_src[guard_lo:guard_lo + src.nbytes] = torch.from_numpy(src).to(dev)
        nph = 63 if chroma else 15
        d_dst = torch.zeros(nph * src.nbytes, dtype=torch.uint8, device=dev)
        A.phase_planes(8, d_src, guard_lo, d_dst, st, rw, chroma=chroma)
        torch.cuda.synchronize()
        got = d_dst.cpu().numpy().reshape(nph, rw, st)
        want = O.phase_planes(8, src, st, rw, chroma=chroma)
        assert np.array_equal(_interior(got), _interior(want)), f"{'chroma' if chroma else 'luma'} planes differ at 4K"
