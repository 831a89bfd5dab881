"""GPU parity of the batch layer's CONSUMER (csrc/me_cache.hip) and of the stage-level seam built on it:

  * x265hip_me_cache: host planes in -> one exhaustive-search launch -> SAD surfaces streamed into pinned host memory row by row;
    the surfaces and x265hip_surf_lookup's address arithmetic against the oracle's exhaustive search;
  * the REAL reference encoder (oracle/_ref/libx265ref<depth>_seam.so) whose motion search looks its integer SADs up in those
    surfaces: byte-identical bitstream vs the pristine reference build, every lookup verified against the C primitive in flight."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = importlib.import_module("x265-yuuki-asuna_amd.frames")


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api
    return oracle_api


@pytest.mark.parametrize("depth,width,height,rng", [(8, 256, 192, 20), (8, 200, 136, 57), (10, 256, 128, 16)])
def test_me_cache_surfaces_equal_oracle(depth, width, height, rng):
    from tools import seam_driver as SD
    O = _oracle()
    geo = SD.geometry(width, height)
    prov = SD.GpuProvider(depth, geo, rng, 2)
    try:
        clip = F.synth_clip(width, height, 3, depth=depth, seed=51)
        planes = [F.pad_plane(y)[0] for (y, _, _) in clip]
        assert planes[0].shape == (geo["height"] + 2 * geo["margin_y"], geo["stride"])
        org = geo["margin_y"] * geo["stride"] + geo["margin_x"]
        L = prov.L
        L.x265hip_me_cache_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.x265hip_me_cache_surface.restype = ctypes.c_void_p
        L.x265hip_me_cache_surface.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.x265hip_me_cache_ready.restype = ctypes.c_void_p
        L.x265hip_me_cache_ready.argtypes = [ctypes.c_void_p, ctypes.c_int]
        nctu = (geo["width"] // 64) * (geo["height"] // 64)
        rows = geo["height"] // 64
        nc = 2 * rng + 1
        ng = (nc + 3) // 4
        zero = np.zeros(nc, np.uint16)
        for slot, (cur, ref) in enumerate([(1, 0), (2, 1)]):
            gen = L.x265hip_me_cache_submit(prov.handle, slot, planes[cur].ctypes.data, cur, planes[ref].ctypes.data)
            assert gen > 0
            flags = np.ctypeslib.as_array((ctypes.c_int * rows).from_address(L.x265hip_me_cache_ready(prov.handle, slot)))
            t0 = time.time()
            while not (flags == gen).all():
                assert time.time() - t0 < 60, "surfaces never arrived"
                time.sleep(0.002)
            surf, _ = O.me_fullsearch(depth, planes[cur], geo["stride"], org, planes[ref], geo["stride"], org, geo["width"], geo["height"], rng,
                                      0, nctu, zero, zero, want_surf=True, want_best=False)
            def cols(v):        # [ctu, dy, group, pu, 4] -> [ctu, dy, dx, pu] with the unspecified pad columns of the last group dropped
                return v.transpose(0, 1, 2, 4, 3).reshape(nctu, nc, ng * 4, v.shape[3])[:, :, :nc, :]
            e = cols(surf.reshape(nctu, nc, ng, 85, 4))
            gb = 720 if depth == 8 else 1360
            raw = np.ctypeslib.as_array((ctypes.c_uint8 * (nctu * nc * ng * gb)).from_address(L.x265hip_me_cache_surface(prov.handle, slot)))
            raw = raw.reshape(nctu, nc, ng, gb)
            if depth == 8:      # X265HIP_SURF_PACKED
                g8 = raw[..., 0:512].copy().view(np.uint16).reshape(nctu, nc, ng, 64, 4)
                g16 = raw[..., 512:640].copy().view(np.uint16).reshape(nctu, nc, ng, 16, 4)
                g32 = raw[..., 640:720].copy().view(np.int32).reshape(nctu, nc, ng, 5, 4)
                assert np.array_equal(cols(g8), e[..., 0:64]) and np.array_equal(cols(g16), e[..., 64:80]) and np.array_equal(cols(g32), e[..., 80:85])
            else:
                assert np.array_equal(cols(raw.copy().view(np.int32).reshape(nctu, nc, ng, 85, 4)), e)
        rep = prov.report()
        assert rep["fills"] == 2 and rep["failed"] == 0
    finally:
        prov.close()


@pytest.mark.parametrize("depth,preset,extra", [(8, "medium", []), (8, "slow", [("me", "star")]), (8, "slower", []), (10, "medium", [])])
def test_seam_encode_on_gpu_surfaces_is_byte_identical(depth, preset, extra):
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "1"), ("crf", "24"), ("no-weightp", None), ("no-weightb", None)] + extra
    # wait=True: a 256x192 picture is encoded faster than its surfaces travel; the test mode lets the lookups wait for their rows
    base, got, rep = T.run_pair(depth, 256, 192, 5, preset, opts, "gpu", rng=20, verify=True, wait=True)
    assert got[0] == base[0], f"seam changed the bitstream: {rep}"
    assert rep["verify"] == 1 and rep["verify_mismatches"] == 0 and rep["failed"] == 0
    assert 1 <= rep["fills"] <= rep["pair_submits"] and rep["pair_submits"] >= 4      # a pair superseded before its turn is skipped
    assert rep["lookups_served"] > 1500, rep


def test_seam_never_waits_by_default():
    """Production mode: a row that has not arrived is answered by the host primitive at once; the bitstream is the same either way."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "1"), ("crf", "24"), ("no-weightp", None), ("no-weightb", None)]
    base, got, rep = T.run_pair(8, 256, 192, 5, "medium", opts, "gpu", rng=20, verify=True)
    assert got[0] == base[0] and rep["verify_mismatches"] == 0 and rep["failed"] == 0
    assert rep["lookups_served"] + rep["row_not_ready"] + rep["outside_window"] > 1000
