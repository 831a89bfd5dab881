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


@pytest.mark.parametrize("depth,width,height,rng,fmt", [(8, 256, 192, 20, 1), (8, 200, 136, 57, 1), (10, 256, 128, 16, 0), (8, 256, 192, 20, 2), (8, 200, 136, 57, 2)])
def test_me_cache_surfaces_equal_oracle(depth, width, height, rng, fmt):
    from tools import seam_driver as SD
    O = _oracle()
    geo = SD.geometry(width, height)
    prov = SD.GpuProvider(depth, geo, rng, 2, fmt)
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
            if fmt == 2:        # X265HIP_SURF_PACKED_T: [chunk 45][group][16 B] inside a motion-vector row -> record-contiguous
                raw = raw.reshape(nctu, nc, 45, ng, 16).transpose(0, 1, 3, 2, 4).reshape(nctu, nc, ng, gb)
            else:
                raw = raw.reshape(nctu, nc, ng, gb)
            if depth == 8:
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


@pytest.mark.parametrize("depth,preset,extra,fmt", [(8, "medium", [], None), (8, "slow", [("me", "star")], None), (8, "slower", [], None), (10, "medium", [], None),
                                                    (8, "slow", [("me", "star")], 2), (8, "slower", [], 2)])
def test_seam_encode_on_gpu_surfaces_is_byte_identical(depth, preset, extra, fmt):
    """fmt 2 = X265HIP_SURF_PACKED_T: the lookups walk the chunk-major rows the record-per-lane kernel writes."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "1"), ("crf", "24"), ("no-weightp", None), ("no-weightb", None)] + extra
    # wait=True: a 256x192 picture is encoded faster than its surfaces travel; the test mode lets the lookups wait for their rows
    base, got, rep = T.run_pair(depth, 256, 192, 5, preset, opts, "gpu", rng=20, verify=True, wait=True, surf_format=fmt)
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


@pytest.mark.parametrize("depth,preset,extra", [(8, "medium", []), (8, "slow", []), (10, "medium", []), (8, "medium", [("bframes", "0")])])
def test_lookahead_seam_on_gpu_is_byte_identical(depth, preset, extra):
    """CostEstimateGroup::estimateFrameCost's block loop served by x265hip_lowres_cost_host inside the real encoder (P and B pictures,
    list reuse, weighted references, AQ weights): same slice decisions, same bitstream."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "1"), ("crf", "24"), ("lookahead-slices", "1")] + extra
    # gpu+verify: the oracle re-scores every (p0, b, p1) triple and every intra estimate from the same inputs, in flight
    base, got, rep = T.run_pair(depth, 320, 192, 12, preset, opts, "gpu", rng=16, verify=True, wait=True, lookahead="gpu+verify")
    la = rep["lookahead_seam"]
    assert la["verify_mismatches"] == 0, la
    assert got[0] == base[0], f"lookahead seam changed the bitstream: {rep}"
    assert la["frame_cost_estimates_served"] >= 10 and la["intra_estimates_served"] >= 12 and la["failed"] == 0, la
    assert rep["verify_mismatches"] == 0 and rep["failed"] == 0


def test_lookahead_seam_survives_a_second_encode_with_other_pictures_at_the_same_addresses():
    """Two encodes in one process: the allocator hands the second one the first one's Lowres addresses and the frame numbers repeat;
    the plane keys carry an encoder-instance number, so no device copy of the first clip is served to the second."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "1"), ("crf", "24"), ("lookahead-slices", "1")]
    for seed in (41, 42, 43, 41):
        base, got, rep = T.run_pair(8, 320, 192, 12, "medium", opts, "gpu", rng=16, verify=False, wait=True, lookahead="gpu+verify", seed=seed)
        assert rep["lookahead_seam"]["verify_mismatches"] == 0 and got[0] == base[0], (seed, rep["lookahead_seam"])


def test_lowres_cost_host_entry_equals_oracle():
    """x265hip_lowres_cost_host through plain host pointers (what the seam passes) against the oracle, P and B, both lists searched."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    depth, W, Hh = 8, 416, 240
    clip = F.synth_clip(W, Hh, 3, depth=depth, seed=57)
    lw, lh = ((W // 2 + 7) >> 3) * 8, ((Hh // 2 + 7) >> 3) * 8
    stride = (lw + 2 * F.MARGIN_X + 31) & ~31
    org = stride * F.MARGIN_Y + F.MARGIN_X
    rows = lh + 2 * F.MARGIN_Y
    planes = []
    for y, _, _ in clip:
        buf, st, og, _, _ = F.pad_plane(y)
        planes.append(O.lowres_init(depth, buf, st, og, stride, org, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y))
    wcu, hcu = lw // 8, lh // 8
    n = wcu * hcu
    icost, _, _ = O.lowres_intra(depth, planes[1][0], stride, org, wcu, hcu, 5)
    half = 4 * max(lw, lh) + 1024
    cq, qoff = F.qpel_cost_table(16, lam=1.0, qmax=half)
    assert qoff == half

    class HP(ctypes.Structure):
        _fields_ = [("depth", ctypes.c_int), ("stride", ctypes.c_ssize_t), ("width_in_cu", ctypes.c_int), ("height_in_cu", ctypes.c_int),
                    ("lines", ctypes.c_int), ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int), ("cur", ctypes.c_void_p),
                    ("ref", ctypes.c_void_p * 4), ("ref1", ctypes.c_void_p * 4), ("ref_bi", ctypes.c_void_p * 4),
                    ("intra_cost", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p), ("cost_q", ctypes.c_void_p), ("cost_q_half", ctypes.c_int),
                    ("bframe_bias", ctypes.c_int), ("do_search", ctypes.c_int * 2), ("mvs", ctypes.c_void_p * 2), ("mv_costs", ctypes.c_void_p * 2),
                    ("lowres_costs", ctypes.c_void_p), ("row_satds", ctypes.c_void_p), ("frame", ctypes.c_void_p),
                    ("plane_key_cur", ctypes.c_uint64), ("plane_key_ref", ctypes.c_uint64), ("plane_key_ref1", ctypes.c_uint64), ("plane_key_ref_bi", ctypes.c_uint64)]
    for bidir in (False, True):
        mvs = [np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32)]
        mvc = [np.zeros(n, np.int32), np.zeros(n, np.int32)]
        lc, rws, frame = np.zeros(n, np.uint16), np.zeros(hcu, np.int32), np.zeros(4, np.int64)
        q = HP()
        q.depth, q.stride, q.width_in_cu, q.height_in_cu, q.lines, q.margin_x, q.margin_y = depth, stride, wcu, hcu, lh, F.MARGIN_X, F.MARGIN_Y
        q.cur = planes[1][0].ctypes.data + org
        for i in range(4):
            q.ref[i] = planes[0][i].ctypes.data + org
            q.ref1[i] = planes[2][i].ctypes.data + org if bidir else None
        q.intra_cost, q.inv_qscale = icost.ctypes.data, None
        q.cost_q, q.cost_q_half, q.bframe_bias = cq.ctypes.data + 2 * qoff, half, 0
        q.do_search[0], q.do_search[1] = 1, int(bidir)
        for l in range(2):
            q.mvs[l], q.mv_costs[l] = mvs[l].ctypes.data, mvc[l].ctypes.data
        q.lowres_costs, q.row_satds, q.frame = lc.ctypes.data, rws.ctypes.data, frame.ctypes.data
        q.plane_key_cur, q.plane_key_ref, q.plane_key_ref1 = (2, 1, 3) if bidir else (0, 0, 0)      # second pass: through the shared plane cache
        f = A.lib().x265hip_lowres_cost_host
        f.argtypes = [ctypes.POINTER(HP)]
        A.check(f(ctypes.byref(q)), "x265hip_lowres_cost_host")
        exp = O.lowres_cost(depth, planes[1][0], planes[0], stride, org, wcu, hcu, cq, qoff, icost, ref1_planes=planes[2] if bidir else None)
        if bidir:
            (e0, e1), (c0, c1), elc, erows, eframe = exp
            assert np.array_equal(mvs[1], e1) and np.array_equal(mvc[1], c1)
        else:
            e0, c0, elc, erows, eframe = exp
        assert np.array_equal(mvs[0], e0) and np.array_equal(mvc[0], c0) and np.array_equal(lc, elc) and np.array_equal(rws, erows)
        assert np.array_equal(frame[:len(eframe)], eframe)


@pytest.mark.parametrize("depth,preset,extra", [(8, "slow", [("me", "star")]), (8, "slower", []), (10, "slow", []), (8, "medium", [])])
def test_subpel_seam_encode_on_gpu_phase_planes_is_byte_identical(depth, preset, extra):
    """The sub-sample seam on the product path: x265hip_phase_cache's planes serve MotionEstimate::subpelCompare of the real encoder;
    every served call is re-evaluated by the reference's own (interpolating) function in flight."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "1"), ("crf", "24")] + extra
    base, got, rep = T.run_pair(depth, 256, 192, 5, preset, opts, "gpu", rng=20, verify=True, wait=True, subpel="gpu")
    sub = rep["subpel_seam"]
    assert got[0] == base[0], f"seam changed the bitstream: {rep}"
    assert sub["verify_mismatches"] == 0 and rep["verify_mismatches"] == 0 and sub["failed"] == 0
    assert sub["subpel_compares_served"] > 500 and sub["fills"] >= 2, sub
