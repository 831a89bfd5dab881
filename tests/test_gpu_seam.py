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
from conftest import missing_reference_build

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


def test_me_cache_pairs_of_two_source_pictures_queued_back_to_back_keep_their_own_source():
    """Round-2 advice: submits of the NEXT source picture must not replace the samples a queued pair of the previous one still has to upload.
    Four pairs of three source pictures go in without waiting for any of them; each surface must be the one of ITS (source, reference)."""
    from tools import seam_driver as SD
    O = _oracle()
    depth, width, height, rng = 8, 256, 192, 12
    geo = SD.geometry(width, height)
    prov = SD.GpuProvider(depth, geo, rng, 4, 1)
    try:
        clip = F.synth_clip(width, height, 5, depth=depth, seed=57)
        planes = [F.pad_plane(y)[0] for (y, _, _) in clip]
        org = geo["margin_y"] * geo["stride"] + geo["margin_x"]
        L = prov.L
        L.x265hip_me_cache_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.x265hip_me_cache_surface.restype = ctypes.c_void_p
        L.x265hip_me_cache_surface.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.x265hip_me_cache_ready.restype = ctypes.c_void_p
        L.x265hip_me_cache_ready.argtypes = [ctypes.c_void_p, ctypes.c_int]
        nctu, rows, nc = (geo["width"] // 64) * (geo["height"] // 64), geo["height"] // 64, 2 * rng + 1
        ng = (nc + 3) // 4
        pairs = [(1, 0), (2, 1), (2, 0), (3, 2)]                 # (source, reference): three source pictures, the middle one twice
        gens = []
        for slot, (cur, ref) in enumerate(pairs):                # a fresh copy per call: the caller's buffers need not outlive it
            a, b = planes[cur].copy(), planes[ref].copy()
            gens.append(L.x265hip_me_cache_submit(prov.handle, slot, a.ctypes.data, 100 + cur, b.ctypes.data))
            a[:] = 0; b[:] = 0
        assert all(g > 0 for g in gens)
        zero = np.zeros(nc, np.uint16)
        for slot, (cur, ref) in enumerate(pairs):
            flags = np.ctypeslib.as_array((ctypes.c_int * rows).from_address(L.x265hip_me_cache_ready(prov.handle, slot)))
            t0 = time.time()
            while not (flags == gens[slot]).all():
                assert time.time() - t0 < 60, "surfaces never arrived"
                time.sleep(0.002)
            surf, _ = O.me_fullsearch(depth, planes[cur], geo["stride"], org, planes[ref], geo["stride"], org, geo["width"], geo["height"], rng,
                                      0, nctu, zero, zero, want_surf=True, want_best=False)
            e = surf.reshape(nctu, nc, ng, 85, 4).transpose(0, 1, 2, 4, 3).reshape(nctu, nc, ng * 4, 85)[:, :, :nc, :]
            raw = np.ctypeslib.as_array((ctypes.c_uint8 * (nctu * nc * ng * 720)).from_address(L.x265hip_me_cache_surface(prov.handle, slot))).reshape(nctu, nc, ng, 720)
            g8 = raw[..., 0:512].copy().view(np.uint16).reshape(nctu, nc, ng, 64, 4).transpose(0, 1, 2, 4, 3).reshape(nctu, nc, ng * 4, 64)[:, :, :nc, :]
            g32 = raw[..., 640:720].copy().view(np.int32).reshape(nctu, nc, ng, 5, 4).transpose(0, 1, 2, 4, 3).reshape(nctu, nc, ng * 4, 5)[:, :, :nc, :]
            assert np.array_equal(g8, e[..., 0:64]) and np.array_equal(g32, e[..., 80:85]), f"pair {slot}: surfaces of another source picture"
        rep = prov.report()
        assert rep["fills"] == 4 and rep["failed"] == 0
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
    A.lib().x265hip_lowres_planes_forget()          # keys 1..3 name THIS test's pictures
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
        if not bidir:
            continue
        # round 6: the searched vectors stay on the device under the picture's key.  The same triple again with NEITHER list searched (the dependency-free launch reads
        # the resident copies), then list 0 reused against another list-1 picture with arrays of its own: the oracle's answers for the same requests
        lc[:], rws[:], frame[:] = 0, 0, 0
        q.do_search[0], q.do_search[1] = 0, 0
        A.check(f(ctypes.byref(q)), "x265hip_lowres_cost_host")
        assert np.array_equal(mvs[0], e0) and np.array_equal(mvs[1], e1) and np.array_equal(lc, elc) and np.array_equal(rws, erows) and np.array_equal(frame[:len(eframe)], eframe)
        mv1b, mc1b = np.zeros((n, 2), np.int32), np.zeros(n, np.int32)
        q.mvs[1], q.mv_costs[1] = mv1b.ctypes.data, mc1b.ctypes.data
        for i in range(4):
            q.ref1[i] = planes[0][i].ctypes.data + org
        q.plane_key_ref1 = 1
        q.do_search[0], q.do_search[1] = 0, 1
        lc[:], rws[:], frame[:] = 0, 0, 0
        A.check(f(ctypes.byref(q)), "x265hip_lowres_cost_host")
        (_, f1), (_, fc1), flc, frows, fframe = O.lowres_cost(depth, planes[1][0], planes[0], stride, org, wcu, hcu, cq, qoff, icost, ref1_planes=planes[0],
                                                              do_search=(0, 1), mvs_in=(e0, None), mv_costs_in=(c0, None))
        assert np.array_equal(mvs[0], e0) and np.array_equal(mv1b, f1) and np.array_equal(mc1b, fc1)
        assert np.array_equal(lc, flc) and np.array_equal(rws, frows) and np.array_equal(frame[:len(fframe)], fframe)
        q.do_search[0], q.do_search[1] = 0, 0
        lc[:], rws[:], frame[:] = 0, 0, 0
        A.check(f(ctypes.byref(q)), "x265hip_lowres_cost_host")
        assert np.array_equal(lc, flc) and np.array_equal(rws, frows) and np.array_equal(frame[:len(fframe)], fframe)
        # a NEW key for the current picture at the SAME addresses (the Lowres was recycled: other vectors in the same arrays): nothing resident may be served -
        # the arrays' present content is what counts.  List 1 now holds the FIRST estimate's list-1 vectors (searched against planes[2]) next to planes[0].
        mv1b[:], mc1b[:] = e1, c1
        q.plane_key_cur = 9
        lc[:], rws[:], frame[:] = 0, 0, 0
        A.check(f(ctypes.byref(q)), "x265hip_lowres_cost_host")
        _, _, glc, grows, gframe = O.lowres_cost(depth, planes[1][0], planes[0], stride, org, wcu, hcu, cq, qoff, icost, ref1_planes=planes[0],
                                                 do_search=(0, 0), mvs_in=(e0, e1), mv_costs_in=(c0, c1))
        assert not np.array_equal(glc, flc)
        assert np.array_equal(lc, glc) and np.array_equal(rws, grows) and np.array_equal(frame[:len(gframe)], gframe)
    A.lib().x265hip_lowres_planes_forget()


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


# ---- round 3: the ROW-GRANULAR services (csrc/me_stream.hip, csrc/phase_stream.hip) -------------------------------------------------
def _stream_entries(L):
    L.x265hip_me_stream_picture_rows.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.x265hip_me_stream_pair_open.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]
    L.x265hip_me_stream_pair_open_weighted.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    L.x265hip_me_stream_surface.restype = ctypes.c_void_p
    L.x265hip_me_stream_surface.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.x265hip_me_stream_ready.restype = ctypes.c_void_p
    L.x265hip_me_stream_ready.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.x265hip_me_stream_record_bytes.argtypes = [ctypes.c_void_p]


def _check_stream_surfaces(O, L, prov, slot, depth, geo, rng, min_level, fenc_plane, ref_plane):
    """slot's downloaded records (whole, or the 16x16-and-up tails) == the oracle's exhaustive search of fenc_plane in ref_plane"""
    ctus_w, ctus_h = geo["width"] // 64, geo["height"] // 64
    nc = 2 * rng + 1
    ng = (nc + 3) // 4
    rec = L.x265hip_me_stream_record_bytes(prov.handle)
    org = geo["margin_y"] * geo["stride"] + geo["margin_x"]
    zero = np.zeros(nc, np.uint16)
    surf, _ = O.me_fullsearch(depth, fenc_plane, geo["stride"], org, ref_plane, geo["stride"], org, geo["width"], geo["height"], rng,
                              0, ctus_w * ctus_h, zero, zero, want_surf=True, want_best=False)
    e = surf.reshape(-1, 85, 4)
    nrec = e.shape[0]
    raw = np.ctypeslib.as_array((ctypes.c_uint8 * (nrec * rec)).from_address(L.x265hip_me_stream_surface(prov.handle, slot))).reshape(nrec, rec)
    def cols(v):        # [record, pu, 4] -> [motion-vector row, dx, pu]; the pad columns of a row's last group hold don't-care values
        v = v.reshape(-1, ng, v.shape[1], 4)
        return v.transpose(0, 1, 3, 2).reshape(v.shape[0], ng * 4, v.shape[2])[:, :nc]
    if depth == 8:
        tail = raw if min_level else raw[:, 512:]
        g16 = tail[:, 0:128].copy().view(np.uint16).reshape(nrec, 16, 4)
        g32 = tail[:, 128:208].copy().view(np.int32).reshape(nrec, 5, 4)
        assert np.array_equal(cols(g16), cols(e[:, 64:80])) and np.array_equal(cols(g32), cols(e[:, 80:85]))
        if not min_level:
            assert np.array_equal(cols(raw[:, 0:512].copy().view(np.uint16).reshape(nrec, 64, 4)), cols(e[:, 0:64]))
    else:
        got = raw.copy().view(np.int32).reshape(nrec, -1, 4)
        assert np.array_equal(cols(got), cols(e[:, 64:] if min_level else e))


@pytest.mark.parametrize("depth,width,height,rng,min_level,band_rows", [(8, 256, 320, 20, 0, 2), (8, 256, 320, 20, 1, 8), (10, 192, 256, 16, 1, 3),
                                                                        (8, 200, 264, 57, 1, 1), (12, 128, 192, 12, 0, 8)])
def test_me_stream_surfaces_equal_oracle_whatever_order_the_rows_arrive_in(depth, width, height, rng, min_level, band_rows):
    """Rows of the reference picture handed over one by one (and out of order), the pair opened in the middle: every CTU row's records
    - whole, or the 16x16-and-up tail - must equal the oracle's exhaustive search, and a row may only be flagged once the rows its
    window reaches have been handed over."""
    from tools import seam_driver as SD
    O = _oracle()
    geo = SD.geometry(width, height)
    clip = F.synth_clip(geo["width"], geo["height"], 3, depth=depth, seed=7)
    dt = np.uint8 if depth == 8 else np.uint16
    def padded(y):      # the PicYuv layout: edges replicated into the margins
        return np.ascontiguousarray(np.pad(y.reshape(geo["height"], geo["width"]).astype(dt), ((geo["margin_y"],) * 2, (geo["margin_x"],) * 2), mode="edge")).reshape(-1)
    planes = [padded(fr[0]) for fr in clip]
    prov = SD.StreamGpuProvider(depth, geo, rng, slots=3, min_level=min_level, pictures=6, band_rows=band_rows)
    L = prov.L
    _stream_entries(L)
    try:
        ctus_w, ctus_h = geo["width"] // 64, geo["height"] // 64
        rec = L.x265hip_me_stream_record_bytes(prov.handle)
        full = 720 if depth == 8 else 1360
        assert rec == (full if not min_level else 208 if depth == 8 else 336)
        lag = (63 + rng) // 64
        for slot, (fi, ri, order) in enumerate([(1, 0, list(range(ctus_h))), (2, 1, [1, 0] + list(range(2, ctus_h))), (2, 0, list(range(ctus_h))[::-1])]):
            fkey, rkey = 100 + 10 * slot + fi, 200 + 10 * slot + ri
            assert L.x265hip_me_stream_picture_rows(prov.handle, fkey, planes[fi].ctypes.data, 0, ctus_h) == 0
            ready = np.ctypeslib.as_array((ctypes.c_int32 * ctus_h).from_address(L.x265hip_me_stream_ready(prov.handle, slot)))
            gen = None
            given = set()
            for i, r in enumerate(order):
                if i == 1:                  # the pair is opened after the first row of the reference is already there
                    gen = L.x265hip_me_stream_pair_open(prov.handle, slot, fkey, rkey)
                    assert gen > 0
                assert L.x265hip_me_stream_picture_rows(prov.handle, rkey, planes[ri].ctypes.data, r, 1) == 0
                given.add(r)
                time.sleep(0.02)
                if gen:
                    for q in range(ctus_h):     # no row is flagged before the rows its window reaches were handed over
                        if ready[q] == gen:
                            assert all(k in given for k in range(max(0, q - lag), min(ctus_h, q + lag + 1))), (q, sorted(given))
            if gen is None:
                gen = L.x265hip_me_stream_pair_open(prov.handle, slot, fkey, rkey)
            t0 = time.time()
            while not all(ready[q] == gen for q in range(ctus_h)) and time.time() - t0 < 20:
                time.sleep(0.01)
            assert all(ready[q] == gen for q in range(ctus_h)), (list(ready), gen, prov.report())
            _check_stream_surfaces(O, L, prov, slot, depth, geo, rng, min_level, planes[fi], planes[ri])
        rep = prov.report()
        assert rep["failed"] == 0 and rep["pairs_completed"] == 3 and rep["rows_searched"] == 3 * ctus_h, rep
    finally:
        prov.close()


@pytest.mark.parametrize("depth,width,height", [(8, 256, 320), (10, 192, 192), (12, 128, 128)])
def test_phase_stream_planes_equal_the_picture_granular_planes(depth, width, height):
    """Rows handed over one by one: progress only ever covers finished lines, and the finished planes equal x265hip_phase_planes run
    on the whole picture (itself pinned on the oracle's interpolation primitives, test_gpu_phase_planes.py)."""
    import torch
    from tools import seam_driver as SD
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    geo = SD.geometry(width, height)
    dt = np.uint8 if depth == 8 else np.uint16
    rng = np.random.default_rng(3 + depth)
    src = [rng.integers(0, 1 << depth, (geo["rows"], geo["stride"])).astype(dt), rng.integers(0, 1 << depth, (geo["rows_c"], geo["stride_c"])).astype(dt),
           rng.integers(0, 1 << depth, (geo["rows_c"], geo["stride_c"])).astype(dt)]
    prov = SD.StreamGpuPhaseProvider(depth, geo, slots=2)
    L = prov.L
    L.x265hip_phase_stream_open.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.x265hip_phase_stream_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.x265hip_phase_stream_planes.restype = ctypes.c_void_p
    L.x265hip_phase_stream_planes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.x265hip_phase_stream_progress.restype = ctypes.c_void_p
    L.x265hip_phase_stream_progress.argtypes = [ctypes.c_void_p, ctypes.c_int]
    try:
        ctu_rows = geo["height"] // 64
        for slot in (1, 0):
            gen = L.x265hip_phase_stream_open(prov.handle, slot)
            assert gen > 0
            prog = np.ctypeslib.as_array((ctypes.c_uint64 * 2).from_address(L.x265hip_phase_stream_progress(prov.handle, slot)))
            for r in range(ctu_rows):
                assert L.x265hip_phase_stream_rows(prov.handle, slot, gen, src[0].ctypes.data, src[1].ctypes.data, src[2].ctypes.data, r, 1) == 0
                time.sleep(0.02)
                have = geo["rows"] if r == ctu_rows - 1 else geo["margin_y"] + (r + 1) * 64
                assert int(prog[0]) >> 32 in (0, gen) and (int(prog[0]) & 0xffffffff) <= have - 8
            t0 = time.time()
            want = [(gen << 32) | (geo["rows"] - 8), (gen << 32) | (geo["rows_c"] - 8)]
            while (int(prog[0]) != want[0] or int(prog[1]) != want[1]) and time.time() - t0 < 20:
                time.sleep(0.01)
            assert [int(prog[0]), int(prog[1])] == want, prov.report()
            for pl in range(3):
                k = min(pl, 1)
                rows, stride, nph = (geo["rows"], geo["stride"], 15) if k == 0 else (geo["rows_c"], geo["stride_c"], 63)
                n = nph * rows * stride
                got = np.ctypeslib.as_array((ctypes.c_uint8 * (n * dt().itemsize)).from_address(L.x265hip_phase_stream_planes(prov.handle, slot, pl))).view(dt).reshape(nph, rows, stride)
                dev = torch.device("cuda:0")
                tsrc = torch.from_numpy(src[pl].view(np.int16 if depth > 8 else np.uint8)).to(dev)
                tdst = torch.zeros(n, dtype=tsrc.dtype, device=dev)
                A.phase_planes(depth, tsrc, 0, tdst, stride, rows, chroma=bool(k))
                torch.cuda.synchronize()
                exp = tdst.cpu().numpy().view(dt).reshape(nph, rows, stride)
                assert np.array_equal(got[:, 4:rows - 8, 8:stride - 8], exp[:, 4:rows - 8, 8:stride - 8]), (slot, pl)
        rep = prov.report()
        assert rep["failed"] == 0 and rep["completed"] == 2, rep
    finally:
        prov.close()


@pytest.mark.parametrize("depth,preset,ft,min_level,extra", [(8, "medium", 3, 0, []), (8, "slow", 3, 1, [("me", "star")]), (8, "slower", 2, 1, []), (10, "medium", 3, 1, [])])
def test_row_granular_seams_on_the_gpu_serve_under_frame_threads(depth, preset, ft, min_level, extra):
    """The real encoder at --frame-threads > 1 on x265hip_me_stream + x265hip_phase_stream: byte-identical, every SAD lookup and every
    sub-sample comparison verified in flight against the reference's own functions."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", str(ft)), ("crf", "24"), ("no-weightp", None), ("no-weightb", None)] + extra
    base, got, rep = T.run_pair(depth, 256, 192, 8, preset, opts, "gpu", rng=20, verify=True, wait=True, streamed=True, min_level=min_level, subpel="gpu",
                                slots=24, subpel_slots=12)
    assert got[0] == base[0], f"seam changed the bitstream: {rep}"
    sub = rep["subpel_seam"]
    assert rep["verify_mismatches"] == 0 and sub["verify_mismatches"] == 0 and rep["failed"] == 0 and sub["failed"] == 0
    assert rep["lookups_served"] > (300 if min_level else 1500) and sub["subpel_compares_served"] > 1000, rep
    assert rep["row_stream"]["recon_rows_refused"] == 0 and rep["stale_pairs"] == 0


# ---- round 4: WEIGHTED references through the row-granular services ----------------------------------------------------------------
@pytest.mark.parametrize("depth,width,height,rng,min_level", [(8, 256, 256, 20, 1), (8, 192, 192, 16, 0), (10, 192, 256, 16, 1), (12, 128, 192, 12, 0)])
def test_me_stream_weighted_pairs_search_the_plane_the_host_would_weight(depth, width, height, rng, min_level):
    """x265hip_me_stream_pair_open_weighted: the reference picture's rows are weighted on the device with the weight_pp arguments
    (pixel.cpp:518-543; reference.cpp:119-178 builds that plane on the host) - the pair's surfaces must equal the oracle's exhaustive
    search in the plane weighted on the CPU, an unweighted pair of the same pictures stays what it was, and two weights of one
    reference are two planes."""
    from tools import seam_driver as SD
    O = _oracle()
    geo = SD.geometry(width, height)
    clip = F.synth_clip(geo["width"], geo["height"], 2, depth=depth, seed=11, fade=(1.0, 0.6))
    dt = np.uint8 if depth == 8 else np.uint16
    def padded(y):
        return np.ascontiguousarray(np.pad(y.reshape(geo["height"], geo["width"]).astype(dt), ((geo["margin_y"],) * 2, (geo["margin_x"],) * 2), mode="edge")).reshape(-1)
    planes = [padded(fr[0]) for fr in clip]
    prov = SD.StreamGpuProvider(depth, geo, rng, slots=3, min_level=min_level, pictures=6, band_rows=2)
    L = prov.L
    _stream_entries(L)
    try:
        ctus_h = geo["height"] // 64
        corr = 14 - depth
        weights = [(38, (1 << 5) << corr, 6 + corr, 6 << (depth - 8)), (90, (1 << 6) << corr, 7 + corr, -(3 << (depth - 8)))]
        assert L.x265hip_me_stream_picture_rows(prov.handle, 1, planes[1].ctypes.data, 0, ctus_h) == 0
        gens = []
        for slot, w in enumerate([weights[0], None, weights[1]]):
            wbuf = (ctypes.c_int * 4)(*w) if w else None
            gens.append(L.x265hip_me_stream_pair_open_weighted(prov.handle, slot, 1, 2, wbuf))
            assert gens[-1] > 0
        for r in list(range(ctus_h))[::-1]:          # rows of the reference arrive after the pairs were opened, bottom up
            assert L.x265hip_me_stream_picture_rows(prov.handle, 2, planes[0].ctypes.data, r, 1) == 0
        for slot, w in enumerate([weights[0], None, weights[1]]):
            ready = np.ctypeslib.as_array((ctypes.c_int32 * ctus_h).from_address(L.x265hip_me_stream_ready(prov.handle, slot)))
            t0 = time.time()
            while not all(ready[q] == gens[slot] for q in range(ctus_h)) and time.time() - t0 < 20:
                time.sleep(0.01)
            assert all(ready[q] == gens[slot] for q in range(ctus_h)), (list(ready), prov.report())
            ref_plane = planes[0] if w is None else SD.weight_plane(planes[0], depth, w)
            _check_stream_surfaces(O, L, prov, slot, depth, geo, rng, min_level, planes[1], ref_plane)
        rep = prov.report()
        assert rep["failed"] == 0 and rep["weighted_pairs"] == 2 and rep["rows_weighted"] == 2 * ctus_h and rep["rows_uploaded"] == 2 * ctus_h, rep
        # bad arguments are refused: the same key twice, a shift without the 14 - depth correction
        assert L.x265hip_me_stream_pair_open_weighted(prov.handle, 0, 5, 5, None) < 0
        assert L.x265hip_me_stream_pair_open_weighted(prov.handle, 0, 1, 2, (ctypes.c_int * 4)(64, 0, corr - 1, 0)) < 0
    finally:
        prov.close()


@pytest.mark.parametrize("depth,width,height", [(8, 256, 256), (10, 192, 192)])
def test_phase_stream_weighted_view_equals_the_phase_planes_of_the_weighted_picture(depth, width, height):
    """x265hip_phase_stream_view_open with weights: a view opened BEFORE the rows arrive and one opened after them, luma and Cb weighted,
    Cr as reconstructed - every plane equals x265hip_phase_planes of the plane weighted on the CPU; an unweighted view of the same
    picture beside them is untouched."""
    import torch
    from tools import seam_driver as SD
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    geo = SD.geometry(width, height)
    dt = np.uint8 if depth == 8 else np.uint16
    rng = np.random.default_rng(13 + depth)
    src = [rng.integers(0, 1 << depth, (geo["rows"], geo["stride"])).astype(dt), rng.integers(0, 1 << depth, (geo["rows_c"], geo["stride_c"])).astype(dt),
           rng.integers(0, 1 << depth, (geo["rows_c"], geo["stride_c"])).astype(dt)]
    prov = SD.StreamGpuPhaseProvider(depth, geo, slots=3, pictures=2)
    L = prov.L
    L.x265hip_phase_stream_view_open.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint]
    L.x265hip_phase_stream_picture_rows.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.x265hip_phase_stream_planes.restype = ctypes.c_void_p
    L.x265hip_phase_stream_planes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.x265hip_phase_stream_progress.restype = ctypes.c_void_p
    L.x265hip_phase_stream_progress.argtypes = [ctypes.c_void_p, ctypes.c_int]
    try:
        ctu_rows = geo["height"] // 64
        corr = 14 - depth
        w3 = [(45, (1 << 5) << corr, 6 + corr, 4 << (depth - 8)), (70, (1 << 5) << corr, 6 + corr, -(2 << (depth - 8))), (64, 0, corr, 0)]
        wbuf = (ctypes.c_int * 12)(*[v for w in w3 for v in w])
        gens = {0: L.x265hip_phase_stream_view_open(prov.handle, 0, 77, wbuf, 3), 1: L.x265hip_phase_stream_view_open(prov.handle, 1, 77, None, 0)}
        for r in range(ctu_rows):
            assert L.x265hip_phase_stream_picture_rows(prov.handle, 77, src[0].ctypes.data, src[1].ctypes.data, src[2].ctypes.data, r, 1) == 0
            time.sleep(0.01)
        gens[2] = L.x265hip_phase_stream_view_open(prov.handle, 2, 77, wbuf, 1)          # after the fact, luma only
        assert all(g > 0 for g in gens.values())
        masks = {0: 3, 1: 0, 2: 1}
        for slot in (0, 1, 2):
            prog = np.ctypeslib.as_array((ctypes.c_uint64 * 2).from_address(L.x265hip_phase_stream_progress(prov.handle, slot)))
            want = [(gens[slot] << 32) | (geo["rows"] - 8), (gens[slot] << 32) | (geo["rows_c"] - 8)]
            t0 = time.time()
            while (int(prog[0]) != want[0] or int(prog[1]) != want[1]) and time.time() - t0 < 20:
                time.sleep(0.01)
            assert [int(prog[0]), int(prog[1])] == want, prov.report()
            for pl in range(3):
                k = min(pl, 1)
                rows, stride, nph = (geo["rows"], geo["stride"], 15) if k == 0 else (geo["rows_c"], geo["stride_c"], 63)
                n = nph * rows * stride
                got = np.ctypeslib.as_array((ctypes.c_uint8 * (n * dt().itemsize)).from_address(L.x265hip_phase_stream_planes(prov.handle, slot, pl))).view(dt).reshape(nph, rows, stride)
                plane = SD.weight_plane(src[pl], depth, w3[pl]) if (masks[slot] >> pl) & 1 else src[pl]
                dev = torch.device("cuda:0")
                tsrc = torch.from_numpy(np.ascontiguousarray(plane).view(np.int16 if depth > 8 else np.uint8)).to(dev)
                tdst = torch.zeros(n, dtype=tsrc.dtype, device=dev)
                A.phase_planes(depth, tsrc, 0, tdst, stride, rows, chroma=bool(k))
                torch.cuda.synchronize()
                exp = tdst.cpu().numpy().view(dt).reshape(nph, rows, stride)
                assert np.array_equal(got[:, 4:rows - 8, 8:stride - 8], exp[:, 4:rows - 8, 8:stride - 8]), (slot, pl)
        rep = prov.report()
        assert rep["failed"] == 0 and rep["completed"] == 3 and rep["weighted_views"] == 2, rep
    finally:
        prov.close()


@pytest.mark.parametrize("depth,preset,ft,min_level,extra", [(8, "slow", 3, 1, [("me", "star"), ("bframes", "0")]), (8, "medium", 3, 0, [("weightb", None)]),
                                                             (10, "medium", 2, 1, [("bframes", "0")]), (8, "slower", 2, 1, [])])
def test_row_granular_seams_on_the_gpu_serve_weighted_references_on_a_fade(depth, preset, ft, min_level, extra):
    """The real encoder with --weightp / --weightb at their defaults on a luma fade, x265hip_me_stream + x265hip_phase_stream weighting the
    reconstructed rows on the device: byte-identical, every SAD lookup and sub-sample comparison on a weighted reference re-evaluated
    in flight by the reference's own function on the host's weighted plane (round-3 verdict, next 1)."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", str(ft)), ("crf", "24")] + extra
    base, got, rep = T.run_fade_pair(depth, 256, 192, 10, preset, opts, "gpu", rng=20, wait=True, min_level=min_level, subpel="gpu", slots=32, subpel_slots=16)
    assert got[0] == base[0], f"seam changed the bitstream: {rep}"
    sub, wr = rep["subpel_seam"], rep["weighted_references"]
    assert rep["verify"] == 1 and rep["verify_mismatches"] == 0 and sub["verify_mismatches"] == 0 and rep["failed"] == 0 and sub["failed"] == 0
    assert wr["pairs_opened_on_weighted_references"] >= 1 and wr["lookups_served_on_weighted_references"] > 200, rep
    assert wr["phase_views_opened_on_weighted_references"] >= 1 and wr["subpel_compares_served_from_weighted_views"] > 500, rep
    assert rep["weighted_pairs"] >= 1 and rep["rows_weighted"] >= 3 and sub["weighted_views"] >= 1, rep


# ---- round 4: PU-major planes and windows centred on each CTU's displacement ---------------------------------------------------------
@pytest.mark.parametrize("depth,width,height,rng,min_level,centre,band_rows", [(8, 256, 256, 12, 1, 0, 2), (8, 256, 256, 12, 0, 40, 8), (10, 192, 256, 10, 1, 32, 3),
                                                                               (12, 128, 192, 8, 0, 24, 1), (8, 320, 192, 16, 1, 57, 2), (8, 256, 256, 12, 2, 40, 2),
                                                                               (10, 192, 256, 10, 2, 0, 3)])
def test_me_stream_planes_layout_and_centres_equal_the_oracle(depth, width, height, rng, min_level, centre, band_rows):
    """X265HIP_STREAM_PLANES: the slot buffer holds one raster per PU (uint16 saturating / uint32) - compared byte for byte with the
    oracle's records transposed by the checker's twin; centre_range: centres = the clamped displacement of each CTU's 64x64 minimum in
    the oracle's +-centre_range search, and every CTU's rasters are the oracle's search of the window around ITS centre."""
    from tools import seam_driver as SD
    O = _oracle()
    geo = SD.geometry(width, height)
    clip = F.synth_clip(geo["width"], geo["height"], 4, depth=depth, seed=17)
    dt = np.uint8 if depth == 8 else np.uint16
    def padded(y):
        return np.ascontiguousarray(np.pad(y.reshape(geo["height"], geo["width"]).astype(dt), ((geo["margin_y"],) * 2, (geo["margin_x"],) * 2), mode="edge")).reshape(-1)
    planes = [padded(fr[0]) for fr in clip]
    prov = SD.StreamGpuProvider(depth, geo, rng, slots=2, min_level=min_level, pictures=6, band_rows=band_rows, layout=SD.LAYOUT_PLANES, centre_range=centre)
    L = prov.L
    _stream_entries(L)
    L.x265hip_me_stream_centres.restype = ctypes.c_void_p
    L.x265hip_me_stream_centres.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.x265hip_me_stream_ctu_bytes.restype = ctypes.c_size_t
    L.x265hip_me_stream_ctu_bytes.argtypes = [ctypes.c_void_p]
    try:
        ctus_w, ctus_h = geo["width"] // 64, geo["height"] // 64
        nctu = ctus_w * ctus_h
        cb = SD.planes_ctu_bytes(rng, min_level)
        assert L.x265hip_me_stream_ctu_bytes(prov.handle) == cb and L.x265hip_me_stream_record_bytes(prov.handle) == 0
        org = geo["margin_y"] * geo["stride"] + geo["margin_x"]
        zero = np.zeros(2 * rng + 1, np.uint16)
        for slot, (fi, ri) in enumerate([(3, 0), (1, 0)]):             # three pictures / one picture apart: (9, 6) and (3, 2) of global motion
            assert L.x265hip_me_stream_picture_rows(prov.handle, 10 + fi, planes[fi].ctypes.data, 0, ctus_h) == 0
            gen = L.x265hip_me_stream_pair_open(prov.handle, slot, 10 + fi, 20 + ri + 5 * slot)
            assert gen > 0
            for r in range(ctus_h):
                assert L.x265hip_me_stream_picture_rows(prov.handle, 20 + ri + 5 * slot, planes[ri].ctypes.data, r, 1) == 0
            ready = np.ctypeslib.as_array((ctypes.c_int32 * ctus_h).from_address(L.x265hip_me_stream_ready(prov.handle, slot)))
            t0 = time.time()
            while not all(ready[q] == gen for q in range(ctus_h)) and time.time() - t0 < 20:
                time.sleep(0.01)
            assert all(ready[q] == gen for q in range(ctus_h)), prov.report()
            cen = np.zeros((nctu, 2), np.int16)
            cptr = L.x265hip_me_stream_centres(prov.handle, slot)
            if centre:
                zc = np.zeros(2 * centre + 1, np.uint16)
                _, best = O.me_fullsearch(depth, planes[fi], geo["stride"], org, planes[ri], geo["stride"], org, geo["width"], geo["height"], centre, 0, nctu, zc, zc,
                                          want_surf=False, want_best=True)
                idx = (best.reshape(-1, 85)[:, 84] & 0xffffffff).astype(np.int64)
                ncb = 2 * centre + 1
                mx, my = min(centre, geo["margin_x"] - rng - 12), min(centre, geo["margin_y"] - rng - 12)
                cen = np.stack([np.clip(idx % ncb - centre, -mx, mx), np.clip(idx // ncb - centre, -my, my)], axis=1).astype(np.int16)
                got_c = np.ctypeslib.as_array((ctypes.c_int16 * (2 * nctu)).from_address(cptr)).reshape(nctu, 2)
                assert np.array_equal(got_c, cen), (got_c[:6], cen[:6])
                assert np.abs(cen).max() > 0                                 # the clip moves: a centre of (0, 0) everywhere would test nothing
            else:
                assert not cptr
            parts = []
            for c in range(nctu):
                o = org + (c // ctus_w) * 64 * geo["stride"] + (c % ctus_w) * 64
                sc, _ = O.me_fullsearch(depth, planes[fi], geo["stride"], o, planes[ri], geo["stride"], o + int(cen[c, 1]) * geo["stride"] + int(cen[c, 0]),
                                        64, 64, rng, 0, 1, zero, zero, want_surf=True, want_best=False)
                parts.append(sc)
            exp = SD.records_to_planes(np.concatenate(parts).reshape(-1, 85, 4), nctu, rng, min_level)
            got = np.ctypeslib.as_array((ctypes.c_uint8 * (nctu * cb)).from_address(L.x265hip_me_stream_surface(prov.handle, slot))).reshape(nctu, cb)
            nc, pitch = 2 * rng + 1, 4 * ((2 * rng + 4) // 4)
            def valid(b):       # the pad columns of a raster row hold don't-care values
                nlo = (0 if min_level else 64) + (0 if min_level > 1 else 16)      # min_level 2: the 32x32 / 64x64 rasters only
                lo = b[:, :nc * pitch * 2 * nlo].copy().view(np.uint16).reshape(nctu, nlo, nc, pitch)[..., :nc]
                hi = b[:, nc * pitch * 2 * nlo:].copy().view(np.uint32).reshape(nctu, 5, nc, pitch)[..., :nc]
                return lo, hi
            (gl, gh), (el, eh) = valid(got), valid(exp)
            assert np.array_equal(gl, el) and np.array_equal(gh, eh), slot
        rep = prov.report()
        assert rep["failed"] == 0 and rep["pairs_completed"] == 2, rep
    finally:
        prov.close()


@pytest.mark.parametrize("depth,shift", [(8, 56), (10, -56)])
def test_me_stream_centred_windows_wait_for_every_reference_row_they_reach(depth, shift):
    """Round-4 advisor (high): with centre_range the window of a CTU row reaches 63 + max(centre_range, max |cy| + range) lines past the row's
    first line - at range 12, centre_range 57, margin 80 that is 131 lines = TWO CTU rows - and the service used to launch a row once
    rows r - 1 .. r + 1 were on the device.  The reference here is the source moved 56 lines (so every centre is (0, +-56)), its rows arrive
    one at a time into a recycled picture entry: no row may be flagged before every row its searches read was handed over, and the
    rasters must equal the oracle's search of the complete picture."""
    from tools import seam_driver as SD
    O = _oracle()
    rng, centre = 12, 57
    geo = SD.geometry(192, 448)
    clip = F.synth_clip(geo["width"], geo["height"], 2, depth=depth, seed=23)
    dt = np.uint8 if depth == 8 else np.uint16
    def padded(y):
        return np.ascontiguousarray(np.pad(y.reshape(geo["height"], geo["width"]).astype(dt), ((geo["margin_y"],) * 2, (geo["margin_x"],) * 2), mode="edge")).reshape(-1)
    cur = clip[1][0].reshape(geo["height"], geo["width"])
    moved = np.roll(cur, shift, axis=0)                      # what sits at line y of the source sits at line y + shift of the reference
    planes = [padded(moved), padded(cur)]
    stale = padded(clip[0][0])                               # what a recycled entry still holds
    prov = SD.StreamGpuProvider(depth, geo, rng, slots=1, min_level=1, pictures=2, band_rows=8, layout=SD.LAYOUT_PLANES, centre_range=centre)
    L = prov.L
    _stream_entries(L)
    L.x265hip_me_stream_centres.restype = ctypes.c_void_p
    L.x265hip_me_stream_centres.argtypes = [ctypes.c_void_p, ctypes.c_int]
    try:
        ctus_w, ctus_h = geo["width"] // 64, geo["height"] // 64
        nctu = ctus_w * ctus_h
        my = min(centre, geo["margin_y"] - rng - 12)
        lag = (63 + max(centre, my + rng)) // 64
        assert lag == 2 and abs(shift) + rng + 63 >= 128
        # fill both picture entries with other content first: the entries the test's pictures get are recycled ones
        assert L.x265hip_me_stream_picture_rows(prov.handle, 1, stale.ctypes.data, 0, ctus_h) == 0
        assert L.x265hip_me_stream_picture_rows(prov.handle, 2, stale.ctypes.data, 0, ctus_h) == 0
        time.sleep(0.2)
        assert L.x265hip_me_stream_picture_rows(prov.handle, 11, planes[1].ctypes.data, 0, ctus_h) == 0
        gen = L.x265hip_me_stream_pair_open(prov.handle, 0, 11, 12)
        assert gen > 0
        ready = np.ctypeslib.as_array((ctypes.c_int32 * ctus_h).from_address(L.x265hip_me_stream_ready(prov.handle, 0)))
        given = set()
        for r in (range(ctus_h) if shift > 0 else reversed(range(ctus_h))):        # the rows the windows reach arrive LAST: below (shift > 0) / above
            assert L.x265hip_me_stream_picture_rows(prov.handle, 12, planes[0].ctypes.data, r, 1) == 0
            given.add(r)
            time.sleep(0.15)                                 # long enough for the worker to search whatever it believes is searchable
            for q in range(ctus_h):
                if ready[q] == gen:
                    assert all(k in given for k in range(max(0, q - lag), min(ctus_h, q + lag + 1))), (q, sorted(given))
        t0 = time.time()
        while not all(ready[q] == gen for q in range(ctus_h)) and time.time() - t0 < 20:
            time.sleep(0.01)
        assert all(ready[q] == gen for q in range(ctus_h)), prov.report()
        org = geo["margin_y"] * geo["stride"] + geo["margin_x"]
        zc, zero = np.zeros(2 * centre + 1, np.uint16), np.zeros(2 * rng + 1, np.uint16)
        _, best = O.me_fullsearch(depth, planes[1], geo["stride"], org, planes[0], geo["stride"], org, geo["width"], geo["height"], centre, 0, nctu, zc, zc,
                                  want_surf=False, want_best=True)
        idx = (best.reshape(-1, 85)[:, 84] & 0xffffffff).astype(np.int64)
        ncb = 2 * centre + 1
        mx = min(centre, geo["margin_x"] - rng - 12)
        cen = np.stack([np.clip(idx % ncb - centre, -mx, mx), np.clip(idx // ncb - centre, -my, my)], axis=1).astype(np.int16)
        got_c = np.ctypeslib.as_array((ctypes.c_int16 * (2 * nctu)).from_address(L.x265hip_me_stream_centres(prov.handle, 0))).reshape(nctu, 2)
        assert np.array_equal(got_c, cen), (got_c, cen)
        inner = cen[ctus_w:(ctus_h - 1) * ctus_w] if shift > 0 else cen[2 * ctus_w:]
        assert np.all(inner[:, 1] == shift), cen             # the windows really sit two CTU rows' reach away
        parts = []
        for c in range(nctu):
            o = org + (c // ctus_w) * 64 * geo["stride"] + (c % ctus_w) * 64
            sc, _ = O.me_fullsearch(depth, planes[1], geo["stride"], o, planes[0], geo["stride"], o + int(cen[c, 1]) * geo["stride"] + int(cen[c, 0]),
                                    64, 64, rng, 0, 1, zero, zero, want_surf=True, want_best=False)
            parts.append(sc)
        cb = SD.planes_ctu_bytes(rng, 1)
        exp = SD.records_to_planes(np.concatenate(parts).reshape(-1, 85, 4), nctu, rng, 1)
        got = np.ctypeslib.as_array((ctypes.c_uint8 * (nctu * cb)).from_address(L.x265hip_me_stream_surface(prov.handle, 0))).reshape(nctu, cb)
        nc, pitch = 2 * rng + 1, 4 * ((2 * rng + 4) // 4)
        def valid(b):
            lo = b[:, :nc * pitch * 2 * 16].copy().view(np.uint16).reshape(nctu, 16, nc, pitch)[..., :nc]
            hi = b[:, nc * pitch * 2 * 16:].copy().view(np.uint32).reshape(nctu, 5, nc, pitch)[..., :nc]
            return lo, hi
        (gl, gh), (el, eh) = valid(got), valid(exp)
        assert np.array_equal(gl, el) and np.array_equal(gh, eh)
        assert prov.report()["failed"] == 0
    finally:
        prov.close()


@pytest.mark.parametrize("depth,preset,ft,min_level,centre,extra", [(8, "slow", 3, 1, 57, [("me", "star")]), (8, "medium", 3, 0, 0, []), (10, "slower", 2, 1, 40, []),
                                                                    (8, "slow", 3, 1, 40, [("me", "star"), ("bframes", "0")]), (8, "slow", 3, 2, 57, [("me", "star")])])
def test_row_granular_seams_on_the_gpu_with_planes_and_centred_windows(depth, preset, ft, min_level, centre, extra):
    """The real encoder on the planes layout (and centred windows) of x265hip_me_stream, weighted references included (the last case
    runs on a fade with --weightp on): byte-identical, every lookup verified in flight."""
    import test_seam_cpu as T
    from tools import seam_driver as SD
    fade = any(k == "bframes" for k, _ in extra)
    opts = [("pools", "4"), ("frame-threads", str(ft)), ("crf", "24")] + ([] if fade else [("no-weightp", None), ("no-weightb", None)]) + extra
    kw = dict(rng=12, wait=True, min_level=min_level, subpel="gpu", slots=32, subpel_slots=16, layout=SD.LAYOUT_PLANES, centre_range=centre)
    if fade:
        base, got, rep = T.run_fade_pair(depth, 256, 192, 10, preset, opts, "gpu", **kw)
    else:
        base, got, rep = T.run_pair(depth, 256, 192, 8, preset, opts, "gpu", verify=True, streamed=True, **kw)
    assert got[0] == base[0], f"seam changed the bitstream: {rep}"
    sub = rep["subpel_seam"]
    assert rep["verify"] == 1 and rep["verify_mismatches"] == 0 and sub["verify_mismatches"] == 0 and rep["failed"] == 0 and sub["failed"] == 0
    assert rep["lookups_served"] > (60 if min_level > 1 else 300 if min_level else 1500) and sub["subpel_compares_served"] > 1000, rep
    if fade:
        assert rep["weighted_references"]["lookups_served_on_weighted_references"] > 100, rep


def test_4k_preset_slow_star_frame_threads_5_every_served_value_verified_in_flight():
    """The size and options the metric is quoted on (BASELINE configs[2]: 3840x2160, --preset slow --me star, the reference's own
    --frame-threads 5), all three seams as bench.py's encoder leg configures them - and every served SAD, every served sub-sample
    comparison and every frame cost estimate re-evaluated in flight by the host's own function (the reference's contract
    check(ref.slot, opt.slot), test/testbench.cpp:181-233, at full size; round-3 verdict, next 3).  The timed leg runs unverified: an
    md5 alone would also pass with a provider that silently served nothing."""
    import test_seam_cpu as T
    from tools import encoder_bench as EB, seam_driver as SD
    try:
        plain = EB.ref_lib(8)
        SD.seam_lib(8)
    except (SystemExit, FileNotFoundError):
        missing_reference_build()
    w, h, n = 3840, 2160, 8
    clip = F.synth_clip(w, h, n, depth=8, seed=265)
    yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
    opts = [("pools", str(EB.effective_cpus())), ("frame-threads", "5"), ("crf", "28"), ("me", "star"), ("lookahead-slices", "1")]
    base = EB.encode(plain, yuv, w, h, n, "slow", opts)
    lib, filler, report, close, prov = SD.install(8, w, h, provider="gpu", rng=12, slots=24, min_pu=16, verify=True, lookahead="gpu+verify", subpel="gpu", subpel_slots=12,
                                                  streamed=True, min_level=1, pictures=24, layout=SD.LAYOUT_PLANES, centre_range=57, lookahead_min_blocks=None, min_ctus=None,
                                                  cost="gpu", cost_cfg=SD.cost_config("slow", opts, set_subme=4))          # round 6: the cost tables in front of the phase planes
    try:
        got = EB.encode(lib, yuv, w, h, n, "slow", opts, filler)
        rep = report()
    finally:
        close()
    assert got[0] == base[0], f"seams changed the bitstream: {rep}"
    sub, la, co = rep["subpel_seam"], rep["lookahead_seam"], rep["cost_seam"]
    assert rep["verify"] == 1 and rep["verify_mismatches"] == 0 and sub["verify_mismatches"] == 0 and la["verify_mismatches"] == 0 and co["verify_mismatches"] == 0, rep
    assert rep["failed"] == 0 and sub["failed"] == 0 and la["failed"] == 0 and co["failed"] == 0
    assert co["comparisons_served_from_records"] > 1_000_000 and co["stale_pairs"] == 0, co
    print("cfg3 cost tables:", {k: co.get(k) for k in ("comparisons_served_from_records", "served_share_of_satd_comparisons_with_context", "passed_on_records_not_arrived", "pairs_opened")})
    assert rep["lookups_served"] > 1_000_000 and sub["subpel_compares_served"] > 200_000 and la["frame_cost_estimates_served"] >= 20, rep
    assert la["left_to_the_reference_by_the_size_gate"] == 0 and not rep["search_seams_left_off_by_the_size_gate"]          # 4K is above the binding's size gates
    assert rep["lookup_hit_rate"] > 0.85, rep


def test_4k_fade_every_seam_on_weighted_references_and_both_host_services_verified_in_flight():
    """The same size on a FADE with x265's default --weightp: weightAnalyse (x265hip_weight_analyse_host) picks the weights, the weighted references go
    through the row-granular search services, calcAdaptiveQuantFrame (x265hip_aq_frame_host) feeds the statistics - all above the binding's size gates,
    every served SAD / comparison / frame cost re-evaluated by the host's function and both host services re-run by the reference's own functions."""
    import test_seam_cpu as T
    from tools import encoder_bench as EB, seam_driver as SD
    try:
        plain = EB.ref_lib(8)
        SD.seam_lib(8)
    except (SystemExit, FileNotFoundError):
        missing_reference_build()
    w, h, n = 3840, 2160, 8
    clip = F.synth_clip(w, h, n, depth=8, seed=265, fade=(1.0, 0.4))
    yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
    # --bframes 1: five P slices in eight frames, three of which come back weighted (with the preset's 4 B frames the two P slices of so short a clip keep weight 1)
    opts = [("pools", str(EB.effective_cpus())), ("frame-threads", "5"), ("crf", "28"), ("me", "star"), ("lookahead-slices", "1"), ("bframes", "1")]
    base = EB.encode(plain, yuv, w, h, n, "slow", opts)
    lib, filler, report, close, prov = SD.install(8, w, h, provider="gpu", rng=12, slots=24, min_pu=16, verify=True, lookahead="gpu+verify", subpel="gpu", subpel_slots=12,
                                                  streamed=True, min_level=1, pictures=24, layout=SD.LAYOUT_PLANES, centre_range=57, lookahead_min_blocks=None, min_ctus=None,
                                                  aq="gpu", aq_min_blocks=None, weight_analyse="gpu", weight_min_blocks=None)
    try:
        got = EB.encode(lib, yuv, w, h, n, "slow", opts, filler)
        rep = report()
    finally:
        close()
    assert got[0] == base[0], f"seams changed the bitstream: {rep}"
    sub, la, aq, wa = rep["subpel_seam"], rep["lookahead_seam"], rep["aq_seam"], rep["weight_analyse_seam"]
    assert rep["verify_mismatches"] == 0 and sub["verify_mismatches"] == 0 and la["verify_mismatches"] == 0 and aq["verify_mismatches"] == 0 and wa["verify_mismatches"] == 0, rep
    assert rep["failed"] == 0 and sub["failed"] == 0 and la["failed"] == 0 and aq["failed"] == 0 and wa["failed"] == 0
    assert aq["pictures_served"] == n and aq["left_to_the_reference_by_the_size_gate"] == 0, aq
    assert wa["slices_served"] >= 2 and wa["served_slices_with_a_weight"] >= 1 and wa["left_to_the_reference_by_the_size_gate"] == 0, wa
    assert rep["weighted_references"]["lookups_served_on_weighted_references"] > 20_000, rep["weighted_references"]
    assert rep["weighted_references"]["subpel_compares_served_from_weighted_views"] > 2_000, rep["weighted_references"]


def _verified_encode(depth, w, h, n, preset, extra, clip=None, slots=40, subpel_slots=12, pictures=24, min_level=1, min_ctus=None, gates=None, cost_slots=40, cost_pictures=40,
                     cost_views=12):
    """The real encoder with its own table, then with every seam the bench legs configure and `verify` on: every served SAD, sub-sample comparison and
    frame cost is re-evaluated by the host's own function in flight, the AQ / weightAnalyse services re-run by the reference's functions.  Returns
    (md5 equal, report)."""
    from tools import encoder_bench as EB, seam_driver as SD
    try:
        plain = EB.ref_lib(depth)
        SD.seam_lib(depth)
    except (SystemExit, FileNotFoundError):
        missing_reference_build()
    if clip is None:
        clip = F.synth_clip(w, h, n, depth=depth, seed=265)
    yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
    opts = [("pools", str(EB.effective_cpus())), ("frame-threads", "5"), ("crf", "28"), ("lookahead-slices", "1")] + extra
    base = EB.encode(plain, yuv, w, h, n, preset, opts)
    lib, filler, report, close, prov = SD.install(depth, w, h, provider="gpu", rng=12, slots=slots, min_pu=16, verify=True, lookahead="gpu+verify", subpel="gpu",
                                                  subpel_slots=subpel_slots, streamed=True, min_level=min_level, pictures=pictures, layout=SD.LAYOUT_PLANES, centre_range=57,
                                                  lookahead_min_blocks=gates, min_ctus=min_ctus, aq="gpu", aq_min_blocks=gates, weight_analyse="gpu", weight_min_blocks=gates,
                                                  cost="gpu", cost_cfg=SD.cost_config(preset, opts, set_subme=4, slots=cost_slots, pictures=cost_pictures, views=cost_views))
    try:
        got = EB.encode(lib, yuv, w, h, n, preset, opts, filler)
        rep = report()
    finally:
        close()
    return got[0] == base[0], rep


def _assert_all_verified(rep):
    sub, la, aq, wa, co = rep["subpel_seam"], rep["lookahead_seam"], rep["aq_seam"], rep["weight_analyse_seam"], rep["cost_seam"]
    assert rep["verify"] == 1
    assert rep["verify_mismatches"] == 0 and sub["verify_mismatches"] == 0 and la["verify_mismatches"] == 0 and aq["verify_mismatches"] == 0 and wa["verify_mismatches"] == 0, rep
    assert co["verify_mismatches"] == 0 and co["failed"] == 0, co          # round 6: the cost tables serve in front of the phase planes
    assert rep["failed"] == 0 and sub["failed"] == 0 and la["failed"] == 0 and aq["failed"] == 0 and wa["failed"] == 0, rep


def test_4k_10bit_preset_slower_frame_threads_5_every_served_value_verified_in_flight():
    """BASELINE configs[3] - 3840x2160 Main10, --preset slower (AMP, rd 6, 5 references, 8 B frames), --frame-threads 5, one GPU's share of the
    frame-parallel job - with every seam of the bench's cfg4 leg and `verify` on (round-4 verdict, next 2a: that leg is only md5-checked)."""
    same, rep = _verified_encode(10, 3840, 2160, 8, "slower", [])
    assert same, f"seams changed the bitstream: {rep}"
    _assert_all_verified(rep)
    sub, la, aq = rep["subpel_seam"], rep["lookahead_seam"], rep["aq_seam"]
    assert rep["lookups_served"] > 1_000_000 and sub["subpel_compares_served"] > 200_000 and la["frame_cost_estimates_served"] >= 10, rep
    assert rep["cost_seam"]["comparisons_served_from_records"] > 2_000_000, rep["cost_seam"]
    assert aq["pictures_served"] == 8 and la["left_to_the_reference_by_the_size_gate"] == 0 and not rep["search_seams_left_off_by_the_size_gate"]
    assert rep["lookup_hit_rate"] > 0.85, rep
    print("cfg4 verified:", {k: rep.get(k) for k in ("lookups_served", "lookup_hit_rate", "bytes_downloaded")}, rep["weighted_references"])


def test_8k_10bit_preset_veryslow_rd6_every_served_value_verified_in_flight():
    """BASELINE configs[4] - 7680x4320 Main10, --preset veryslow --ctu 64 --rd 6 - three frames with every seam and `verify` on (round-4 verdict, next 2b);
    fewer resident pairs / views than at 4K: a reference picture's phase planes are 3.4 GB of pinned memory at this size."""
    same, rep = _verified_encode(10, 7680, 4320, 3, "veryslow", [("ctu", "64"), ("rd", "6")], slots=12, subpel_slots=4, pictures=8, cost_slots=12, cost_pictures=12, cost_views=6)
    assert same, f"seams changed the bitstream: {rep}"
    _assert_all_verified(rep)
    sub, la, aq = rep["subpel_seam"], rep["lookahead_seam"], rep["aq_seam"]
    assert rep["lookups_served"] > 1_000_000 and sub["subpel_compares_served"] > 50_000 and la["frame_cost_estimates_served"] >= 2, rep
    assert rep["cost_seam"]["comparisons_served_from_records"] > 200_000, rep["cost_seam"]
    assert aq["pictures_served"] == 3 and not rep["search_seams_left_off_by_the_size_gate"]
    print("cfg5 verified:", {k: rep.get(k) for k in ("lookups_served", "lookup_hit_rate", "bytes_downloaded")})


def test_saturated_16_bit_plane_entries_go_back_to_the_host_10bit():
    """Above 8 bits a 16x16 SAD can exceed what the planes layout's uint16 rasters hold (16 * 16 * 1023 > 65535): the entry saturates to 65535 and the
    lookup must hand the candidate back to the host's primitive.  A 10-bit clip whose middle picture is inverted makes such entries certain; every
    served value is verified in flight and the bitstream stays the reference's."""
    depth, w, h, n = 10, 256, 192, 5
    clip = F.synth_clip(w, h, n, depth=depth, seed=77)
    y, u, v = clip[2]
    clip[2] = ((1023 - y).astype(y.dtype), u, v)
    same, rep = _verified_encode(depth, w, h, n, "slow", [("me", "star"), ("bframes", "0")], clip=clip, slots=16, subpel_slots=6, pictures=8, min_ctus=0, gates=0)
    assert same, f"seams changed the bitstream: {rep}"
    _assert_all_verified(rep)
    assert rep["weighted_references"]["lookups_on_saturated_16_bit_entries"] > 0, rep["weighted_references"]
    assert rep["lookups_served"] > 1000, rep


def test_stream_services_can_be_pinned_to_a_device():
    """device_plus_1 of x265hip_me_stream_params / x265hip_phase_stream_params: an instance per GPU for a host that spreads its frame
    encoders over a node (encoder/encoder.cpp:304-321).  Device 0 named explicitly works like the default; a device the box does not
    have is refused with X265HIP_ENODEV and a message."""
    from tools import seam_driver as SD
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    geo = SD.geometry(128, 128)
    prov = SD.StreamGpuProvider(8, geo, 8, slots=1, min_level=1, pictures=2, layout=SD.LAYOUT_PLANES, device=0)
    ph = SD.StreamGpuPhaseProvider(8, geo, slots=1, device=0)
    prov.close(); ph.close()
    n = A.lib().x265hip_device_count()
    for cls, args in ((SD.StreamGpuProvider, (8, geo, 8, 1)), (SD.StreamGpuPhaseProvider, (8, geo, 1))):
        with pytest.raises(A.X265HipError) as e:
            cls(*args, device=n)
        assert "device" in str(e.value)


@pytest.mark.parametrize("depth,preset,ft,extra", [(8, "slow", 3, [("me", "star")]), (10, "medium", 4, [("bframes", "2")])])
def test_two_service_instances_serve_one_encode_frame_encoders_spread_over_devices(depth, preset, ft, extra):
    """The multi-GPU mapping of the consumer services driven from ONE real encode (round-4 verdict, next 6): an x265hip_me_stream and an
    x265hip_phase_stream instance per device (device_plus_1), pairs / views dealt over the instances, every reconstructed CTU row handed to every
    instance (the hand-over SURVEY 8(e) puts on RCCL between GPUs).  One GPU box: device 0 named twice - the same code path, two instances with
    their own workers, streams and pinned buffers.  Every served value verified in flight, bitstream identical, both instances did work."""
    import test_seam_cpu as T
    from tools import encoder_bench as EB, seam_driver as SD
    try:
        plain = EB.ref_lib(depth)
        SD.seam_lib(depth)
    except (SystemExit, FileNotFoundError):
        missing_reference_build()
    w, h, n = 320, 256, 9
    clip = F.synth_clip(w, h, n, depth=depth, seed=31)
    yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
    opts = [("pools", str(EB.effective_cpus())), ("frame-threads", str(ft)), ("crf", "28")] + extra
    base = EB.encode(plain, yuv, w, h, n, preset, opts)
    lib, filler, report, close, prov = SD.install(depth, w, h, provider="gpu", rng=12, slots=16, min_pu=16, verify=True, subpel="gpu", subpel_slots=8, streamed=True,
                                                  min_level=1, pictures=12, layout=SD.LAYOUT_PLANES, centre_range=40, min_ctus=0, devices=[0, 0])
    try:
        got = EB.encode(lib, yuv, w, h, n, preset, opts, filler)
        rep = report()
    finally:
        close()
    assert got[0] == base[0], f"seams changed the bitstream: {rep}"
    sub = rep["subpel_seam"]
    assert rep["verify"] == 1 and rep["verify_mismatches"] == 0 and sub["verify_mismatches"] == 0 and rep["failed"] == 0 and sub["failed"] == 0, rep
    assert rep["lookups_served"] > 1000 and sub["subpel_compares_served"] > 1000, rep
    assert len(rep["instances"]) == 2 and all(i["pairs_completed"] > 0 and i["failed"] == 0 for i in rep["instances"]), rep["instances"]
    assert all(i["rows_uploaded"] > 0 for i in rep["instances"])                       # every instance received the reconstructed rows
    assert len(sub["instances"]) == 2 and all(i["completed"] > 0 for i in sub["instances"]), sub["instances"]


# ---- round 4: the pre-lookahead's adaptive-quantisation pass from x265hip_aq_frame_host ------------------------------------------------
@pytest.mark.parametrize("depth,w,h,extra", [(8, 256, 192, []), (8, 256, 192, [("aq-mode", "3"), ("aq-strength", "1.4")]), (8, 256, 192, [("qg-size", "8")]),
                                             (10, 192, 128, [("aq-mode", "1")]), (8, 256, 192, [("no-weightp", None), ("no-weightb", None)])])
def test_aq_seam_on_the_gpu_fills_the_arrays_the_reference_loop_would(depth, w, h, extra):
    """The real encoder with LookaheadTLD::calcAdaptiveQuantFrame served by x265hip_aq_frame_host for every source picture; after each one the
    reference's own function recomputes qpAqOffset / qpCuTreeOffset / invQscaleFactor (+ 8x8) / wp_sum / wp_ssd and the binding compares bit for
    bit.  Byte-identical bitstream."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "2"), ("crf", "24")] + extra
    base, got, rep = T.run_pair(depth, w, h, 6, "medium", opts, "gpu", rng=8, streamed=True, min_level=1, slots=16, aq="gpu")
    a = rep["aq_seam"]
    assert got[0] == base[0], f"seam changed the bitstream: {a}"
    assert a["pictures_served"] == 6 and a["verify_mismatches"] == 0 and a["failed"] == 0 and a["passed_to_reference_loop"] == 0, a


# ---- round 4: the frame encoder's weightAnalyse from x265hip_weight_analyse_host ------------------------------------------------------------
@pytest.mark.parametrize("depth,preset,fade,extra", [(8, "slow", (1.0, 0.35), [("bframes", "0")]), (8, "medium", (0.4, 1.0), [("weightb", None), ("bframes", "3")]),
                                                     (10, "medium", (1.0, 0.35), [("weightb", None)])])
def test_weight_analyse_seam_on_the_gpu_chooses_the_reference_weights(depth, preset, fade, extra):
    """The real encoder on a fading clip with weightAnalyse served by x265hip_weight_analyse_host (and calcAdaptiveQuantFrame, whose wp_ssd / wp_sum it
    reads, by x265hip_aq_frame_host); after every served slice the reference's own weightAnalyse runs and the weight tables are compared.  The
    weighted references it chooses then go through the row-granular search services as in the tests above.  Byte-identical."""
    import test_seam_cpu as T
    opts = [("pools", "4"), ("frame-threads", "2"), ("crf", "24")] + extra
    base, got, rep = T.run_fade_pair(depth, 256, 192, 10, preset, opts, "gpu", rng=8, fade=fade, min_level=1, slots=24, subpel="gpu", subpel_slots=12, lookahead="gpu",
                                     weight_analyse="gpu", aq="gpu")
    w = rep["weight_analyse_seam"]
    assert got[0] == base[0], f"seam changed the bitstream: {w}"
    assert w["verify_mismatches"] == 0 and w["failed"] == 0 and w["passed_to_reference_loop"] == 0 and w["slices_served"] >= 2, w
    assert w["served_slices_with_a_weight"] > 0, w
    assert rep["aq_seam"]["verify_mismatches"] == 0 and rep["aq_seam"]["pictures_served"] == 10
    assert rep["verify_mismatches"] == 0 and rep["weighted_references"]["lookups_served_on_weighted_references"] > 0, rep
