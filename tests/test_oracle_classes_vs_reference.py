"""Pin the oracle's search + sub-pel chain against the REAL reference class: x265::MotionEstimate::motionEstimate()
(encoder/motion.cpp:739-1561) run through oracle/ref_motion.cpp on the same padded planes, --me full, every PU of every
CTU, several sub-pel levels.  The oracle side is exactly what the GPU parity tests compare the HIP kernels with
(oracle_api.me_fullsearch -> subpel_refine), so GPU == oracle == reference for integer search + sub-pel refinement.

The BitCost table of the reference at QP 24 (8-bit) / QP 12... is lambda = 4.0 (constants.cpp lambda tables), the value
frames.mv_cost_table() uses, and the predictor is (0,0), so both sides price a motion vector identically."""
import ctypes
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
F = importlib.import_module("x265-yuuki-asuna_amd.frames")

X265_FULL_SEARCH = 5          # x265.h:492-497


class Job(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("px", "py", "w", "h", "qmvpx", "qmvpy", "out_qmvx", "out_qmvy", "out_cost")]


def _ref(depth):
    path = os.path.join(ROOT, "oracle", "_ref", f"libx265ref{depth}.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = ctypes.CDLL(path)
    if not hasattr(lib, "x265ref_motion_estimate"):
        pytest.skip("oracle/_ref predates ref_motion.cpp")
    lib.x265ref_motion_estimate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int]
    return lib


def _zxy(z):
    return (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)


@pytest.mark.parametrize("depth,width,height,rng,seed", [(8, 128, 128, 8, 21), (8, 192, 64, 12, 22), (10, 128, 64, 6, 23), (12, 128, 64, 6, 24)])
def test_fullsearch_subpel_chain_equals_reference_motion_estimate(depth, width, height, rng, seed):
    import oracle_api as O
    lib = _ref(depth)
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    cur, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    cost = F.mv_cost_table(rng)
    cq, qoff = F.qpel_cost_table(rng)
    nctu = (w64 // 64) * (h64 // 64)
    es = cur.itemsize
    qp = {8: 24, 10: 12, 12: 0}[depth]     # lambda 4.0 in the reference's table for this bit depth (constants.cpp:31-150)
    _, best = O.me_fullsearch(depth, cur, stride, org, ref, stride, org, w64, h64, rng, 0, nctu, cost, cost, want_surf=False)
    for subme in (0, 1, 2, 3, 5, 7):
        mv = O.subpel_refine(depth, cur, stride, org, ref, stride, org, w64, h64, rng, 0, nctu, best, cq, qoff, subme).reshape(-1, 2)
        jobs = (Job * (nctu * 85))()
        for ctu in range(nctu):
            cx, cy = (ctu % (w64 // 64)) * 64, (ctu // (w64 // 64)) * 64
            for base, n, size in ((0, 64, 8), (64, 16, 16), (80, 4, 32), (84, 1, 64)):
                for z in range(n):
                    bx, by = _zxy(z)
                    j = jobs[ctu * 85 + base + z]
                    j.px, j.py, j.w, j.h, j.qmvpx, j.qmvpy = cx + bx * size, cy + by * size, size, size, 0, 0
        lib.x265ref_motion_estimate(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, X265_FULL_SEARCH, subme, rng, qp,
                                    -rng, -rng, rng, rng, jobs, nctu * 85)
        got = np.array([(j.out_cost, j.out_qmvx, j.out_qmvy) for j in jobs], dtype=np.int64)
        q = mv[:, 1].astype(np.int64)
        exp = np.stack([mv[:, 0].astype(np.int64), ((q & 0xffff) ^ 0x8000) - 0x8000, q >> 16], axis=1)
        bad = np.nonzero((got != exp).any(axis=1))[0]
        assert bad.size == 0, f"subme {subme}: {bad.size} of {len(got)} PUs differ, first {bad[:3]}: ref {got[bad[:3]]} oracle {exp[bad[:3]]}"


ALL_PU_DIMS = [(8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 8), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (16, 12),
               (12, 16), (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32), (64, 48), (48, 64), (64, 16), (16, 64)]
METHODS = {"dia": 0, "hex": 1, "umh": 2, "star": 3, "full": 5}          # x265.h:492-497


def random_me_jobs(rng, n, width, height):
    jobs = (Job * n)()
    for j in jobs:
        w, h = ALL_PU_DIMS[int(rng.integers(0, len(ALL_PU_DIMS)))]
        j.px, j.py = int(rng.integers(0, (width - w) // 4 + 1)) * 4, int(rng.integers(0, (height - h) // 4 + 1)) * 4
        j.w, j.h = w, h
        if rng.integers(0, 4):                                     # mostly a non-zero, usually fractional predictor
            j.qmvpx, j.qmvpy = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
    return jobs


def copy_jobs(jobs):
    out = (Job * len(jobs))()
    ctypes.memmove(out, jobs, ctypes.sizeof(jobs))
    return out


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("method", ["dia", "hex", "umh", "star", "full"])
def test_search_driver_restatement_equals_reference_motion_estimate(depth, method, seed=31):
    """oracle/x265_oracle_search.c (predictor start, DIA / HEX / STAR / FULL patterns, predictor-vs-search choice, sub-pel
    refinement) against the real MotionEstimate::motionEstimate: random PU sizes (all 24 inter partitions), positions,
    quarter-pel predictors, search bounds, merange and every sub-pel level."""
    lib = _ref(depth)
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle.so"))
    f = getattr(orc, f"x265oracle_motion_estimate_d{depth}")
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + \
                 [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    width, height = 256, 192
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    cur, stride, org, _, _ = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    es = cur.itemsize
    rng = np.random.default_rng([7, depth, METHODS[method], seed])
    qp = {8: 24, 10: 12, 12: 0}[depth]
    total = 0
    for subme in range(8):
        for merange in ((4, 16, 57) if method != "full" else (6,)):
            bound = merange if method == "full" else 57
            cq, qoff = F.qpel_cost_table(bound, qmax=8 * 64 + 300)
            mn, mx = (-bound, -bound), (bound, bound)
            if rng.integers(0, 3) == 0:                            # tight, asymmetric bounds exercise the border branches
                mn = (-int(rng.integers(3, 20)), -int(rng.integers(3, 20)))
                mx = (int(rng.integers(3, 20)), int(rng.integers(3, 20)))
            ja = random_me_jobs(rng, 40, width, height)
            jb = copy_jobs(ja)
            lib.x265ref_motion_estimate(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, METHODS[method], subme, merange, qp,
                                        mn[0], mn[1], mx[0], mx[1], ja, len(ja))
            assert f(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, METHODS[method], subme, merange, cq.ctypes.data, qoff,
                     mn[0], mn[1], mx[0], mx[1], jb, len(jb), 1) == 0
            for a, b in zip(ja, jb):
                assert (a.out_cost, a.out_qmvx, a.out_qmvy) == (b.out_cost, b.out_qmvx, b.out_qmvy), \
                    f"{method} subme {subme} merange {merange} PU {(a.px, a.py, a.w, a.h)} mvp {(a.qmvpx, a.qmvpy)} bounds {mn}..{mx}"
                total += 1
    assert total >= 300


SEA_UNSUPPORTED = {(8, 4), (4, 8), (32, 8), (8, 32)}      # their DC terms read source samples outside the PU (motion.cpp:1259-1260,1305)


@pytest.mark.parametrize("depth", [8, 10])
def test_sea_search_restatement_equals_reference_motion_estimate(depth, seed=37):
    """X265_SEA (motion.cpp:1241-1395): the restatement takes its block sums straight from the reference samples; the real class
    gets the twelve integral planes built by the library's own integral_init primitives the way FrameFilter does.  Every
    supported PU size (the ADS variant, plane and offsets differ per size - including the sizes whose plane does not match
    their DC block), random predictors, bounds, merange and sub-pel levels."""
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_motion_estimate_sea"):
        pytest.skip("oracle/_ref predates the SEA entry point")
    lib.x265ref_motion_estimate_sea.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_int] * 11 + [ctypes.c_void_p, ctypes.c_int]
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle.so"))
    f = getattr(orc, f"x265oracle_motion_estimate_d{depth}")
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + \
                 [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    width, height = 256, 192
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    cur, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    es = cur.itemsize
    rng = np.random.default_rng([11, depth, seed])
    qp = {8: 24, 10: 12, 12: 0}[depth]
    cq, qoff = F.qpel_cost_table(57, qmax=8 * 64 + 300)
    seen, moved = set(), 0
    for subme in (0, 1, 2, 3, 5, 7):
        for merange in (5, 16, 40):
            mn, mx = (-44, -44), (44, 44)
            if rng.integers(0, 3) == 0:
                mn = (-int(rng.integers(3, 20)), -int(rng.integers(3, 20)))
                mx = (int(rng.integers(3, 20)), int(rng.integers(3, 20)))
            ja = random_me_jobs(rng, 60, width, height)
            keep = [j for j in ja if (j.w, j.h) not in SEA_UNSUPPORTED]
            ja = (Job * len(keep))(*keep)
            jb = copy_jobs(ja)
            assert lib.x265ref_motion_estimate_sea(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, w64, h64, F.MARGIN_X, F.MARGIN_Y,
                                                   subme, merange, qp, mn[0], mn[1], mx[0], mx[1], ja, len(ja)) == len(ja)
            assert f(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, 4, subme, merange, cq.ctypes.data, qoff,
                     mn[0], mn[1], mx[0], mx[1], jb, len(jb), 1) == 0
            for a, b in zip(ja, jb):
                assert (a.out_cost, a.out_qmvx, a.out_qmvy) == (b.out_cost, b.out_qmvx, b.out_qmvy), \
                    f"subme {subme} merange {merange} PU {(a.px, a.py, a.w, a.h)} mvp {(a.qmvpx, a.qmvpy)} bounds {mn}..{mx}"
                seen.add((a.w, a.h))
                moved += (a.out_qmvx, a.out_qmvy) != (0, 0)
    assert len(seen) == len(ALL_PU_DIMS) - len(SEA_UNSUPPORTED) and moved > 300
    # the unsupported sizes are refused, not approximated
    bad = (Job * 1)()
    bad[0].w, bad[0].h = 32, 8
    assert f(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, 4, 2, 8, cq.ctypes.data, qoff, -8, -8, 8, 8, bad, 1, 1) != 0


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 192, 144), (10, 128, 128), (12, 128, 128)])
def test_lookahead_restatement_equals_reference_classes(depth, width, height):
    """oracle/x265_oracle_pipeline3.c against the real Lowres::init + LookaheadTLD::lowresIntraEstimate
    (oracle/ref_lookahead.cpp): the four half-resolution planes including their extended borders, intraCost, intraMode
    and lowresCosts of every 8x8 block."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_lowres_intra"):
        pytest.skip("oracle/_ref predates ref_lookahead.cpp")
    y = F.synth_clip(width, height, 1, depth=depth, seed=77)[0][0]
    src, stride, org, w64, h64 = F.pad_plane(y)
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    lw, lh = wcu * 8, hcu * 8
    geo = (ctypes.c_int * 5)()
    rstride = (width // 2 + 2 * F.MARGIN_X + 31) & ~31                  # Lowres::create, lowres.cpp:59-61
    rows = lh + 2 * F.MARGIN_Y
    rplanes = [np.zeros(rstride * rows, dtype=y.dtype) for _ in range(4)]
    rcost, rmode, rlc = np.zeros(wcu * hcu, np.int32), np.zeros(wcu * hcu, np.uint8), np.zeros(wcu * hcu, np.uint16)
    lib.x265ref_lowres_intra.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_void_p] * 7
    rc = lib.x265ref_lowres_intra(src.ctypes.data, width, height, geo, *[p.ctypes.data for p in rplanes],
                                  rcost.ctypes.data, rmode.ctypes.data, rlc.ctypes.data)
    assert rc == 0
    assert tuple(geo) == (lw, lh, rstride, wcu, hcu)
    penalty = {8: 5, 10: 80, 12: 1280}[depth]  # 5 * (int)x265_lambda_tab[X265_LOOKAHEAD_QP]: 1.0 (8-bit, QP 12) / 16.0 (10-bit, QP 24) / 256.0 (12-bit, QP 36)
    lorg = rstride * F.MARGIN_Y + F.MARGIN_X
    planes = O.lowres_init(depth, src, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    for i in range(4):
        a = planes[i].reshape(rows, rstride)[:, :lw + 2 * F.MARGIN_X]
        b = rplanes[i].reshape(rows, rstride)[:, :lw + 2 * F.MARGIN_X]
        if lw + 2 * F.MARGIN_X > rstride:      # rounded-up width: the reference's right margin runs into the next row (lowres.cpp:59-66)
            a, b = a[:, :F.MARGIN_X + lw], b[:, :F.MARGIN_X + lw]
        assert np.array_equal(a, b), f"lowres plane {i} differs from Lowres::init"
    cost, mode, lc = O.lowres_intra(depth, planes[0], rstride, lorg, wcu, hcu, penalty)
    assert np.array_equal(cost, rcost) and np.array_equal(mode, rmode) and np.array_equal(lc, rlc)


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 208, 144), (10, 192, 128)])
def test_lowres_frame_cost_restatement_equals_reference_classes(depth, width, height, seed=91, check_coverage=True):
    """oracle/x265_oracle_search.c::x265oracle_lowres_cost against the real CostEstimateGroup::singleCost(0, 1, 1)
    (oracle/ref_lookahead.cpp): the lookahead's P-frame cost estimate - neighbour-mv predictors, HEX search on the four
    half-pel phase planes, intra competition - mv, mv cost and lowresCosts of every 8x8 block, rowSatds, costEst, intraMbs."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_lowres_cost"):
        pytest.skip("oracle/_ref predates x265ref_lowres_cost")
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    rng = np.random.default_rng([13, depth, width, seed])
    y0 = clip[0][0]
    # the current picture: the reference shifted by a global motion plus local differences, so that mvs, intra wins and
    # zero-residual blocks all occur
    y1 = np.roll(y0, (3, -5), axis=(0, 1)).copy()
    y1[: height // 3] = clip[1][0][: height // 3]
    y1[-32:, -64:] = y0[-32:, -64:]
    noise = rng.integers(-2, 3, size=y1.shape) << (depth - 8)
    y1[:, : width // 2] = np.clip(y1[:, : width // 2].astype(np.int32) + noise[:, : width // 2], 0, (1 << depth) - 1).astype(y1.dtype)
    cur, stride, org, w64, h64 = F.pad_plane(y1)
    ref = F.pad_plane(y0)[0]
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    lw, lh = wcu * 8, hcu * 8
    n = wcu * hcu
    rmv, rmc, rlc = np.zeros((n, 2), np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint16)
    rrows, rframe = np.zeros(hcu, np.int32), np.zeros(4, np.int64)
    lib.x265ref_lowres_cost.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    assert lib.x265ref_lowres_cost(cur.ctypes.data, ref.ctypes.data, width, height, rmv.ctypes.data, rmc.ctypes.data,
                                   rlc.ctypes.data, rrows.ctypes.data, rframe.ctypes.data) == 0
    rstride = (width // 2 + 2 * F.MARGIN_X + 31) & ~31
    rows = lh + 2 * F.MARGIN_Y
    lorg = rstride * F.MARGIN_Y + F.MARGIN_X
    cplanes = O.lowres_init(depth, cur, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    rplanes = O.lowres_init(depth, ref, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    lam = {8: 1.0, 10: 16.0, 12: 256.0}[depth]    # x265_lambda_tab[X265_LOOKAHEAD_QP]
    icost, _, _ = O.lowres_intra(depth, cplanes[0], rstride, lorg, wcu, hcu, 5 * int(lam))
    cq, qoff = F.qpel_cost_table(16, lam=lam, qmax=4 * (max(lw, lh) + 64))
    mvs, mvc, lc, rws, frame = O.lowres_cost(depth, cplanes[0], rplanes, rstride, lorg, wcu, hcu, cq, qoff, icost)
    assert np.array_equal(mvs, rmv), f"mvs differ at {np.flatnonzero((mvs != rmv).any(axis=1))[:8]}"
    assert np.array_equal(mvc, rmc) and np.array_equal(lc, rlc) and np.array_equal(rws, rrows)
    assert (frame[0], frame[1], frame[2]) == (rframe[1], rframe[2], rframe[3]) and rframe[0] == rframe[1]
    assert not check_coverage or ((mvs != 0).any() and ((lc >> 14) == 0).any() and ((lc >> 14) == 1).any())


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 208, 144), (10, 192, 128)])
def test_lowres_b_frame_cost_restatement_equals_reference_classes(depth, width, height, seed=92, check_coverage=True):
    """The B-picture flavour (two lists, skip shortcut, the two bi-directional candidates, the 100 / (130 + bias) score scaling)
    against the real CostEstimateGroup::singleCost(0, 2, 1)."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_lowres_cost_b"):
        pytest.skip("oracle/_ref predates x265ref_lowres_cost_b")
    clip = F.synth_clip(width, height, 3, depth=depth, seed=seed)
    rng = np.random.default_rng([14, depth, width, seed])
    y0, y2 = clip[0][0], clip[2][0]
    y1 = np.roll(y0, (2, -3), axis=(0, 1)).copy()
    y1[: height // 3] = np.roll(y2, (-1, 2), axis=(0, 1))[: height // 3]
    y1[-32:, -64:] = y0[-32:, -64:]                                        # zero-residual blocks (skip shortcut)
    y1[-64:-32, :64] = ((y0[-64:-32, :64].astype(np.int32) + y2[-64:-32, :64] + 1) >> 1).astype(y1.dtype)   # bi-directional wins
    noise = rng.integers(-1, 2, size=y1.shape) << (depth - 8)
    y1[:, width // 2:] = np.clip(y1[:, width // 2:].astype(np.int32) + noise[:, width // 2:], 0, (1 << depth) - 1).astype(y1.dtype)
    cur, stride, org, w64, h64 = F.pad_plane(y1)
    r0, r1 = F.pad_plane(y0)[0], F.pad_plane(y2)[0]
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    lw, lh = wcu * 8, hcu * 8
    n = wcu * hcu
    rmv = [np.zeros((n, 2), np.int32) for _ in range(2)]
    rmc = [np.zeros(n, np.int32) for _ in range(2)]
    rlc, rrows, rframe = np.zeros(n, np.uint16), np.zeros(hcu, np.int32), np.zeros(4, np.int64)
    lib.x265ref_lowres_cost_b.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 7
    assert lib.x265ref_lowres_cost_b(cur.ctypes.data, r0.ctypes.data, r1.ctypes.data, width, height, rmv[0].ctypes.data, rmc[0].ctypes.data,
                                     rmv[1].ctypes.data, rmc[1].ctypes.data, rlc.ctypes.data, rrows.ctypes.data, rframe.ctypes.data) == 0
    rstride = (width // 2 + 2 * F.MARGIN_X + 31) & ~31
    rows = lh + 2 * F.MARGIN_Y
    lorg = rstride * F.MARGIN_Y + F.MARGIN_X
    cplanes = O.lowres_init(depth, cur, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    p0 = O.lowres_init(depth, r0, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    p1 = O.lowres_init(depth, r1, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    lam = {8: 1.0, 10: 16.0, 12: 256.0}[depth]
    icost, _, _ = O.lowres_intra(depth, cplanes[0], rstride, lorg, wcu, hcu, 5 * int(lam))
    cq, qoff = F.qpel_cost_table(16, lam=lam, qmax=4 * (max(lw, lh) + 64))
    mvs, mvc, lc, rws, frame = O.lowres_cost(depth, cplanes[0], p0, rstride, lorg, wcu, hcu, cq, qoff, icost, ref1_planes=p1)
    for i in range(2):
        assert np.array_equal(mvs[i], rmv[i]), f"list {i} mvs differ at {np.flatnonzero((mvs[i] != rmv[i]).any(axis=1))[:8]}"
        assert np.array_equal(mvc[i], rmc[i])
    assert np.array_equal(lc, rlc) and np.array_equal(rws, rrows)
    assert frame[3] == rframe[0] == rframe[1] and frame[1] == rframe[2] and rframe[3] == 0
    used = lc >> 14
    assert not check_coverage or ((used == 1).any() and (used == 2).any() and (used == 3).any())


def fade_pair(depth, width, height, gain, lift, seed):
    """A reference picture and a current picture that is the same scene (static but for a little noise: weightCostLuma compares at
    zero motion) with a fade applied (gain, lift)."""
    clip = F.synth_clip(width, height, 1, depth=depth, seed=seed)
    y0 = clip[0][0]
    pmax = (1 << depth) - 1
    noise = np.random.default_rng(seed).integers(-1, 2, size=y0.shape) * (1 << (depth - 8))
    y1 = np.clip(np.rint(y0.astype(np.float64) * gain + lift * (1 << (depth - 8))) + noise, 0, pmax).astype(y0.dtype)
    return y0, y1


def lowres_stats(plane, rows, stride, lorg, lw, lh):
    """wp_sum / wp_ssd of a picture, taken over its lowres plane 0 (so that the means weightsAnalyse derives from them are the
    plane's): sum, and sum of squares minus sum^2 / n (lowres.h:220)."""
    a = plane.reshape(rows, stride)[lorg // stride: lorg // stride + lh, lorg % stride: lorg % stride + lw].astype(np.int64)
    sm = int(a.sum())
    return int((a * a).sum()) - sm * sm // a.size, sm


@pytest.mark.parametrize("depth,width,height,gain,lift", [(8, 256, 128, 0.75, 6), (8, 208, 144, 1.0, 0), (8, 256, 128, 1.3, -20), (8, 192, 128, 0.5, 40),
                                                        (8, 256, 144, 1.0, 9), (10, 192, 128, 0.8, 12), (10, 256, 128, 1.15, -6), (8, 192, 128, 0.25, 150)])
def test_weighted_reference_analysis_equals_reference_class(depth, width, height, gain, lift, seed=93, check_expectation=True):
    """LookaheadTLD::weightsAnalyse + weightCostLuma (slicetype.cpp:807-957) - the float guess (scale from the variance ratio, offset
    from the means), the two scored candidates, the denominator reduction, the 0.998 acceptance test and the weighting of the four
    lowres planes - restatement against the real class on fades of different strength (accepted, rejected, early exit, clamped
    offset)."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_weights_analyse"):
        pytest.skip("oracle/_ref predates x265ref_weights_analyse")
    y0, y1 = fade_pair(depth, width, height, gain, lift, seed=seed)
    cur, stride, org, w64, h64 = F.pad_plane(y1)
    ref = F.pad_plane(y0)[0]
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    lw, lh = wcu * 8, hcu * 8
    rstride = (width // 2 + 2 * F.MARGIN_X + 31) & ~31
    rows = lh + 2 * F.MARGIN_Y
    lorg = rstride * F.MARGIN_Y + F.MARGIN_X
    cplanes = O.lowres_init(depth, cur, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    rplanes = O.lowres_init(depth, ref, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    st = [lowres_stats(cplanes[0], rows, rstride, lorg, lw, lh), lowres_stats(rplanes[0], rows, rstride, lorg, lw, lh)]
    ssd = np.array([st[0][0], st[1][0]], np.uint64)
    sm = np.array([st[0][1], st[1][1]], np.uint64)
    out = np.zeros(4, np.int32)
    wpl = [np.zeros(rstride * rows, dtype=y0.dtype) for _ in range(4)]
    ricost = np.zeros(wcu * hcu, np.int32)
    lib.x265ref_weights_analyse.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8
    assert lib.x265ref_weights_analyse(cur.ctypes.data, ref.ctypes.data, width, height, ssd.ctypes.data, sm.ctypes.data, out.ctypes.data,
                                       *[p.ctypes.data for p in wpl], ricost.ctypes.data) == 0
    lam = {8: 1.0, 10: 16.0, 12: 256.0}[depth]
    icost, _, _ = O.lowres_intra(depth, cplanes[0], rstride, lorg, wcu, hcu, 5 * int(lam))
    assert np.array_equal(icost, ricost)
    weight, minscore, origscore = O.weights_analyse(depth, cplanes[0], rplanes[0], rstride, lorg, lw, lh, icost, ssd, sm)
    assert (weight is not None) == bool(out[0]), f"weighted: restatement {weight} (scores {minscore} / {origscore}), reference {out[0]}"
    expect_weighted = not (gain == 1.0 and abs(lift) < 1)
    assert not check_expectation or (weight is not None) == expect_weighted
    if weight is not None:
        assert minscore < origscore
        for i in range(4):
            a = O.weight_plane(depth, rplanes[i], weight).reshape(rows, rstride)[:, :F.MARGIN_X + lw]
            b = wpl[i].reshape(rows, rstride)[:, :F.MARGIN_X + lw]
            assert np.array_equal(a, b), f"weighted plane {i} differs ({weight})"


@pytest.mark.parametrize("depth,width,height,gain,lift", [(8, 256, 128, 0.75, 6), (8, 208, 144, 1.2, -12), (10, 192, 128, 0.8, 10), (8, 256, 128, 1.0, 0)])
def test_weighted_p_frame_cost_equals_reference_classes(depth, width, height, gain, lift):
    """--weightp in the lookahead, end to end: the real CostEstimateGroup::singleCost with bEnableWeightedPred runs weightsAnalyse and
    searches the weighted planes (slicetype.cpp:3136-3138,3222,3267); the restatement chains weights_analyse -> weight_plane ->
    lowres_cost with the weighted planes as the reference.  A fade with a small global motion (and the no-fade control)."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_lowres_cost_weightp"):
        pytest.skip("oracle/_ref predates x265ref_lowres_cost_weightp")
    y0, y1 = fade_pair(depth, width, height, gain, lift, seed=97)
    y1 = np.roll(y1, (2, -2), axis=(0, 1)).copy()
    cur, stride, org, w64, h64 = F.pad_plane(y1)
    ref = F.pad_plane(y0)[0]
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    lw, lh = wcu * 8, hcu * 8
    n = wcu * hcu
    rstride = (width // 2 + 2 * F.MARGIN_X + 31) & ~31
    rows = lh + 2 * F.MARGIN_Y
    lorg = rstride * F.MARGIN_Y + F.MARGIN_X
    cplanes = O.lowres_init(depth, cur, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    rplanes = O.lowres_init(depth, ref, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    st = [lowres_stats(cplanes[0], rows, rstride, lorg, lw, lh), lowres_stats(rplanes[0], rows, rstride, lorg, lw, lh)]
    ssd, sm = np.array([st[0][0], st[1][0]], np.uint64), np.array([st[0][1], st[1][1]], np.uint64)
    rmv, rmc, rlc = np.zeros((n, 2), np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint16)
    rrows, rframe, rw = np.zeros(hcu, np.int32), np.zeros(4, np.int64), np.zeros(1, np.int32)
    lib.x265ref_lowres_cost_weightp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8
    assert lib.x265ref_lowres_cost_weightp(cur.ctypes.data, ref.ctypes.data, width, height, ssd.ctypes.data, sm.ctypes.data, rmv.ctypes.data,
                                           rmc.ctypes.data, rlc.ctypes.data, rrows.ctypes.data, rframe.ctypes.data, rw.ctypes.data) == 0
    lam = {8: 1.0, 10: 16.0, 12: 256.0}[depth]
    icost, _, _ = O.lowres_intra(depth, cplanes[0], rstride, lorg, wcu, hcu, 5 * int(lam))
    weight, _, _ = O.weights_analyse(depth, cplanes[0], rplanes[0], rstride, lorg, lw, lh, icost, ssd, sm)
    assert (weight is not None) == bool(rw[0]) == (gain != 1.0)
    search_planes = rplanes if weight is None else [O.weight_plane(depth, p, weight) for p in rplanes]
    cq, qoff = F.qpel_cost_table(16, lam=lam, qmax=4 * (max(lw, lh) + 64))
    mvs, mvc, lc, rws, frame = O.lowres_cost(depth, cplanes[0], search_planes, rstride, lorg, wcu, hcu, cq, qoff, icost)
    assert np.array_equal(mvs, rmv), f"mvs differ at {np.flatnonzero((mvs != rmv).any(axis=1))[:8]}"
    assert np.array_equal(mvc, rmc) and np.array_equal(lc, rlc) and np.array_equal(rws, rrows)
    assert (frame[0], frame[1], frame[2]) == (rframe[1], rframe[2], rframe[3])
    if weight is not None:                                   # the weighting matters: the unweighted search gives another answer
        plain = O.lowres_cost(depth, cplanes[0], rplanes, rstride, lorg, wcu, hcu, cq, qoff, icost)
        assert plain[4][0] != frame[0]


@pytest.mark.parametrize("depth,width,height,qg,mode,strength,chroma", [(8, 256, 128, 16, 2, 1.0, True), (8, 256, 128, 16, 1, 1.0, True), (8, 208, 144, 16, 3, 0.8, True),
                                                                    (8, 256, 128, 8, 2, 1.0, True), (8, 192, 128, 8, 1, 1.5, True), (8, 250, 138, 16, 2, 1.0, False),
                                                                    (10, 192, 128, 16, 2, 1.0, True), (10, 192, 128, 16, 1, 0.6, False), (8, 256, 128, 16, 0, 1.0, True)])
def test_adaptive_quant_pass_equals_reference_class(depth, width, height, qg, mode, strength, chroma, seed=99):
    """LookaheadTLD::calcAdaptiveQuantFrame (slicetype.cpp:439-694): block energies through cu[].var (luma + 4:2:0 chroma), the
    double-precision QP offsets of AQ modes 1-3, invQscaleFactor through x265_exp2fix8, and the wp_sum / wp_ssd statistics the
    weight analysis reads - restatement against the real class."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_aq_frame"):
        pytest.skip("oracle/_ref predates x265ref_aq_frame")
    clip = F.synth_clip(width, height, 1, depth=depth, seed=seed)
    yimg, cbimg, crimg = clip[0]
    yp, stride, org, w64, h64 = F.pad_plane(yimg)
    cpad = [pad_any(np.ascontiguousarray(c), margin=16) for c in (cbimg, crimg)] if chroma else None
    n = ((width + qg - 1) // qg) * ((height + qg - 1) // qg)
    rqp, rinv = np.zeros(4 * n + 64, np.float64), np.zeros(4 * n + 64, np.int32)
    rsum, rssd, rn = np.zeros(3, np.uint64), np.zeros(3, np.uint64), np.zeros(1, np.int32)
    lib.x265ref_aq_frame.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 5
    cbc, crc = (np.ascontiguousarray(cbimg), np.ascontiguousarray(crimg)) if chroma else (None, None)
    assert lib.x265ref_aq_frame(yp.ctypes.data, None if cbc is None else cbc.ctypes.data, None if crc is None else crc.ctypes.data, width, height,
                                qg, mode, strength, 1, rqp.ctypes.data, rinv.ctypes.data, rsum.ctypes.data, rssd.ctypes.data, rn.ctypes.data) == 0
    assert rn[0] == n
    if chroma:
        energy, qp, inv, sm, ssd = O.aq_frame(depth, yp, stride, org, width, height, cpad[0][0], cpad[1][0], cpad[0][1], cpad[0][2], qg, mode, strength, True)
    else:
        energy, qp, inv, sm, ssd = O.aq_frame(depth, yp, stride, org, width, height, qg_size=qg, aq_mode=mode, aq_strength=strength, weightp=True)
    assert np.array_equal(sm, rsum) and np.array_equal(ssd, rssd), f"wp statistics: {sm} {ssd} vs {rsum} {rssd}"
    if mode:
        assert np.array_equal(qp, rqp[:n]), f"{np.count_nonzero(qp != rqp[:n])} QP offsets differ (max {np.abs(qp - rqp[:n]).max()})"
        assert np.array_equal(inv, rinv[:n])
        assert len(np.unique(inv)) > 4


@pytest.mark.parametrize("depth,width,height,qg,rng,chroma", [(8, 256, 128, 16, 1.0, True), (8, 208, 144, 32, 2.5, True), (8, 250, 138, 64, 1.0, False),
                                                              (8, 192, 136, 8, 6.0, True), (10, 192, 128, 16, 1.0, True), (10, 232, 120, 8, 3.0, False),
                                                              (12, 128, 80, 32, 1.0, True), (12, 136, 72, 64, 2.0, False), (8, 304, 206, 16, 4.0762005658023845, False), (10, 202, 94, 8, 5.123456789, False)])
def test_hevc_aq_pass_equals_reference_class(depth, width, height, qg, rng, chroma, seed=101):
    """--hevc-aq: LookaheadTLD::xPreanalyze / xPreanalyzeQp inside calcAdaptiveQuantFrame (slicetype.cpp:293-441, 507-511) - quadrant
    variances of every enabled layer's partitions (clipped at the picture edge), activities, the layers' QP offsets, invQscaleFactor from
    the deepest layer and the wp statistics gathered on the way - restatement against the real class.  Pictures whose sizes are not
    multiples of the partitions exercise the clipped quadrants."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_aq_hevc_frame"):
        pytest.skip("oracle/_ref predates x265ref_aq_hevc_frame")
    clip = F.synth_clip(width, height, 1, depth=depth, seed=seed)
    yimg, cbimg, crimg = clip[0]
    yp, stride, org, w64, h64 = F.pad_plane(yimg)
    cpad = [pad_any(np.ascontiguousarray(c), margin=16) for c in (cbimg, crimg)] if chroma else None
    if chroma:
        parts, act, qp, avg, inv, sm, ssd = O.aq_hevc_frame(depth, yp, stride, org, width, height, cpad[0][0], cpad[1][0], cpad[0][1], cpad[0][2], qg, rng, True)
    else:
        parts, act, qp, avg, inv, sm, ssd = O.aq_hevc_frame(depth, yp, stride, org, width, height, qg_size=qg, qp_adaptation_range=rng, weightp=True)
    total = int(parts.sum())
    rparts, ract, rqp, ravg = np.zeros(4, np.int32), np.zeros(total + 64, np.float64), np.zeros(total + 64, np.float64), np.zeros(4, np.float64)
    rinv, rsum, rssd = np.zeros(total + 64, np.int32), np.zeros(3, np.uint64), np.zeros(3, np.uint64)
    lib.x265ref_aq_hevc_frame.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 7
    cbc, crc = (np.ascontiguousarray(cbimg), np.ascontiguousarray(crimg)) if chroma else (None, None)
    assert lib.x265ref_aq_hevc_frame(yp.ctypes.data, None if cbc is None else cbc.ctypes.data, None if crc is None else crc.ctypes.data, width, height,
                                     qg, rng, 1, rparts.ctypes.data, ract.ctypes.data, rqp.ctypes.data, ravg.ctypes.data, rinv.ctypes.data,
                                     rsum.ctypes.data, rssd.ctypes.data) == 0
    assert np.array_equal(parts, rparts), f"layers {parts} vs {rparts}"
    assert [bool(v) for v in parts] == [bool(v) for v in O.AQ_LAYER_DEPTH[qg]]
    assert np.array_equal(act, ract[:total]), f"{np.count_nonzero(act != ract[:total])} activities differ"
    assert np.array_equal(avg, ravg), f"average activities {avg} vs {ravg}"
    assert np.array_equal(qp, rqp[:total]), f"{np.count_nonzero(qp != rqp[:total])} QP offsets differ (max {np.abs(qp - rqp[:total]).max()})"
    assert np.array_equal(inv, rinv[:inv.size]) and len(np.unique(inv)) > 3
    assert np.array_equal(sm, rsum) and np.array_equal(ssd, rssd), f"wp statistics: {sm} {ssd} vs {rsum} {rssd}"
    assert qp.min() < 0 < qp.max()


@pytest.mark.parametrize("depth,width,height,gain,lift", [(8, 256, 128, 0.8, 8), (8, 208, 144, 0.7, 10), (10, 192, 128, 0.7, 12)])
def test_weighted_b_frame_cost_equals_reference_classes(depth, width, height, gain, lift):
    """--weightp on a B picture: the list-0 search (predictor candidates, skip cost, motionEstimate) sees the weighted list-0 planes,
    the two bi-directional candidates the unweighted ones (slicetype.cpp:3222,3267,3328) - against the real singleCost(0, 2, 1) with
    bEnableWeightedPred.  List 0 is a brighter / darker version of the scene, list 1 the scene itself."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_lowres_cost_b_weightp"):
        pytest.skip("oracle/_ref predates x265ref_lowres_cost_b_weightp")
    y0, y1 = fade_pair(depth, width, height, gain, lift, seed=103)            # y0: list-0 reference, y1: current (faded y0)
    y1 = np.roll(y1, (2, -2), axis=(0, 1)).copy()
    y2 = np.roll(y1, (-2, 4), axis=(0, 1)).copy()                             # list 1: the current scene, moved
    y1[-64:-32, :64] = ((y0[-64:-32, :64].astype(np.int32) + y2[-64:-32, :64] + 1) >> 1).astype(y1.dtype)
    cur, stride, org, w64, h64 = F.pad_plane(y1)
    r0, r1 = F.pad_plane(y0)[0], F.pad_plane(y2)[0]
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    lw, lh = wcu * 8, hcu * 8
    n = wcu * hcu
    rstride = (width // 2 + 2 * F.MARGIN_X + 31) & ~31
    rows = lh + 2 * F.MARGIN_Y
    lorg = rstride * F.MARGIN_Y + F.MARGIN_X
    cplanes = O.lowres_init(depth, cur, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    p0 = O.lowres_init(depth, r0, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    p1 = O.lowres_init(depth, r1, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    st = [lowres_stats(cplanes[0], rows, rstride, lorg, lw, lh), lowres_stats(p0[0], rows, rstride, lorg, lw, lh)]
    ssd, sm = np.array([st[0][0], st[1][0]], np.uint64), np.array([st[0][1], st[1][1]], np.uint64)
    rmv = [np.zeros((n, 2), np.int32) for _ in range(2)]
    rmc = [np.zeros(n, np.int32) for _ in range(2)]
    rlc, rrows, rframe, rw = np.zeros(n, np.uint16), np.zeros(hcu, np.int32), np.zeros(4, np.int64), np.zeros(1, np.int32)
    lib.x265ref_lowres_cost_b_weightp.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 10
    assert lib.x265ref_lowres_cost_b_weightp(cur.ctypes.data, r0.ctypes.data, r1.ctypes.data, width, height, ssd.ctypes.data, sm.ctypes.data,
                                             rmv[0].ctypes.data, rmc[0].ctypes.data, rmv[1].ctypes.data, rmc[1].ctypes.data, rlc.ctypes.data,
                                             rrows.ctypes.data, rframe.ctypes.data, rw.ctypes.data) == 0
    lam = {8: 1.0, 10: 16.0, 12: 256.0}[depth]
    icost, _, _ = O.lowres_intra(depth, cplanes[0], rstride, lorg, wcu, hcu, 5 * int(lam))
    weight, _, _ = O.weights_analyse(depth, cplanes[0], p0[0], rstride, lorg, lw, lh, icost, ssd, sm)
    assert weight is not None and rw[0] == 1
    w0 = [O.weight_plane(depth, p, weight) for p in p0]
    cq, qoff = F.qpel_cost_table(16, lam=lam, qmax=4 * (max(lw, lh) + 64))
    mvs, mvc, lc, rws, frame = O.lowres_cost(depth, cplanes[0], w0, rstride, lorg, wcu, hcu, cq, qoff, icost, ref1_planes=p1, ref_bi_planes=p0)
    for i in range(2):
        assert np.array_equal(mvs[i], rmv[i]), f"list {i} mvs differ at {np.flatnonzero((mvs[i] != rmv[i]).any(axis=1))[:8]}"
        assert np.array_equal(mvc[i], rmc[i])
    assert np.array_equal(lc, rlc) and np.array_equal(rws, rrows)
    assert frame[3] == rframe[0] == rframe[1] and frame[1] == rframe[2]
    # both distinctions matter on this content: weighted planes for the search, unweighted for the bi-directional candidates
    allw = O.lowres_cost(depth, cplanes[0], w0, rstride, lorg, wcu, hcu, cq, qoff, icost, ref1_planes=p1)
    plain = O.lowres_cost(depth, cplanes[0], p0, rstride, lorg, wcu, hcu, cq, qoff, icost, ref1_planes=p1)
    assert not np.array_equal(allw[2], lc) and not np.array_equal(plain[2], lc)


def clip_duration(f):
    return min(max(f, 0.01), 1.0)          # CLIP_DURATION, ratecontrol.h:44-47


@pytest.mark.parametrize("depth,width,height,bframe,referenced,wbp,avg", [(8, 256, 128, False, True, 0, 1 / 30), (8, 256, 128, True, True, 0, 1 / 24),
                                                                         (8, 208, 144, True, False, 1, 1 / 30), (10, 192, 128, True, True, 1, 1 / 60),
                                                                         (8, 320, 192, False, False, 0, 0.5)])
def test_cutree_propagation_step_equals_reference_class(depth, width, height, bframe, referenced, wbp, avg, seed=105):
    """Lookahead::estimateCUPropagate + primitives.propagateCost (slicetype.cpp:2641-2753, pixel.cpp:914-940): the real class runs on
    the motion vectors / costs its own singleCost produced; the restatement gets those same inputs and must leave the same
    propagateCost in the references (bilinear split over four blocks, picture-border drops, bi-prediction weights, saturation)."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_cutree_propagate"):
        pytest.skip("oracle/_ref predates x265ref_cutree_propagate")
    clip = F.synth_clip(width, height, 3, depth=depth, seed=seed)
    rng = np.random.default_rng([15, depth, width, int(bframe), seed])
    y0, y2 = clip[0][0], clip[2][0]
    y1 = np.roll(y0, (5, -9), axis=(0, 1)).copy()
    y1[: height // 3] = np.roll(y2, (-7, 12), axis=(0, 1))[: height // 3]
    y1[-32:, -64:] = y0[-32:, -64:]
    y1[-64:-32, :64] = ((y0[-64:-32, :64].astype(np.int32) + y2[-64:-32, :64] + 1) >> 1).astype(y1.dtype)
    cur = F.pad_plane(y1)[0]
    r0, r1 = F.pad_plane(y0)[0], F.pad_plane(y2)[0]
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    n = wcu * hcu
    invq = rng.integers(128, 513, size=n).astype(np.int32)
    prop_cur = rng.integers(0, 30000, size=n).astype(np.uint16) if referenced else None
    pr0 = rng.integers(0, 65536, size=n).astype(np.uint16)
    pr0[::3] = 65533                                                       # a third of the blocks are about to saturate
    pr1 = rng.integers(0, 40000, size=n).astype(np.uint16)
    g0, g1 = pr0.copy(), pr1.copy()
    icost, lcost = np.zeros(n, np.int32), np.zeros(n, np.uint16)
    mv0, mv1 = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32)
    lib.x265ref_cutree_propagate.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + \
                                           [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 4
    assert lib.x265ref_cutree_propagate(cur.ctypes.data, r0.ctypes.data, r1.ctypes.data if bframe else None, width, height, invq.ctypes.data,
                                        None if prop_cur is None else prop_cur.ctypes.data, g0.ctypes.data, g1.ctypes.data, 30, 1, avg, wbp,
                                        icost.ctypes.data, lcost.ctypes.data, mv0.ctypes.data, mv1.ctypes.data) == 0
    fps_factor = clip_duration(1 / 30) / clip_duration(avg)
    # distScaleFactor / bipredWeight (slicetype.cpp:2644-2646) for b - p0 = 1, p1 - p0 = 2 (B) or 1 (P)
    span = 2 if bframe else 1
    dist_scale = ((1 << 8) + (span >> 1)) // span
    bipred = 64 - (dist_scale >> 2) if wbp else 32
    e0, e1 = O.cutree_propagate(depth, wcu, hcu, prop_cur, icost, lcost, invq, mv0, mv1 if bframe else None, fps_factor, bipred, pr0, pr1 if bframe else None)
    assert np.array_equal(e0, g0), f"list-0 reference: {np.count_nonzero(e0 != g0)} blocks differ"
    if bframe:
        assert np.array_equal(e1, g1), f"list-1 reference: {np.count_nonzero(e1 != g1)} blocks differ"
        assert (lcost >> 14 == 3).any()
    assert (g0 != pr0).any() and (g0 == 65535).any() and (mv0 != 0).any()


@pytest.mark.parametrize("width,height,avg,qcomp,dist,wdelta", [(256, 128, 1 / 30, 0.6, 0, 0.0), (416, 240, 1 / 24, 0.6, 1, 0.4), (256, 144, 0.2, 0.8, 2, 0.0),
                                                              (640, 360, 1 / 60, 0.5, 1, 1.0)])
def test_cutree_finish_equals_reference_class(width, height, avg, qcomp, dist, wdelta):
    """Lookahead::cuTreeFinish (slicetype.cpp:2889-2937): the restatement AND the library's host-side entry x265hip_cutree_finish
    against the real class on random per-block arrays (zero intra costs, saturated propagate costs, the weighted-cost delta)."""
    import importlib
    import oracle_api as O
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    lib = _ref(8)
    if not hasattr(lib, "x265ref_cutree_finish"):
        pytest.skip("oracle/_ref predates x265ref_cutree_finish")
    rng = np.random.default_rng([17, width, height])
    n = ((width // 2 + 7) >> 3) * ((height // 2 + 7) >> 3)
    intra = rng.integers(0, 9000, size=n).astype(np.int32)
    intra[rng.random(n) < 0.05] = 0
    invq = rng.integers(64, 1024, size=n).astype(np.int32)
    prop = rng.integers(0, 65536, size=n).astype(np.uint16)
    prop[::5] = 65535
    qpaq = rng.normal(0, 2, size=n)
    preset = rng.normal(0, 1, size=n)
    got = preset.copy()
    lib.x265ref_cutree_finish.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                                                                  ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    assert lib.x265ref_cutree_finish(width, height, intra.ctypes.data, invq.ctypes.data, prop.ctypes.data, qpaq.ctypes.data, 30, 1, avg, qcomp, dist, wdelta,
                                     got.ctypes.data) == n
    fps_q8 = int(clip_duration(avg) / clip_duration(1 / 30) * 256)
    weight_delta = (1.0 - wdelta) if (dist and wdelta > 0) else 0.0
    strength = 5.0 * (1.0 - qcomp)
    exp = O.cutree_finish(8, intra, invq, prop, qpaq, fps_q8, weight_delta, strength, preset)
    assert np.array_equal(exp, got), f"{np.count_nonzero(exp != got)} offsets differ from the real class"
    mine = A.cutree_finish(intra, invq, prop, qpaq, fps_q8, weight_delta, strength, preset)
    assert np.array_equal(mine, got), f"{np.count_nonzero(mine != got)} offsets of the library entry differ from the real class"
    assert (got == preset).any() and (got != preset).any()


@pytest.mark.parametrize("width,height", [(256, 128), (416, 240), (48, 32), (32, 160), (1280, 720)])
def test_frame_cost_recalculate_equals_reference_class(width, height):
    """Lookahead::frameCostRecalculate (slicetype.cpp:2941-3011, P pictures): the restatement and the library's host-side entry
    against the real class (row sums, interior-only score, the at-most-two-blocks rule, x265_exp2fix8's clamps)."""
    import importlib
    import oracle_api as O
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    lib = _ref(8)
    if not hasattr(lib, "x265ref_frame_cost_recalculate"):
        pytest.skip("oracle/_ref predates x265ref_frame_cost_recalculate")
    rng = np.random.default_rng([19, width, height])
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    n = wcu * hcu
    lc = (rng.integers(0, 16384, size=n) | (rng.integers(0, 4, size=n) << 14)).astype(np.uint16)
    qp = rng.normal(0, 3, size=n)
    qp[::11] = 60.0; qp[5::13] = -60.0                                # beyond both clamps of x265_exp2fix8
    rrows, rscore = np.zeros(hcu, np.int32), np.zeros(1, np.int64)
    lib.x265ref_frame_cost_recalculate.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    assert lib.x265ref_frame_cost_recalculate(width, height, lc.ctypes.data, qp.ctypes.data, rrows.ctypes.data, rscore.ctypes.data) == n
    score, rows = O.frame_cost_recalculate(8, wcu, hcu, lc, qp)
    assert score == int(rscore[0]) and np.array_equal(rows, rrows)
    score2, rows2 = A.frame_cost_recalculate(wcu, hcu, lc, qp)
    assert score2 == int(rscore[0]) and np.array_equal(rows2, rrows)


@pytest.mark.parametrize("width,height,avg,qcomp,dist,wdelta", [(256, 128, 1 / 30, 0.6, 0, 0.0), (416, 240, 1 / 24, 0.6, 1, 0.4), (640, 360, 1 / 60, 0.5, 1, 1.0)])
def test_cutree_finish_and_recalculation_qg8_equal_reference_class(width, height, avg, qcomp, dist, wdelta):
    """--qg-size 8: the branches of Lookahead::cuTreeFinish / frameCostRecalculate that keep the offsets on the full-resolution 8x8 grid
    (slicetype.cpp:2903-2921, 2990-3002) - restatement and the library's host-side entries against the real class."""
    import importlib
    import oracle_api as O
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    lib = _ref(8)
    if not hasattr(lib, "x265ref_cutree_finish_qg8"):
        pytest.skip("oracle/_ref predates x265ref_cutree_finish_qg8")
    rng = np.random.default_rng([29, width, height])
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    n = wcu * hcu
    intra = rng.integers(0, 9000, size=n).astype(np.int32)
    intra[rng.random(n) < 0.05] = 0
    intra[rng.random(n) < 0.05] = 3                                  # a quarter of it is 0
    invq = rng.integers(64, 1024, size=n).astype(np.int32)
    prop = rng.integers(0, 65536, size=n).astype(np.uint16)
    prop[::5] = 65535
    qpaq = rng.normal(0, 2, size=4 * n)
    preset = rng.normal(0, 1, size=4 * n)
    got = preset.copy()
    lib.x265ref_cutree_finish_qg8.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                                                                      ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    assert lib.x265ref_cutree_finish_qg8(width, height, intra.ctypes.data, invq.ctypes.data, prop.ctypes.data, qpaq.ctypes.data, 30, 1, avg, qcomp, dist, wdelta,
                                         got.ctypes.data) == n
    fps_q8 = int(clip_duration(avg) / clip_duration(1 / 30) * 256)
    weight_delta = (1.0 - wdelta) if (dist and wdelta > 0) else 0.0
    strength = 5.0 * (1.0 - qcomp)
    exp = O.cutree_finish_qg8(8, wcu, hcu, intra, invq, prop, qpaq, fps_q8, weight_delta, strength, preset)
    assert np.array_equal(exp, got), f"{np.count_nonzero(exp != got)} offsets differ from the real class"
    mine = A.cutree_finish_qg8(wcu, hcu, intra, invq, prop, qpaq, fps_q8, weight_delta, strength, preset)
    assert np.array_equal(mine, got), f"{np.count_nonzero(mine != got)} offsets of the library entry differ from the real class"
    assert (got == preset).any() and (got != preset).any()
    # the recalculation on those offsets
    lc = (rng.integers(0, 16384, size=n) | (rng.integers(0, 4, size=n) << 14)).astype(np.uint16)
    qp = got.copy()
    qp[::11] = 60.0; qp[5::13] = -60.0
    rrows, rscore = np.zeros(hcu, np.int32), np.zeros(1, np.int64)
    lib.x265ref_frame_cost_recalculate_qg8.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    assert lib.x265ref_frame_cost_recalculate_qg8(width, height, lc.ctypes.data, qp.ctypes.data, rrows.ctypes.data, rscore.ctypes.data) == n
    score, rows = O.frame_cost_recalculate_qg8(8, wcu, hcu, lc, qp)
    assert score == int(rscore[0]) and np.array_equal(rows, rrows)
    score2, rows2 = A.frame_cost_recalculate_qg8(wcu, hcu, lc, qp)
    assert score2 == int(rscore[0]) and np.array_equal(rows2, rrows)


@pytest.mark.parametrize("width,height,qg,avg,qcomp,dist,wdelta", [(256, 128, 16, 1 / 30, 0.6, 0, 0.0), (416, 240, 32, 1 / 24, 0.6, 1, 0.4), (250, 138, 16, 1 / 60, 0.5, 1, 1.0),
                                                                 (640, 360, 32, 0.2, 0.8, 2, 0.0)])
def test_cutree_finish_with_hevc_aq_equals_reference_class(width, height, qg, avg, qcomp, dist, wdelta):
    """cuTreeFinish with rc.hevcAq = Lookahead::computeCUTreeQpOffset (slicetype.cpp:2749-2887): every layer's dCuTreeOffset from its
    dQpOffset and the block costs, strength 6 * (1 - qCompress); restatement and the library's host-side entry against the real class.
    Blocks with a zero intra cost make their partition infinite / NaN upstream - kept (compared with NaN == NaN)."""
    import importlib
    import oracle_api as O
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    lib = _ref(8)
    if not hasattr(lib, "x265ref_cutree_finish_hevc_aq"):
        pytest.skip("oracle/_ref predates x265ref_cutree_finish_hevc_aq")
    rng = np.random.default_rng([31, width, height])
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    n = wcu * hcu
    intra = rng.integers(1, 9000, size=n).astype(np.int32)
    intra[rng.random(n) < 0.01] = 0
    invq = rng.integers(64, 1024, size=n).astype(np.int32)
    prop = rng.integers(0, 65536, size=n).astype(np.uint16)
    prop[::7] = 0
    layers = [d for d in range(4) if O.AQ_LAYER_DEPTH[qg][d]]
    counts = [((width + (64 >> d) - 1) // (64 >> d)) * ((height + (64 >> d) - 1) // (64 >> d)) for d in layers]
    qp_in = rng.normal(0, 2, size=sum(counts))
    got, parts = np.zeros(sum(counts)), np.zeros(4, np.int32)
    lib.x265ref_cutree_finish_hevc_aq.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                                                                             ctypes.c_double] + [ctypes.c_void_p] * 3
    assert lib.x265ref_cutree_finish_hevc_aq(width, height, qg, intra.ctypes.data, invq.ctypes.data, prop.ctypes.data, 30, 1, avg, qcomp, dist, wdelta,
                                             qp_in.ctypes.data, got.ctypes.data, parts.ctypes.data) == n
    assert [int(parts[d]) for d in layers] == counts
    fps_q8 = int(clip_duration(avg) / clip_duration(1 / 30) * 256)
    weight_delta = (1.0 - wdelta) if (dist and wdelta > 0) else 0.0
    strength = 6.0 * (1.0 - qcomp)
    at = 0
    for d, cnt in zip(layers, counts):
        args = (width, height, 64 >> d, wcu, intra, invq, prop, fps_q8, weight_delta, strength, qp_in[at:at + cnt])
        exp = O.cutree_finish_hevc_aq(8, *args)
        assert np.array_equal(exp, got[at:at + cnt], equal_nan=True), f"layer {d}: {np.count_nonzero(exp != got[at:at + cnt])} offsets differ from the real class"
        mine = A.cutree_finish_hevc_aq(*args)
        assert np.array_equal(mine, got[at:at + cnt], equal_nan=True), f"layer {d}: the library entry differs from the real class"
        assert np.isfinite(exp).sum() > cnt // 2
        at += cnt


def sao_case(depth, width, height, seed):
    """Source / deblocked-like pair and random per-CTU SAO parameters (all five types, off, merge-left runs)."""
    rng = np.random.default_rng([21, depth, width, seed])
    y = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0][0]
    sm = (y.astype(np.int32) * 2 + np.roll(y, 1, axis=1) + np.roll(y, 1, axis=0) + 2) >> 2
    rec = np.clip(sm + (rng.integers(-3, 4, size=y.shape) << (depth - 8)), 0, (1 << depth) - 1).astype(y.dtype)
    rec[: height // 4, : width // 4] = y[: height // 4, : width // 4]                    # zero differences
    rec[-9:, -70:] = (1 << depth) - 1                                                     # clipping at the top of the range
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    cw = (width + 63) // 64
    lim = 7 if depth == 8 else 31
    params = np.zeros((nctu, 7), np.int32)
    params[:, 0] = rng.integers(-1, 5, size=nctu)
    params[:, 1] = rng.integers(0, 32, size=nctu)
    params[:, 2:6] = rng.integers(-lim, lim + 1, size=(nctu, 4))
    for a in range(nctu):
        if a % cw and rng.random() < 0.3:
            params[a] = params[a - 1]
            params[a, 6] = 1
    return y, rec, params


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 200, 150), (10, 192, 136), (8, 64, 64), (12, 192, 136)])
def test_sao_restatement_equals_reference_class(depth, width, height):
    """oracle/x265_oracle_pipeline4.c's SAO passes against the real SAO class (oracle/ref_sao.cpp): calcSaoStatsCTU's count /
    offsetOrg of every CTU, type and class, and the picture after generateLumaOffsets over all CTUs - including pictures that
    are not a multiple of the CTU size, merge-left CTUs and clipping."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_sao"):
        pytest.skip("oracle/_ref predates ref_sao.cpp")
    y, rec, params = sao_case(depth, width, height, 5)
    fenc, stride, org, w64, h64 = F.pad_plane(y)
    recp = F.pad_plane(rec)[0]
    nctu = params.shape[0]
    rcnt, roff = np.zeros((nctu, 5, 32), np.int32), np.zeros((nctu, 5, 32), np.int32)
    rout = recp.copy()
    lib.x265ref_sao.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.x265ref_sao(fenc.ctypes.data, rout.ctypes.data, width, height, params.ctypes.data, rcnt.ctypes.data, roff.ctypes.data) == 0
    cnt, off = O.sao_stats(depth, fenc, recp, stride, org, width, height)
    assert np.array_equal(cnt, rcnt), f"count differs in CTU/type {np.argwhere((cnt != rcnt).any(axis=2))[:6].tolist()}"
    assert np.array_equal(off, roff), f"offsetOrg differs in CTU/type {np.argwhere((off != roff).any(axis=2))[:6].tolist()}"
    if hasattr(lib, "x265ref_sao_last_initial_offsets"):          # SAO::saoStatsInitialOffset (sao.cpp:1378-1433) on the same statistics
        rinit = np.zeros((nctu, 5, 32), np.int32)
        lib.x265ref_sao_last_initial_offsets.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert lib.x265ref_sao_last_initial_offsets(rinit.ctypes.data, nctu) == 0
        init, chosen = O.sao_decide(depth, cnt, off)
        assert np.array_equal(init[:, :4, 1:5], rinit[:, :4, 1:5]) and np.array_equal(init[:, 4, :], rinit[:, 4, :]), "initial offsets differ"
        assert (chosen[:, 0] >= 0).any() and (init != 0).any()
    out = O.sao_apply(depth, recp, stride, org, width, height, params)
    rows = out.size // stride
    a = out.reshape(rows, stride)[org // stride: org // stride + height, org % stride: org % stride + width]
    b = rout.reshape(rows, stride)[org // stride: org // stride + height, org % stride: org % stride + width]
    assert np.array_equal(a, b), f"{np.count_nonzero(a != b)} samples differ, first at {np.argwhere(a != b)[:4].tolist()}"
    assert (a != recp.reshape(rows, stride)[org // stride: org // stride + height, org % stride: org % stride + width]).any()


@pytest.mark.parametrize("depth,level,qp,offs", [(8, 2, 32, (0, 0)), (8, 1, 30, (0, 0)), (8, 0, 27, (2, -1)), (10, 2, 44, (0, 0)), (10, 1, 36, (-2, 3)), (12, 1, 40, (1, -1))])
def test_deblock_restatement_equals_reference_class(depth, level, qp, offs):
    """oracle/x265_oracle_pipeline4.c's boundary strengths + luma edge filter against the real Deblock::deblockCTU
    (oracle/ref_deblock.cpp) on a reconstruction the oracle chain itself produced (search -> sub-pel -> prediction / residual round
    trip), cut into 2Nx2N inter CUs of one size with the chain's mvs and coded flags: the filtered picture must be identical."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_deblock"):
        pytest.skip("oracle/_ref predates ref_deblock.cpp")
    R, subme = 8, 2
    clip = F.synth_clip(192, 128, 2, depth=depth, seed=71 + level)
    cur, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    nctu = (w64 // 64) * (h64 // 64)
    cost = F.mv_cost_table(R)
    cq, qoff = F.qpel_cost_table(R)
    _, best = O.me_fullsearch(depth, cur, stride, org, ref, stride, org, w64, h64, R, 0, nctu, cost, cost, want_surf=False)
    mv = O.subpel_refine(depth, cur, stride, org, ref, stride, org, w64, h64, R, 0, nctu, best, cq, qoff, subme)
    rec, _, ns, _ = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv, qp)
    bv, bh = O.deblock_bs_inter(depth, w64, h64, level, mv, ns)
    exp = O.deblock_luma(depth, rec.reshape(-1), stride, org, w64, h64, bv, bh, qp, beta_offset_div2=offs[0], tc_offset_div2=offs[1])
    got = np.ascontiguousarray(rec.reshape(-1)).copy()
    m = np.ascontiguousarray(mv, dtype=np.int32)
    n = np.ascontiguousarray(ns, dtype=np.uint32)
    lib.x265ref_deblock.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3
    assert lib.x265ref_deblock(got.ctypes.data, w64, h64, level, m.ctypes.data, n.ctypes.data, qp, offs[0], offs[1]) == 0
    assert np.array_equal(got, exp.reshape(-1)), f"{np.count_nonzero(got != exp.reshape(-1))} samples differ"
    assert np.count_nonzero(exp.reshape(-1) != rec.reshape(-1)) > 50 and bv.any() and bh.any()


@pytest.mark.parametrize("depth,level,qp,offs,cq", [(8, 2, 32, (0, 0), (0, 0)), (8, 1, 36, (1, -2), (3, -4)), (8, 0, 30, (0, 0), (-2, 6)), (10, 1, 33, (0, 2), (1, 1))])
def test_deblock_with_intra_blocks_and_chroma_equals_reference_class(depth, level, qp, offs, cq):
    """Mixed intra / inter pictures in 4:2:0: Bs 2 on the edges of intra CUs, luma filtered with the Bs-2 tc, and the chroma planes
    (edgeFilterChroma: Bs 2 edges on the 8-sample chroma grid, chroma QP mapping, PPS chroma offsets) - against the real class."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_deblock420"):
        pytest.skip("oracle/_ref predates x265ref_deblock420")
    R, subme = 8, 2
    clip = F.synth_clip(192, 128, 2, depth=depth, seed=75 + level)
    cur, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    nctu = (w64 // 64) * (h64 // 64)
    cost = F.mv_cost_table(R)
    cqt, qoff = F.qpel_cost_table(R)
    _, best = O.me_fullsearch(depth, cur, stride, org, ref, stride, org, w64, h64, R, 0, nctu, cost, cost, want_surf=False)
    mv = O.subpel_refine(depth, cur, stride, org, ref, stride, org, w64, h64, R, 0, nctu, best, cqt, qoff, subme)
    rec, _, ns, _ = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv, qp + 6 * (depth - 8))
    rng = np.random.default_rng([41, depth, level])
    npu = (64 >> (3 + level)) ** 2
    intra = (rng.random((nctu, npu)) < 0.3).astype(np.uint8)
    # chroma planes of the reconstruction: the clip's chroma with coding-like steps at 8-sample boundaries
    cw, ch = w64 // 2, h64 // 2
    chroma = []
    for c in (1, 2):
        pl = np.zeros((ch, cw), dtype=cur.dtype)
        src = clip[1][c]
        pl[:src.shape[0], :src.shape[1]] = src
        pl[src.shape[0]:, :] = pl[src.shape[0] - 1]
        pl[:, src.shape[1]:] = pl[:, src.shape[1] - 1:src.shape[1]]
        steps = rng.integers(-6, 7, size=(ch // 8, cw // 8)) << (depth - 8)
        pl = np.clip(pl.astype(np.int32) + np.kron(steps, np.ones((8, 8), np.int32)), 0, (1 << depth) - 1).astype(cur.dtype)
        chroma.append(np.ascontiguousarray(pl))
    bv, bh = O.deblock_bs_inter(depth, w64, h64, level, mv, ns, intra=intra)
    assert (bv == 2).any() and (bh == 2).any()
    exp_y = O.deblock_luma(depth, rec.reshape(-1), stride, org, w64, h64, bv, bh, qp, beta_offset_div2=offs[0], tc_offset_div2=offs[1])
    pads = [pad_any(c, margin=16) for c in chroma]
    exp_cb, exp_cr = O.deblock_chroma(depth, pads[0][0], pads[1][0], pads[0][1], pads[0][2], w64, h64, bv, bh, qp, cb_qp_offset=cq[0],
                                      cr_qp_offset=cq[1], tc_offset_div2=offs[1])
    got_y = np.ascontiguousarray(rec.reshape(-1)).copy()
    got_c = [c.copy() for c in chroma]
    m, n = np.ascontiguousarray(mv, dtype=np.int32), np.ascontiguousarray(ns, dtype=np.uint32)
    lib.x265ref_deblock420.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5
    assert lib.x265ref_deblock420(got_y.ctypes.data, got_c[0].ctypes.data, got_c[1].ctypes.data, w64, h64, level, m.ctypes.data, n.ctypes.data,
                                  intra.ctypes.data, qp, offs[0], offs[1], cq[0], cq[1]) == 0
    assert np.array_equal(got_y, exp_y.reshape(-1)), f"luma: {np.count_nonzero(got_y != exp_y.reshape(-1))} samples differ"
    for c, (exp, name) in enumerate(((exp_cb, "Cb"), (exp_cr, "Cr"))):
        e2 = exp.reshape(-1, pads[c][1])[16:16 + ch, 16:16 + cw]
        assert np.array_equal(got_c[c], e2), f"{name}: {np.count_nonzero(got_c[c] != e2)} samples differ"
        assert np.count_nonzero(e2 != chroma[c]) > 20


def motion_field_case(depth, level, seed, qp):
    """A reconstruction from the oracle chain plus a random two-list motion field for it: list-1 mvs a small perturbation of the
    list-0 ones (so both sides of the 4-quarter-sample threshold occur), reference picture ids in -1..2 per list (at least one list
    used), sparse coded flags so that most edges are decided by the motion comparison."""
    import oracle_api as O
    R, subme = 8, 2
    clip = F.synth_clip(192, 128, 2, depth=depth, seed=seed)
    cur, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    nctu = (w64 // 64) * (h64 // 64)
    cost = F.mv_cost_table(R)
    cqt, qoff = F.qpel_cost_table(R)
    _, best = O.me_fullsearch(depth, cur, stride, org, ref, stride, org, w64, h64, R, 0, nctu, cost, cost, want_surf=False)
    mv0 = O.subpel_refine(depth, cur, stride, org, ref, stride, org, w64, h64, R, 0, nctu, best, cqt, qoff, subme)
    rec, _, ns, _ = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv0, qp + 6 * (depth - 8))
    rng = np.random.default_rng([43, depth, level, seed])
    npu = (64 >> (3 + level)) ** 2
    mv0 = np.ascontiguousarray(mv0, dtype=np.int32).reshape(-1, 2).copy()
    # neighbouring blocks often share a vector in real pictures: quantise the field so that equal / near-equal pairs are common
    qx = ((mv0[:, 1] << 16 >> 16) // 8) * 8
    qy = ((mv0[:, 1] >> 16) // 8) * 8
    d = rng.integers(-4, 5, size=(len(mv0), 2))
    mv1 = mv0.copy()
    mv0[:, 1] = (qx & 0xffff) | (qy << 16)
    mv1[:, 1] = ((qx + d[:, 0]) & 0xffff) | ((qy + d[:, 1]) << 16)
    combos = np.array([(0, 1), (1, 0), (0, 0), (0, -1), (-1, 0), (1, 1), (2, 1), (-1, 2)], dtype=np.int8)   # few pictures: every branch of :231-246 occurs
    pick = rng.choice(len(combos), size=(nctu, npu), p=[0.3, 0.2, 0.2, 0.1, 0.05, 0.05, 0.05, 0.05])
    ref0, ref1 = np.ascontiguousarray(combos[pick, 0]), np.ascontiguousarray(combos[pick, 1])
    ns = (np.asarray(ns).reshape(nctu, npu) * (rng.random((nctu, npu)) < 0.15)).astype(np.uint32)
    intra = (rng.random((nctu, npu)) < 0.05).astype(np.uint8)
    return rec, stride, org, w64, h64, mv0, mv1, ref0, ref1, ns, intra


@pytest.mark.parametrize("depth,level,slice_b,qp", [(8, 0, 1, 30), (8, 1, 1, 34), (8, 2, 1, 28), (8, 0, 0, 32), (8, 1, 0, 30), (10, 0, 1, 33), (10, 1, 0, 36)])
def test_deblock_b_picture_boundary_strengths_equal_reference_class(depth, level, slice_b, qp, seed=None, check_coverage=True):
    """getBoundaryStrength in full (deblock.cpp:217-247): several reference pictures per list and the B-picture four-way comparison
    of (ref0, ref1) x (mv0, mv1) - the restatement's Bs maps + luma filter against the real Deblock::deblockCTU with B_SLICE /
    P_SLICE, distinct Frame objects per picture id, m_refIdx / m_mv of both lists filled."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_deblock_b"):
        pytest.skip("oracle/_ref predates x265ref_deblock_b")
    rec, stride, org, w64, h64, mv0, mv1, ref0, ref1, ns, intra = motion_field_case(depth, level, 81 + level if seed is None else seed, qp)
    if not slice_b:
        ref0 = np.maximum(ref0, 0)                      # P pictures: every inter block uses list 0
    bv, bh = O.deblock_bs_b(depth, w64, h64, level, mv0, mv1, ref0, ref1, ns, slice_b=slice_b, intra=intra)
    both = np.concatenate([bv, bh])
    assert not check_coverage or ((both == 0).any() and (both == 1).any() and (both == 2).any())
    exp = O.deblock_luma(depth, rec.reshape(-1), stride, org, w64, h64, bv, bh, qp)
    got = np.ascontiguousarray(rec.reshape(-1)).copy()
    lib.x265ref_deblock_b.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3
    assert lib.x265ref_deblock_b(got.ctypes.data, w64, h64, level, slice_b, mv0.ctypes.data, mv1.ctypes.data, ref0.ctypes.data, ref1.ctypes.data,
                                 ns.ctypes.data, intra.ctypes.data, qp, 0, 0) == 0
    assert np.array_equal(got, exp.reshape(-1)), f"{np.count_nonzero(got != exp.reshape(-1))} samples differ"
    # the motion comparison matters: the P-only single-reference rule gives different maps on this field
    pv, ph = O.deblock_bs_inter(depth, w64, h64, level, mv0, ns, intra=intra)
    assert not check_coverage or not (np.array_equal(pv, bv) and np.array_equal(ph, bh))


@pytest.mark.parametrize("chroma", [False, True])
@pytest.mark.parametrize("depth,n,qp,islice", [(8, 4, 24, 1), (8, 4, 30, 0), (8, 8, 27, 1), (8, 16, 33, 0), (8, 32, 22, 1), (8, 32, 45, 0),
                                               (10, 4, 36, 1), (10, 16, 40, 1), (10, 32, 30, 0)])
def test_intra_tu_round_trip_equals_reference_quant_class(depth, n, qp, islice, chroma):
    """The residual half of the intra TU stage (oracle/x265_oracle_pipeline2.c::x265oracle_intra_recon) against the real
    Quant::transformNxN + Quant::invtransformNxN (oracle/ref_quant.cpp): DC prediction from flat neighbours makes the prediction a
    constant, so levels, numSig and the reconstruction can be compared block by block (DST-VII for 4x4, DC shortcut, empty blocks).
    chroma: the 4:2:0 chroma flavour (x265oracle_intra_recon_chroma) against the same class with TEXT_CHROMA_U - the 4x4 TU takes
    the DCT there."""
    import oracle_api as O
    lib = _ref(depth)
    entry = "x265ref_tu_roundtrip_chroma" if chroma else "x265ref_tu_roundtrip"
    if not hasattr(lib, entry):
        pytest.skip("oracle/_ref predates " + entry)
    rng = np.random.default_rng([31, depth, n, qp])
    dt = np.uint8 if depth == 8 else np.uint16
    pmax, v = (1 << depth) - 1, 1 << (depth - 1)
    ntu = 40
    W = n * ntu
    yy, xx = np.mgrid[0:n, 0:W]
    amp = np.repeat(rng.choice([0.0, 0.02, 0.2, 0.45], size=ntu), n)[None, :]
    src = np.clip(np.rint(v + amp * pmax * np.sin(xx / 3.0 + yy / 2.0) + rng.normal(0, 1.5 * (1 << (depth - 8)), (n, W))), 0, pmax).astype(dt)
    src[:, :n] = v                                           # an empty residual
    src[:, n:2 * n] = v + (3 << (depth - 8))                 # a DC-only residual
    nbw = 4 * n + 1
    nb = np.full(2 * nbw, v, dtype=dt)
    jobs = np.zeros(ntu, dtype=np.dtype([("off", "<i8", 4), ("arg", "<i4", 4)]))
    for t in range(ntu):
        jobs["off"][t] = (t * n, 0, nbw, t * n * n)
        jobs["arg"][t, 0] = 1                                # DC mode
    rec, lev, ns, _ = O.intra_recon(depth, n, src.reshape(-1), W, nb, ntu * n * n, n, qp, islice, jobs, chroma=chroma)
    resi = np.stack([src[:, t * n:(t + 1) * n].astype(np.int16) - v for t in range(ntu)]).reshape(-1)
    rlev, rns, rout = np.zeros(ntu * n * n, np.int16), np.zeros(ntu, np.uint32), np.zeros(ntu * n * n, np.int16)
    fn = getattr(lib, entry)
    fn.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
    assert fn(resi.ctypes.data, n, qp, 1, islice, ntu, rlev.ctypes.data, rns.ctypes.data, rout.ctypes.data) == 0
    assert np.array_equal(ns, rns) and np.array_equal(lev, rlev)
    assert np.array_equal(rec.astype(np.int32), np.clip(v + rout.astype(np.int32), 0, pmax))
    assert (rns == 0).any() and (rns > 1).any()
    if n == 4:                                               # the transform really differs between the two flavours at 4x4
        other = O.intra_recon(depth, n, src.reshape(-1), W, nb, ntu * n * n, n, qp, islice, jobs, chroma=not chroma)[1]
        assert not np.array_equal(other, lev)


@pytest.mark.parametrize("depth,n", [(8, 4), (8, 8), (8, 16), (8, 32), (10, 8)])
def test_chroma_intra_prediction_uses_unfiltered_neighbours_without_edge_smoothing(depth, n, repo_root):
    """Predict::predIntraChromaAng (predict.cpp:590-598, 4:2:0): intra_pred[mode](dst, stride, intraNeighbourBuf[0], mode, 0) - the
    chroma flavour's prediction (recovered from a lossless-enough candidate: source == prediction gives an empty residual and the
    reconstruction IS the prediction) equals the table primitive called that way, for every mode."""
    import oracle_api as O
    rng = np.random.default_rng([33, depth, n])
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    nbw = 4 * n + 1
    unf = rng.integers(0, pmax + 1, size=nbw).astype(dt)
    flt = rng.integers(0, pmax + 1, size=nbw).astype(dt)      # a decoy: the chroma flavour must never read it
    nb = np.concatenate([unf, flt])
    import harness as H
    orc = H.load_oracle(depth, repo_root)
    ci = {4: 0, 8: 1, 16: 2, 32: 3}[n]
    for mode in range(35):
        src = np.zeros((n, n), dtype=dt)
        orc.fn(f"cu[{ci}].intra_pred[{mode}]")(H.ptr(src), n, H.ptr(unf), mode, 0)
        jobs = np.zeros(1, dtype=np.dtype([("off", "<i8", 4), ("arg", "<i4", 4)]))
        jobs["off"][0] = (0, 0, nbw, 0)
        jobs["arg"][0, 0] = mode
        rec, lev, ns, dist = O.intra_recon(depth, n, src.reshape(-1), n, nb, n * n, n, 30, 1, jobs, chroma=True)
        assert ns[0] == 0 and dist[0] == 0 and np.array_equal(rec.reshape(n, n), src), f"mode {mode}"


@pytest.mark.parametrize("depth,level,qp", [(8, 2, 30), (8, 0, 24), (10, 1, 38), (12, 1, 50)])
def test_inter_tu_round_trip_equals_reference_quant_class(depth, level, qp):
    """Same for the inter TU stage (x265oracle_inter_recon) with zero motion, where the prediction is the reference picture itself."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_tu_roundtrip"):
        pytest.skip("oracle/_ref predates ref_quant.cpp")
    clip = F.synth_clip(128, 64, 2, depth=depth, seed=81)
    cur, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    nctu = (w64 // 64) * (h64 // 64)
    n = 8 << level
    nblk = (64 // n) ** 2
    mv = np.zeros((nctu * 85, 2), np.int32)
    rec, lev, ns, _ = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv, qp)
    rows = cur.size // stride
    c2, r2, o2 = cur.reshape(rows, stride), ref.reshape(rows, stride), rec.reshape(rows, stride)
    y0, x0 = org // stride, org % stride
    resi, pos = [], []
    for ctu in range(nctu):
        for z in range(nblk):
            bx = sum(((z >> (2 * b)) & 1) << b for b in range(3))
            by = sum(((z >> (2 * b + 1)) & 1) << b for b in range(3))
            y, x = y0 + (ctu // (w64 // 64)) * 64 + by * n, x0 + (ctu % (w64 // 64)) * 64 + bx * n
            resi.append(c2[y:y + n, x:x + n].astype(np.int16) - r2[y:y + n, x:x + n].astype(np.int16))
            pos.append((y, x))
    resi = np.ascontiguousarray(np.stack(resi).reshape(-1))
    nj = len(pos)
    rlev, rns, rout = np.zeros(nj * n * n, np.int16), np.zeros(nj, np.uint32), np.zeros(nj * n * n, np.int16)
    lib.x265ref_tu_roundtrip.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
    assert lib.x265ref_tu_roundtrip(resi.ctypes.data, n, qp, 0, 0, nj, rlev.ctypes.data, rns.ctypes.data, rout.ctypes.data) == 0
    assert np.array_equal(ns.reshape(-1), rns) and np.array_equal(lev, rlev)
    pmax = (1 << depth) - 1
    for j, (y, x) in enumerate(pos):
        exp = np.clip(r2[y:y + n, x:x + n].astype(np.int32) + rout[j * n * n:(j + 1) * n * n].reshape(n, n), 0, pmax)
        assert np.array_equal(o2[y:y + n, x:x + n].astype(np.int32), exp), f"block {j} reconstruction differs"
    assert (rns > 1).any()


def pad_any(img, margin=32):
    """A plane of any size inside `margin` zero samples: (flat buffer, stride, element offset of sample (0, 0))."""
    h, w = img.shape
    buf = np.zeros((h + 2 * margin, w + 2 * margin), dtype=img.dtype)
    buf[margin:margin + h, margin:margin + w] = img
    return np.ascontiguousarray(buf).reshape(-1), w + 2 * margin, margin * (w + 2 * margin) + margin


def sao_chroma_case(depth, width, height, seed):
    """Cb / Cr source and deblocked-like planes of a 4:2:0 picture + per-CTU parameters (both planes share the type)."""
    rng = np.random.default_rng([23, depth, width, seed])
    cw, ch = width // 2, height // 2
    clip = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0]
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    lim = 7 if depth == 8 else 31
    src, rec, params = [], [], []
    typ = rng.integers(-1, 5, size=nctu)
    for c in (1, 2):
        y = clip[c][:ch, :cw]
        sm = (y.astype(np.int32) * 2 + np.roll(y, 1, axis=1) + np.roll(y, 1, axis=0) + 2) >> 2
        r = np.clip(sm + (rng.integers(-3, 4, size=y.shape) << (depth - 8)), 0, (1 << depth) - 1).astype(y.dtype)
        p = np.zeros((nctu, 7), np.int32)
        p[:, 0] = typ
        p[:, 1] = rng.integers(0, 32, size=nctu)
        p[:, 2:6] = rng.integers(-lim, lim + 1, size=(nctu, 4))
        src.append(np.ascontiguousarray(y)); rec.append(np.ascontiguousarray(r)); params.append(p)
    return src, rec, params


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 200, 152), (10, 192, 136)])
def test_sao_chroma_restatement_equals_reference_class(depth, width, height):
    """The same SAO passes on the chroma planes of a 4:2:0 picture (32x32 CTU footprint, plane_offset 2) against the real class:
    calcSaoStatsCTU(addr, 1 / 2) and generateChromaOffsets."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_sao_chroma"):
        pytest.skip("oracle/_ref predates x265ref_sao_chroma")
    src, rec, params = sao_chroma_case(depth, width, height, 7)
    cw, ch = width // 2, height // 2
    nctu = params[0].shape[0]
    rcnt = [np.zeros((nctu, 5, 32), np.int32) for _ in range(2)]
    roff = [np.zeros((nctu, 5, 32), np.int32) for _ in range(2)]
    rout = [r.copy() for r in rec]
    arr = lambda xs: (ctypes.c_void_p * 2)(*[x.ctypes.data for x in xs])
    lib.x265ref_sao_chroma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.x265ref_sao_chroma(arr(src), arr(rout), width, height, arr(params), arr(rcnt), arr(roff)) == 0
    for c in range(2):
        fp, st, og = pad_any(src[c])
        rp = pad_any(rec[c])[0]
        cnt, off = O.sao_stats(depth, fp, rp, st, og, cw, ch, ctu=(32, 32), plane_offset=2)
        assert np.array_equal(cnt, rcnt[c]), f"plane {c + 1}: count differs in CTU/type {np.argwhere((cnt != rcnt[c]).any(axis=2))[:6].tolist()}"
        assert np.array_equal(off, roff[c]), f"plane {c + 1}: offsetOrg differs"
        out = O.sao_apply(depth, rp, st, og, cw, ch, params[c], ctu=(32, 32))
        got = out.reshape(-1, st)[32:32 + ch, 32:32 + cw]
        assert np.array_equal(got, rout[c]), f"plane {c + 1}: {np.count_nonzero(got != rout[c])} samples differ"
        assert (got != rec[c]).any() and cnt[:, :4, :5].sum() > 0


@pytest.mark.parametrize("depth,level,qp", [(8, 0, 26), (8, 2, 32), (10, 1, 40), (12, 1, 52)])
def test_chroma_inter_tu_round_trip_equals_reference_quant_class(depth, level, qp):
    """The chroma flavour of the inter TU stage (x265oracle_inter_recon_chroma, half-size blocks incl. the 4x4 DCT) with zero motion
    against the real Quant class: levels, numSig and reconstruction block by block."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_tu_roundtrip"):
        pytest.skip("oracle/_ref predates ref_quant.cpp")
    clip = F.synth_clip(128, 64, 2, depth=depth, seed=83)
    w64, h64 = 128, 64
    cw, ch, margin = w64 // 2, h64 // 2, 16
    dt = clip[0][0].dtype
    mk = lambda src: np.ascontiguousarray(np.pad(src[:ch, :cw], margin, mode="edge")).reshape(-1)
    cur, ref = mk(clip[1][1]), mk(clip[0][1])
    stride, org = cw + 2 * margin, margin * (cw + 2 * margin) + margin
    nctu = (w64 // 64) * (h64 // 64)
    nc = 4 << level
    nblk = (32 // nc) ** 2
    mv = np.zeros((nctu * 85, 2), np.int32)
    rec, lev, ns, _ = O.inter_recon_chroma(depth, cur, ref, stride, org, w64, h64, level, mv, qp)
    c2, r2, o2 = cur.reshape(-1, stride), ref.reshape(-1, stride), rec.reshape(-1, stride)
    resi, pos = [], []
    for ctu in range(nctu):
        for z in range(nblk):
            bx = sum(((z >> (2 * b)) & 1) << b for b in range(3))
            by = sum(((z >> (2 * b + 1)) & 1) << b for b in range(3))
            y, x = margin + (ctu // (w64 // 64)) * 32 + by * nc, margin + (ctu % (w64 // 64)) * 32 + bx * nc
            resi.append(c2[y:y + nc, x:x + nc].astype(np.int16) - r2[y:y + nc, x:x + nc].astype(np.int16))
            pos.append((y, x))
    resi = np.ascontiguousarray(np.stack(resi).reshape(-1))
    nj = len(pos)
    rlev, rns, rout = np.zeros(nj * nc * nc, np.int16), np.zeros(nj, np.uint32), np.zeros(nj * nc * nc, np.int16)
    lib.x265ref_tu_roundtrip.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
    assert lib.x265ref_tu_roundtrip(resi.ctypes.data, nc, qp, 0, 0, nj, rlev.ctypes.data, rns.ctypes.data, rout.ctypes.data) == 0
    assert np.array_equal(ns.reshape(-1), rns) and np.array_equal(lev, rlev)
    pmax = (1 << depth) - 1
    for j, (y, x) in enumerate(pos):
        exp = np.clip(r2[y:y + nc, x:x + nc].astype(np.int32) + rout[j * nc * nc:(j + 1) * nc * nc].reshape(nc, nc), 0, pmax)
        assert np.array_equal(o2[y:y + nc, x:x + nc].astype(np.int32), exp), f"block {j} reconstruction differs"
    assert (rns > 0).any()


@pytest.mark.parametrize("depth", [8, 10])
def test_search_driver_with_extra_candidates_equals_reference(depth):
    """motionEstimate's mvc[] candidates (motion.cpp:800-812: measured with SAD + mv cost against the predictor's cost, skipping
    zero / predictor / current-best duplicates), up to the 12 the encoder passes (search.cpp:2094)."""
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_motion_estimate_mvc"):
        pytest.skip("oracle/_ref predates the mvc entry point")
    lib.x265ref_motion_estimate_mvc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_int] * 8 + \
                                                [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle.so"))
    f = getattr(orc, f"x265oracle_motion_estimate_mvc_d{depth}")
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + \
                 [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    width, height = 256, 192
    clip = F.synth_clip(width, height, 2, depth=depth, seed=33)
    cur, stride, org, _, _ = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    es = cur.itemsize
    rng = np.random.default_rng([9, depth])
    qp = {8: 24, 10: 12, 12: 0}[depth]
    cq, qoff = F.qpel_cost_table(57, qmax=8 * 64 + 300)
    for method in (1, 2, 3):
        for subme in (1, 2, 5):
            n = 48
            ja = random_me_jobs(rng, n, width, height)
            jb = copy_jobs(ja)
            num = rng.integers(0, 13, size=n).astype(np.int32)
            mvc = rng.integers(-60, 61, size=(n, 12, 2)).astype(np.int32)
            mvc[rng.integers(0, n, size=8), 0] = 0                                      # some zero / duplicate candidates
            for i in range(0, n, 5):
                mvc[i, 1] = (ja[i].qmvpx, ja[i].qmvpy)
            lib.x265ref_motion_estimate_mvc(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, method, subme, 16, qp,
                                            -57, -57, 57, 57, ja, n, mvc.ctypes.data, num.ctypes.data)
            assert f(cur.ctypes.data + org * es, ref.ctypes.data + org * es, stride, method, subme, 16, cq.ctypes.data, qoff,
                     -57, -57, 57, 57, jb, n, 1, mvc.ctypes.data, num.ctypes.data) == 0
            for a, b in zip(ja, jb):
                assert (a.out_cost, a.out_qmvx, a.out_qmvy) == (b.out_cost, b.out_qmvx, b.out_qmvy)


# ------------------------------------------------------------------------------------------------ sign-bit hiding (the x265 default)
def test_scan_orders_equal_reference_tables():
    """The oracle generates the up-right diagonal / horizontal / vertical coefficient scans from the standard's definition; the
    reference ships them as tables (g_scanOrder, constants.cpp)."""
    import oracle_api as O
    lib = _ref(8)
    tab = (ctypes.c_void_p * 12).in_dll(lib, "_ZN4x26511g_scanOrderE")
    L = O.lib()
    for t in range(3):
        for s in range(4):
            n = 16 << (2 * s)
            want = np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(tab[t * 4 + s]))
            got = np.zeros(n, np.uint16)
            L.x265oracle_scan_order_d8.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            L.x265oracle_scan_order_d8(t, s + 2, got.ctypes.data)
            assert np.array_equal(got, want), f"scan type {t}, log2 size {s + 2}"


@pytest.mark.parametrize("chroma", [False, True])
@pytest.mark.parametrize("depth,n,qp,islice", [(8, 4, 24, 1), (8, 8, 27, 0), (8, 16, 30, 1), (8, 32, 22, 0), (8, 32, 38, 1), (10, 8, 36, 1), (10, 16, 40, 0), (12, 8, 50, 1)])
def test_intra_tu_sign_hiding_equals_reference_quant_class(depth, n, qp, islice, chroma):
    """pps.bSignHideEnabled = 1 (x265's default): Quant::transformNxN runs signBitHidingHDQ (quant.cpp:247-395, 471-476) after the
    quantiser.  Flat neighbours make every mode's prediction a constant, so DC (diagonal scan), mode 26 (horizontal scan for 4x4 / luma
    8x8) and mode 10 (vertical scan) can each be compared block by block with the real class."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_tu_roundtrip_ex"):
        pytest.skip("oracle/_ref predates x265ref_tu_roundtrip_ex")
    rng = np.random.default_rng([35, depth, n, qp])
    dt = np.uint8 if depth == 8 else np.uint16
    pmax, v = (1 << depth) - 1, 1 << (depth - 1)
    ntu = 48
    W = n * ntu
    yy, xx = np.mgrid[0:n, 0:W]
    amp = np.repeat(rng.choice([0.0, 0.03, 0.2, 0.45], size=ntu), n)[None, :]
    src = np.clip(np.rint(v + amp * pmax * np.sin(xx / 2.3 + yy / 1.7) + rng.normal(0, 2.5 * (1 << (depth - 8)), (n, W))), 0, pmax).astype(dt)
    nbw = 4 * n + 1
    nb = np.full(2 * nbw, v, dtype=dt)
    changed = 0
    for mode in (1, 26, 10):
        jobs = np.zeros(ntu, dtype=np.dtype([("off", "<i8", 4), ("arg", "<i4", 4)]))
        for t in range(ntu):
            jobs["off"][t] = (t * n, 0, nbw, t * n * n)
            jobs["arg"][t, 0] = mode
        rec, lev, ns, _ = O.intra_recon(depth, n, src.reshape(-1), W, nb, ntu * n * n, n, qp, islice | O.TU_SIGN_HIDE, jobs, chroma=chroma)
        plain = O.intra_recon(depth, n, src.reshape(-1), W, nb, ntu * n * n, n, qp, islice, jobs, chroma=chroma)[1]
        resi = np.stack([src[:, t * n:(t + 1) * n].astype(np.int16) - v for t in range(ntu)]).reshape(-1)
        rlev, rns, rout = np.zeros(ntu * n * n, np.int16), np.zeros(ntu, np.uint32), np.zeros(ntu * n * n, np.int16)
        fn = lib.x265ref_tu_roundtrip_ex
        fn.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p] * 3
        assert fn(resi.ctypes.data, n, qp, 1, islice, 1, mode, int(chroma), ntu, rlev.ctypes.data, rns.ctypes.data, rout.ctypes.data) == 0
        assert np.array_equal(lev, rlev), f"mode {mode}: levels differ in TUs {np.unique(np.nonzero(lev != rlev)[0] // (n * n))[:8]}"
        assert np.array_equal(ns, rns)
        assert np.array_equal(rec.astype(np.int32), np.clip(v + rout.astype(np.int32), 0, pmax))
        changed += int(np.count_nonzero(plain != lev))
    assert changed > 0, "sign hiding never changed a level: the case does not exercise it"


@pytest.mark.parametrize("depth,level,qp", [(8, 2, 30), (8, 0, 24), (8, 1, 27), (10, 1, 38), (12, 1, 50)])
def test_inter_tu_sign_hiding_equals_reference_quant_class(depth, level, qp):
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_tu_roundtrip_ex"):
        pytest.skip("oracle/_ref predates x265ref_tu_roundtrip_ex")
    clip = F.synth_clip(128, 64, 2, depth=depth, seed=83)
    cur, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref = F.pad_plane(clip[0][0])[0]
    nctu = (w64 // 64) * (h64 // 64)
    n = 8 << level
    nblk = (64 // n) ** 2
    mv = np.zeros((nctu * 85, 2), np.int32)
    rec, lev, ns, dist = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv, qp, intra_slice=O.TU_SIGN_HIDE)
    plain = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv, qp)[1]
    r0, c0 = org // stride, org % stride
    blocks = []
    for ctu in range(nctu):
        cx, cy = (ctu % (w64 // 64)) * 64, (ctu // (w64 // 64)) * 64
        for z in range(nblk):
            bx, by = _zxy(z)
            y0, x0 = r0 + cy + by * n, c0 + cx + bx * n
            blocks.append((cur[y0:y0 + n, x0:x0 + n].astype(np.int16) - ref[y0:y0 + n, x0:x0 + n].astype(np.int16)).reshape(-1))
    resi = np.concatenate(blocks)
    nj = len(blocks)
    rlev, rns, rout = np.zeros(nj * n * n, np.int16), np.zeros(nj, np.uint32), np.zeros(nj * n * n, np.int16)
    fn = lib.x265ref_tu_roundtrip_ex
    fn.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p] * 3
    assert fn(resi.ctypes.data, n, qp, 0, 0, 1, 1, 0, nj, rlev.ctypes.data, rns.ctypes.data, rout.ctypes.data) == 0
    assert np.array_equal(lev, rlev) and np.array_equal(ns, rns)
    assert np.count_nonzero(plain != lev) > 0


@pytest.mark.parametrize("depth,n,qp,intra,chroma,lists,nr", [(8, 8, 27, 0, False, True, False), (8, 16, 30, 0, False, True, True), (8, 32, 24, 1, False, True, False),
                                                             (8, 4, 26, 1, False, True, True), (8, 8, 33, 0, True, True, True), (10, 16, 38, 0, False, True, True),
                                                             (8, 32, 30, 0, False, False, True), (12, 8, 50, 1, False, True, False)])
def test_tu_scaling_lists_and_denoiser_equal_reference_quant_class(depth, n, qp, intra, chroma, lists, nr):
    """The optional tables of the TU stages: the HEVC default scaling lists (quantiser coefficients, dequant_scaling) and the denoiser
    (denoiseDct before the quantiser, running residual sums) - the oracle's intra TU stage with flat neighbours (constant prediction)
    against the real Quant::transformNxN / invtransformNxN, sign hiding on."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_tu_roundtrip_ex2"):
        pytest.skip("oracle/_ref predates x265ref_tu_roundtrip_ex2")
    rng = np.random.default_rng([37, depth, n, qp])
    dt = np.uint8 if depth == 8 else np.uint16
    pmax, v = (1 << depth) - 1, 1 << (depth - 1)
    ntu = 40
    W = n * ntu
    yy, xx = np.mgrid[0:n, 0:W]
    amp = np.repeat(rng.choice([0.0, 0.03, 0.2, 0.45], size=ntu), n)[None, :]
    src = np.clip(np.rint(v + amp * pmax * np.sin(xx / 2.3 + yy / 1.7) + rng.normal(0, 2.5 * (1 << (depth - 8)), (n, W))), 0, pmax).astype(dt)
    resi = np.stack([src[:, t * n:(t + 1) * n].astype(np.int16) - v for t in range(ntu)]).reshape(-1)
    nr_off = rng.integers(0, 6 << (depth - 8), size=n * n).astype(np.uint16) if nr else None
    rlev, rns, rout = np.zeros(ntu * n * n, np.int16), np.zeros(ntu, np.uint32), np.zeros(ntu * n * n, np.int16)
    qc, dqc, rsum = np.zeros(n * n, np.int32), np.zeros(n * n, np.int32), np.zeros(n * n, np.uint32)
    fn = lib.x265ref_tu_roundtrip_ex2
    fn.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
    if intra:
        mode, flags = 1, 1 | O.TU_SIGN_HIDE
    else:
        mode, flags = 1, O.TU_SIGN_HIDE
    assert fn(resi.ctypes.data, n, qp, intra, intra, 1, mode, int(chroma), int(lists), None if nr_off is None else nr_off.ctypes.data, ntu,
              rlev.ctypes.data, rns.ctypes.data, rout.ctypes.data, qc.ctypes.data, dqc.ctypes.data, rsum.ctypes.data) == 0
    if lists:
        assert len(np.unique(qc)) > 1 or n == 4            # the default lists are not flat (4x4 is)
    osum = np.zeros(n * n, np.uint32)
    O.set_tu_tables(depth, qc if lists else None, dqc if lists else None, nr_off, osum if nr else None)
    try:
        if intra:
            nbw = 4 * n + 1
            nb = np.full(2 * nbw, v, dtype=dt)
            jobs = np.zeros(ntu, dtype=np.dtype([("off", "<i8", 4), ("arg", "<i4", 4)]))
            for t in range(ntu):
                jobs["off"][t] = (t * n, 0, nbw, t * n * n)
                jobs["arg"][t, 0] = mode
            rec, lev, ns, _ = O.intra_recon(depth, n, src.reshape(-1), W, nb, ntu * n * n, n, qp, flags, jobs, chroma=chroma)
            assert np.array_equal(rec.astype(np.int32), np.clip(v + rout.astype(np.int32), 0, pmax))
        else:
            # inter flavour through the intra entry is not possible (list type differs): use the luma / chroma inter stage on a synthetic picture
            # whose reference is flat, so the residual of every block is the block itself minus v
            nblk = ntu
            level = {8: 0, 16: 1, 32: 2}[n] if not chroma else {4: 0, 8: 1, 16: 2}[n]
            pytest_blocks = (64 // (8 << level)) ** 2
            nctu = (nblk + pytest_blocks - 1) // pytest_blocks
            w64, h64 = 64 * nctu, 64
            y = np.full((h64, w64), v, dt)
            if not chroma:
                bs = 8 << level
                k = 0
                for ctu in range(nctu):
                    for z in range(pytest_blocks):
                        if k >= ntu:
                            break
                        bx, by = _zxy(z)
                        y[by * bs:(by + 1) * bs, ctu * 64 + bx * bs:ctu * 64 + (bx + 1) * bs] = src[:, k * n:(k + 1) * n]
                        k += 1
                cur, stride, org, _, _ = F.pad_plane(y)
                ref = F.pad_plane(np.full((h64, w64), v, dt))[0]
                mv = np.zeros((nctu * 85, 2), np.int32)
                _, lev, ns, _ = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv, qp, intra_slice=flags)
            else:
                cs = 4 << level
                c = np.full((h64 // 2, w64 // 2), v, dt)
                k = 0
                for ctu in range(nctu):
                    for z in range(pytest_blocks):
                        if k >= ntu:
                            break
                        bx, by = _zxy(z)
                        c[by * cs:(by + 1) * cs, ctu * 32 + bx * cs:ctu * 32 + (bx + 1) * cs] = src[:, k * n:(k + 1) * n]
                        k += 1
                cp, sc, oc = F.pad_chroma(c, w64, h64)
                rp = F.pad_chroma(np.full_like(c, v), w64, h64)[0]
                mv = np.zeros((nctu * 85, 2), np.int32)
                _, lev, ns, _ = O.inter_recon_chroma(depth, cp.reshape(-1), rp.reshape(-1), sc, oc, w64, h64, level, mv, qp, intra_slice=flags)
            lev, ns = lev[:ntu * n * n], ns[:ntu]
        assert np.array_equal(lev, rlev), f"levels differ in TUs {np.unique(np.nonzero(lev != rlev)[0] // (n * n))[:8]}"
        assert np.array_equal(ns, rns)
        if nr:
            assert np.array_equal(osum, rsum) and rsum.sum() > 0
    finally:
        O.set_tu_tables(depth)


@pytest.mark.parametrize("depth,level,slice_b,use_wp,use_wbp,wl0,wl1", [
    (8, 2, 0, 0, 0, None, None),                                                                   # plain P picture
    (8, 1, 0, 1, 0, [(1, 45, 6, 6), (1, 70, -9, 5), (1, 60, 3, 6)], None),                          # weighted P picture
    (8, 0, 1, 0, 0, None, None),                                                                   # B picture, addAvg
    (8, 2, 1, 0, 1, [(1, 45, 6, 6), (1, 70, -9, 6), (0, 64, 0, 6)], [(1, 70, -9, 6), (0, 64, 0, 6), (1, 50, 10, 6)]),   # weighted B
    (10, 1, 1, 0, 1, [(0, 64, 0, 6), (0, 64, 0, 6), (0, 64, 0, 6)], [(1, 127, -128, 7), (1, 3, 1, 0), (1, -20, 100, 4)]),
    (10, 2, 0, 1, 0, [(1, 120, 20, 7), (0, 64, 0, 6), (1, 1, 0, 0)], None),
    (12, 1, 1, 0, 1, [(1, -20, 100, 4), (1, 90, 7, 3), (1, 33, -4, 5)], [(1, 90, 7, 3), (1, 61, 4, 6), (0, 1, 0, 0)])])
def test_inter_stage_predictions_equal_the_real_motion_compensation(depth, level, slice_b, use_wp, use_wbp, wl0, wl1, seed=0):
    """The PREDICTION half of the oracle's inter TU stages - uni- and bi-directional, luma and both 4:2:0 chroma planes, explicit
    weights per list and plane (addWeightUni / addWeightBi), every fractional phase - against the real Predict::motionCompensation
    driven CU by CU (oracle/ref_predict.cpp): P slices with / without pps.bUseWeightPred, B slices with / without bUseWeightedBiPred."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_motion_compensation"):
        pytest.skip("oracle/_ref predates x265ref_motion_compensation")
    rng = np.random.default_rng([61, depth, level, slice_b, seed])
    width, height = 256, 128
    clip = F.synth_clip(width, height, 3, depth=depth, seed=75 + level + seed)
    luma = [F.pad_plane(c[0]) for c in clip]
    stride, org, w64, h64 = luma[0][1:5]
    def edge_pad(img, margin=24):                          # (flat plane, stride, org) with replicated edges, like extendPicBorder
        buf = np.ascontiguousarray(np.pad(img, margin, mode="edge"))
        return buf.reshape(-1), img.shape[1] + 2 * margin, margin * (img.shape[1] + 2 * margin) + margin
    chroma = [[edge_pad(np.ascontiguousarray(c[k])) for k in (1, 2)] for c in clip]
    nctu = (w64 // 64) * (h64 // 64)
    mvs = []
    for _ in range(2):
        qx, qy = rng.integers(-40, 41, size=nctu * 85), rng.integers(-40, 41, size=nctu * 85)
        qx[::4] &= ~7; qy[::3] &= ~7; qx[1::5] &= ~3
        m = np.zeros((nctu * 85, 2), np.int32)
        m[:, 1] = (qx & 0xffff) | (qy << 16)
        mvs.append(m)
    nblk = (64 >> (3 + level)) ** 2
    dirs = rng.integers(1, 4, size=nctu * nblk).astype(np.uint8) if slice_b else np.ones(nctu * nblk, np.uint8)
    none3 = [(0, 1, 0, 0)] * 3
    wtab = np.asarray([wl0 or none3, wl1 or none3], dtype=np.int32)
    dt = clip[0][0].dtype
    ry, rcb, rcr = np.zeros((height, width), dt), np.zeros((height // 2, width // 2), dt), np.zeros((height // 2, width // 2), dt)
    compact = lambda a: np.ascontiguousarray(a)
    lib.x265ref_motion_compensation.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
    c0, c1 = [compact(clip[0][k]) for k in (1, 2)], [compact(clip[2][k]) for k in (1, 2)]
    assert lib.x265ref_motion_compensation(luma[0][0].ctypes.data, c0[0].ctypes.data, c0[1].ctypes.data, luma[2][0].ctypes.data, c1[0].ctypes.data, c1[1].ctypes.data,
                                           width, height, level, mvs[0].ctypes.data, mvs[1].ctypes.data, dirs.ctypes.data, slice_b, use_wp, use_wbp,
                                           wtab.ctypes.data, ry.ctypes.data, rcb.ctypes.data, rcr.ctypes.data) == 0
    # the oracle's view of the same picture: a list has a weight table iff the slice type's flag is on
    have = (bool(use_wp), False) if not slice_b else (bool(use_wbp), bool(use_wbp))
    def wplane(p):             # the reference decides on the LUMA entry's wtPresent for every plane (pwp0->wtPresent, predict.cpp:94, :187, :196)
        return tuple((int(wtab[l][0][0]),) + tuple(int(v) for v in wtab[l][p][1:]) if have[l] else None for l in range(2))
    cur = luma[1][0]
    py = np.zeros((height, width), dt)
    with O.pred_capture(depth, py):
        O.inter_recon_bi(depth, cur.reshape(-1), stride, org, luma[0][0].reshape(-1), luma[2][0].reshape(-1), w64, h64, level, mvs[0], mvs[1], 30 + 6 * (depth - 8),
                         dir_flags=dirs, weights=wplane(0))
    assert np.array_equal(py, ry), f"luma: {np.count_nonzero(py != ry)} predicted samples differ from Predict::motionCompensation"
    for k, rc in ((0, rcb), (1, rcr)):
        pc = np.zeros((height // 2, width // 2), dt)
        cst, corg = chroma[1][k][1], chroma[1][k][2]
        with O.pred_capture(depth, pc):
            O.inter_recon_chroma_bi(depth, chroma[1][k][0], chroma[0][k][0], chroma[2][k][0], cst, corg, w64, h64, level, mvs[0], mvs[1], 30 + 6 * (depth - 8),
                                    dir_flags=dirs, weights=wplane(1 + k))
        assert np.array_equal(pc, rc), f"chroma plane {k}: {np.count_nonzero(pc != rc)} predicted samples differ"
    if not slice_b and not use_wp:                      # the uni-directional stages (x265hip_inter_recon / _chroma) predict the same picture
        py2 = np.zeros((height, width), dt)
        with O.pred_capture(depth, py2):
            O.inter_recon(depth, cur.reshape(-1), stride, org, luma[0][0].reshape(-1), stride, org, w64, h64, level, mvs[0], 30 + 6 * (depth - 8))
        assert np.array_equal(py2, ry)
        pc2 = np.zeros((height // 2, width // 2), dt)
        with O.pred_capture(depth, pc2):
            O.inter_recon_chroma(depth, chroma[1][0][0], chroma[0][0][0], chroma[1][0][1], chroma[1][0][2], w64, h64, level, mvs[0], 30 + 6 * (depth - 8))
        assert np.array_equal(pc2, rcb)


@pytest.mark.parametrize("depth,n,chroma", [(8, 4, False), (8, 8, False), (8, 16, False), (8, 32, False), (8, 4, True), (8, 16, True), (10, 8, False), (10, 32, False),
                                            (10, 8, True), (12, 16, False)])
def test_intra_stage_predictions_equal_the_real_predict_class(depth, n, chroma):
    """The PREDICTION half of the intra TU stage - all 35 modes, the choice between the unfiltered and the filtered neighbours
    (g_intraFilterFlags & size), edge smoothing up to 16x16, the chroma flavour's unfiltered / unsmoothed form - against the real
    Predict::predIntraLumaAng / predIntraChromaAng (oracle/ref_predict.cpp) on random neighbour arrays."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_pred_intra"):
        pytest.skip("oracle/_ref predates x265ref_pred_intra")
    rng = np.random.default_rng([37, depth, n, int(chroma)])
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    nbw = 4 * n + 1
    sets = 3
    nb = rng.integers(0, pmax + 1, size=2 * nbw * sets).astype(dt)          # per set: unfiltered then "filtered" (any values: the choice is what is tested)
    njobs = 35 * sets
    src = np.zeros((n, n * njobs), dt)
    jobs = np.zeros(njobs, dtype=np.dtype([("off", "<i8", 4), ("arg", "<i4", 4)]))
    for t in range(njobs):
        s_, mode = divmod(t, 35)
        jobs["off"][t] = (t * n, 2 * nbw * s_, 2 * nbw * s_ + nbw, t * n * n)
        jobs["arg"][t, 0] = mode
    pred = np.zeros((njobs * n, n), dt)
    with O.pred_capture(depth, pred):
        O.intra_recon(depth, n, src.reshape(-1), n * njobs, nb, njobs * n * n, n, 30 + 6 * (depth - 8), 0, jobs, chroma=chroma)
    lib.x265ref_pred_intra.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    log2n = int(np.log2(n))
    for t in range(njobs):
        s_, mode = divmod(t, 35)
        unf, fil = nb[2 * nbw * s_:2 * nbw * s_ + nbw], nb[2 * nbw * s_ + nbw:2 * nbw * (s_ + 1)]
        want = np.zeros((n, n), dt)
        assert lib.x265ref_pred_intra(mode, log2n, np.ascontiguousarray(unf).ctypes.data, np.ascontiguousarray(fil).ctypes.data, int(chroma), want.ctypes.data) == 0
        got = pred[t * n:(t + 1) * n]
        assert np.array_equal(got, want), f"mode {mode} (neighbour set {s_}): {np.count_nonzero(got != want)} predicted samples differ"


def sao_rdo_case(depth, width, height, seed, noise):
    """Source / deblocked-like Y, Cb, Cr of a 4:2:0 picture in the PicYuv layout; `noise` steers how much SAO can gain."""
    rng = np.random.default_rng([29, depth, width, seed])
    clip = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0]
    planes_src, planes_rec = [], []
    for c in range(3):
        y = clip[c]
        sm = (y.astype(np.int32) * 2 + np.roll(y, 1, axis=1) + np.roll(y, 1, axis=0) + 2) >> 2
        r = np.clip(sm + (rng.integers(-noise, noise + 1, size=y.shape) << (depth - 8)), 0, (1 << depth) - 1).astype(y.dtype)
        h, w = y.shape
        r[: h // 4, : w // 4] = y[: h // 4, : w // 4]                 # zero differences: SAO off is the best choice there
        r[h // 2:, w // 2:] = np.clip(y[h // 2:, w // 2:].astype(np.int32) + (3 << (depth - 8)), 0, (1 << depth) - 1)      # a constant shift: band offsets
        planes_src.append(np.ascontiguousarray(y)); planes_rec.append(np.ascontiguousarray(r))
    return planes_src, planes_rec


@pytest.mark.parametrize("depth,width,height,slice_type,qp,csp400", [(8, 256, 192, 1, 27, 0), (8, 200, 150, 2, 22, 0), (8, 320, 128, 0, 37, 0), (10, 192, 136, 1, 30, 0),
                                                                      (12, 192, 136, 2, 14, 0), (8, 256, 128, 1, 32, 1), (10, 128, 128, 0, 24, 1),
                                                                      (8, 2560, 192, 1, 27, 0), (8, 1920, 1080, 0, 30, 0)])
def test_sao_rdo_restatement_equals_the_real_class(depth, width, height, slice_type, qp, csp400):
    """oracle/x265_oracle_pipeline6.c against the real SAO::rdoSaoUnitCu (oracle/ref_sao.cpp::x265ref_sao_rdo) driven row by row like
    FrameFilter does: type, band position, offsets and merge mode of every CTU and plane, and the count of CTUs left without SAO; B / P /
    I slices (the context initialisation differs), per-CTU QPs, 4:2:0 and 4:0:0."""
    import oracle_api as O
    lib = _ref(depth)
    if not hasattr(lib, "x265ref_sao_rdo"):
        pytest.skip("oracle/_ref predates x265ref_sao_rdo")
    rng = np.random.default_rng([31, depth, width, qp])
    src, rec = sao_rdo_case(depth, width, height, 11, 2 + qp // 12)
    fenc, stride, org, w64, h64 = F.pad_plane(src[0])
    recp = F.pad_plane(rec[0])[0]
    ctus_w, ctus_h = w64 // 64, h64 // 64
    nctu = ctus_w * ctus_h
    planes = 1 if csp400 else 3
    bufs_src, bufs_rec = [fenc], [recp]
    cgeo = None
    if not csp400:
        for c in (1, 2):
            b, st, og = F.pad_chroma(src[c], w64, h64)
            bufs_src.append(b); bufs_rec.append(F.pad_chroma(rec[c], w64, h64)[0])
            cgeo = (st, og)
    ctu_qp = np.clip(qp + rng.integers(-3, 4, size=nctu), 0, 51).astype(np.int32)
    rparams = [np.zeros((nctu, 7), np.int32) for _ in range(3)]
    info = np.zeros(8, np.int64)
    P3 = ctypes.c_void_p * 3
    lib.x265ref_sao_rdo.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    pad3 = lambda xs: P3(*([x.ctypes.data for x in xs] + [None] * (3 - len(xs))))
    assert lib.x265ref_sao_rdo(pad3(bufs_src), pad3(bufs_rec), width, height, csp400, slice_type, qp, ctu_qp.ctypes.data, 0, P3(*[p.ctypes.data for p in rparams]),
                               info.ctypes.data) == 0
    # the host's tables: per-state bit costs and the lambdas of every CTU's QP
    lib.x265ref_entropy_bits_table.restype = ctypes.POINTER(ctypes.c_uint32)
    bits = np.ctypeslib.as_array(lib.x265ref_entropy_bits_table(), shape=(128,)).copy()
    lam2 = (ctypes.c_double * 70).in_dll(lib, "_ZN4x26516x265_lambda2_tabE")
    cscale = (ctypes.c_uint8 * 70).in_dll(lib, "_ZN4x26513g_chromaScaleE")
    lam = np.zeros((nctu, 2), np.int64)
    for a in range(nctu):
        q = int(ctu_qp[a])
        qc = q if csp400 else int(cscale[min(max(q, 0), 69)])
        lam[a] = (int(np.floor(256.0 * lam2[q])), int(np.floor(256.0 * lam2[qc])))
    assert (lam[0] == info[0:2]).all()
    counts, orgs = [], []
    c0, o0 = O.sao_stats(depth, fenc, recp, stride, org, width, height)
    counts.append(c0); orgs.append(o0)
    if not csp400:
        for c in (1, 2):
            cc, oo = O.sao_stats(depth, bufs_src[c], bufs_rec[c], cgeo[0], cgeo[1], width // 2, height // 2, ctu=(32, 32), plane_offset=2)
            counts.append(cc); orgs.append(oo)
    params, nos = O.sao_rdo(depth, counts, orgs, ctus_w, ctus_h, lam, int(info[2]), int(info[3]), bits, sao_flag=(1, 0 if csp400 else 1))
    for pl in range(planes):
        bad = np.argwhere((params[pl] != rparams[pl]).any(axis=1))[:5].reshape(-1).tolist()
        assert np.array_equal(params[pl], rparams[pl]), f"plane {pl}: CTUs {bad}: oracle {params[pl][bad].tolist()} reference {rparams[pl][bad].tolist()}"
    assert int(nos[0]) == int(info[4]) and (csp400 or int(nos[1]) == int(info[5]))
    # the case set exercises every outcome
    t = rparams[0][:, 0]
    assert (t >= 0).any(), "no CTU took SAO"
    if nctu >= 6:
        assert (rparams[0][:, 6] > 0).any(), "no CTU took a merge candidate"
