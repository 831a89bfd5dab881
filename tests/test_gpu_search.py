"""GPU parity: the motion search drivers (x265hip_me_search) vs the oracle's restatement of
MotionEstimate::motionEstimate (oracle/x265_oracle_search.c), which tests/test_oracle_classes_vs_reference.py pins against the
real reference class - so GPU == oracle == x265 for predictor start, DIA / HEX / STAR / FULL and the sub-pel refinement."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")

ALL_PU_DIMS = [(8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 8), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (16, 12),
               (12, 16), (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32), (64, 48), (48, 64), (64, 16), (16, 64)]
METHODS = {"dia": A.ME_DIA, "hex": A.ME_HEX, "umh": A.ME_UMH, "star": A.ME_STAR, "full": A.ME_FULL}


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _jobs(rng, n, width, height, zero_mvp=False):
    jobs = np.zeros(n, dtype=A.me_search_job_dtype())
    for j in jobs:
        w, h = ALL_PU_DIMS[int(rng.integers(0, len(ALL_PU_DIMS)))]
        j["px"], j["py"] = int(rng.integers(0, (width - w) // 4 + 1)) * 4, int(rng.integers(0, (height - h) // 4 + 1)) * 4
        j["w"], j["h"] = w, h
        if not zero_mvp and rng.integers(0, 4):
            j["qmvpx"], j["qmvpy"] = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
    return jobs


def _check(depth, method, width, height, seed, njobs, submes, meranges, extreme=None):
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    y0, y1 = clip[0][0], clip[1][0]
    if extreme == "flat":
        y0 = np.zeros_like(y0); y1 = np.full_like(y1, (1 << depth) - 1)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(y0, dev)
    rng = np.random.default_rng([seed, depth, METHODS[method]])
    for subme in submes:
        for merange in meranges:
            bound = merange if method == "full" else 57
            cq, qoff = F.qpel_cost_table(bound, qmax=8 * 64 + 300)
            cq_d = torch.from_numpy(cq.view(np.int16)).to(dev)
            mn, mx = (-bound, -bound), (bound, bound)
            if rng.integers(0, 3) == 0:
                mn = (-int(rng.integers(3, 20)), -int(rng.integers(3, 20)))
                mx = (int(rng.integers(3, 20)), int(rng.integers(3, 20)))
            jobs = _jobs(rng, njobs, width, height)
            exp = O.motion_estimate(depth, cur.host, ref.host, cur.stride, cur.org, METHODS[method], subme, merange, cq, qoff, mn, mx, jobs)
            jd = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
            A.me_search(depth, cur.t, cur.stride, cur.org, ref.t, ref.stride, ref.org, METHODS[method], subme, merange, cq_d, qoff, mn, mx, jd, njobs)
            torch.cuda.synchronize()
            got = jd.cpu().numpy().view(A.me_search_job_dtype())
            for f in ("out_cost", "out_qmvx", "out_qmvy"):
                bad = np.nonzero(got[f] != exp[f])[0]
                assert bad.size == 0, (f"{method} depth {depth} subme {subme} merange {merange} bounds {mn}..{mx}: {f} differs for {bad.size} of {njobs} "
                                       f"jobs, first {jobs[bad[0]]}: got {got[bad[0]]} expected {exp[bad[0]]}")


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("method", ["dia", "hex", "umh", "star"])
def test_pattern_search_all_partitions(depth, method):
    _check(depth, method, 256, 192, seed=41, njobs=64, submes=(0, 2, 3, 5, 7), meranges=(4, 16, 57))


@pytest.mark.parametrize("depth", [8, 10])
def test_full_search_driver(depth):
    _check(depth, "full", 192, 128, seed=42, njobs=32, submes=(1, 3), meranges=(6,))


def test_search_extremes():
    _check(8, "hex", 128, 128, seed=43, njobs=32, submes=(2,), meranges=(16,), extreme="flat")
    _check(8, "star", 128, 128, seed=43, njobs=32, submes=(3,), meranges=(57,), extreme="flat")


SEA_UNSUPPORTED = {(8, 4), (4, 8), (32, 8), (8, 32)}


@pytest.mark.parametrize("depth", [8, 10])
def test_sea_block_sum_planes(depth):
    """x265hip_sea_integral: each of the twelve planes holds, at (x, y), the sum of the bw x bh block of the padded reference whose
    corner is (x, y) - what the reference's integral_init primitives leave there (the oracle's SEA restatement, pinned against
    the real class fed with planes from those primitives, takes exactly these sums)."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(200, 136, 1, depth=depth, seed=48)
    ref = P.DevicePicture(clip[0][0], dev)
    planes, org = A.sea_integral(depth, ref.t, ref.stride, ref.org, ref.w64, ref.h64, F.MARGIN_X, F.MARGIN_Y)
    torch.cuda.synchronize()
    host = ref.host.astype(np.int64)
    rows = host.shape[0]
    cs = np.zeros((rows + 1, ref.stride + 1), np.int64)
    cs[1:, 1:] = host.cumsum(0).cumsum(1)
    got = planes.cpu().numpy().view(np.uint32).reshape(12, rows, ref.stride)
    for k, (bw, bh) in enumerate(A.SEA_PLANE_DIMS):
        exp = cs[bh:, bw:] - cs[:-bh, bw:] - cs[bh:, :-bw] + cs[:-bh, :-bw]          # rows - bh + 1 x stride - bw + 1 corners
        vw = ref.w64 + 2 * F.MARGIN_X - bw + 1
        assert np.array_equal(got[k, :rows - bh + 1, :vw], exp[:, :vw]), f"plane {k} ({bw}x{bh}) differs"
        assert not got[k, rows - bh + 1:, :].any()                                       # nothing written outside the valid corners


def _check_sea(depth, width, height, seed, njobs, submes, meranges):
    """X265_SEA through the device planes against the oracle restatement; returns (PU sizes seen, jobs whose mv moved)."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    integral = A.sea_integral(depth, ref.t, ref.stride, ref.org, ref.w64, ref.h64, F.MARGIN_X, F.MARGIN_Y)
    rng = np.random.default_rng([seed, depth])
    cq, qoff = F.qpel_cost_table(57, qmax=8 * 64 + 300)
    cq_d = torch.from_numpy(cq.view(np.int16)).to(dev)
    seen, moved = set(), 0
    for subme in submes:
        for merange in meranges:
            mn, mx = (-44, -44), (44, 44)
            if rng.integers(0, 3) == 0:
                mn = (-int(rng.integers(3, 20)), -int(rng.integers(3, 20)))
                mx = (int(rng.integers(3, 20)), int(rng.integers(3, 20)))
            jobs = _jobs(rng, njobs, width, height)
            ok = np.array([(int(j["w"]), int(j["h"])) not in SEA_UNSUPPORTED for j in jobs])
            exp = O.motion_estimate(depth, cur.host, ref.host, cur.stride, cur.org, A.ME_SEA, subme, merange, cq, qoff, mn, mx, jobs[ok])
            jd = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
            A.me_search(depth, cur.t, cur.stride, cur.org, ref.t, ref.stride, ref.org, A.ME_SEA, subme, merange, cq_d, qoff, mn, mx, jd, len(jobs),
                        integral=integral)
            torch.cuda.synchronize()
            got = jd.cpu().numpy().view(A.me_search_job_dtype())
            assert (got["out_cost"][~ok] == -1).all()
            for f in ("out_cost", "out_qmvx", "out_qmvy"):
                bad = np.nonzero(got[f][ok] != exp[f])[0]
                assert bad.size == 0, (f"sea depth {depth} subme {subme} merange {merange} bounds {mn}..{mx}: {f} differs for {bad.size} jobs, "
                                       f"first {jobs[ok][bad[0]]}: got {got[ok][bad[0]]} expected {exp[bad[0]]}")
            seen |= {(int(j["w"]), int(j["h"])) for j in jobs[ok]}
            moved += int(np.count_nonzero(exp["out_qmvx"] | exp["out_qmvy"]))
    return seen, moved


@pytest.mark.parametrize("depth", [8, 10])
def test_sea_search_driver(depth):
    """X265_SEA through the device planes: every supported PU size, random predictors / bounds / merange / sub-pel levels against
    the oracle restatement (pinned against the real MotionEstimate with real integral planes); the four sizes whose DC terms the
    reference reads from outside the PU come back refused (out_cost -1)."""
    seen, moved = _check_sea(depth, 256, 192, 49, 96, (0, 2, 3, 5, 7), (5, 16, 40))
    assert len(seen) == len(ALL_PU_DIMS) - len(SEA_UNSUPPORTED) and moved > 500


def test_unimplemented_methods_are_rejected():
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(64, 64, 2, depth=8, seed=1)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    cq, qoff = F.qpel_cost_table(8, qmax=400)
    cq_d = torch.from_numpy(cq.view(np.int16)).to(dev)
    jd = torch.zeros(36, dtype=torch.uint8, device=dev)
    for m in (A.ME_SEA, 6):
        with pytest.raises(A.X265HipError):
            A.me_search(8, cur.t, cur.stride, cur.org, ref.t, ref.stride, ref.org, m, 2, 16, cq_d, qoff, (-8, -8), (8, 8), jd, 1)


@pytest.mark.parametrize("depth", [8, 10])
def test_search_with_extra_candidates(depth):
    """mvc[] candidates of motionEstimate (motion.cpp:800-812), 0..12 per job."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    width, height = 256, 192
    clip = F.synth_clip(width, height, 2, depth=depth, seed=44)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    rng = np.random.default_rng([44, depth])
    cq, qoff = F.qpel_cost_table(57, qmax=8 * 64 + 300)
    cq_d = torch.from_numpy(cq.view(np.int16)).to(dev)
    for method in ("hex", "umh", "star"):
        n = 64
        jobs = _jobs(rng, n, width, height)
        num = rng.integers(0, 13, size=n).astype(np.int32)
        mvc = rng.integers(-60, 61, size=(n, 12, 2)).astype(np.int32)
        mvc[::7, 0] = 0
        exp = O.motion_estimate(depth, cur.host, ref.host, cur.stride, cur.org, METHODS[method], 3, 16, cq, qoff, (-57, -57), (57, 57), jobs,
                                mvc=mvc, num_mvc=num)
        jd = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
        A.me_search(depth, cur.t, cur.stride, cur.org, ref.t, ref.stride, ref.org, METHODS[method], 3, 16, cq_d, qoff, (-57, -57), (57, 57), jd, n,
                    mvc=torch.from_numpy(mvc).to(dev), num_mvc=torch.from_numpy(num).to(dev))
        torch.cuda.synchronize()
        got = jd.cpu().numpy().view(A.me_search_job_dtype())
        for f in ("out_cost", "out_qmvx", "out_qmvy"):
            assert np.array_equal(got[f], exp[f]), f"{method}: {f} differs with extra candidates"


def test_stream_release_pools_the_scratch_of_a_stream_the_host_destroys():
    """x265hip_stream_release (round-4 advisor, low): the search driver keeps a scratch buffer per (device, stream); a host that creates and destroys streams hands
    each one back before destroying it - its buffer goes to the pool the next stream adopts from (nothing is freed: a captured graph might hold it), results on the
    next stream are still exact, a second release of the same stream finds nothing."""
    import ctypes
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    L = A.lib()
    L.x265hip_stream_release.argtypes = [ctypes.c_void_p]
    clip = F.synth_clip(256, 192, 2, depth=8, seed=77)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    rng = np.random.default_rng(5)
    cq, qoff = F.qpel_cost_table(57, qmax=8 * 64 + 300)
    cq_d = torch.from_numpy(cq.view(np.int16)).to(dev)
    for round_ in range(3):
        s = torch.cuda.Stream()
        jobs = _jobs(rng, 48, 256, 192)
        exp = O.motion_estimate(8, cur.host, ref.host, cur.stride, cur.org, METHODS["hex"], 2, 16, cq, qoff, (-57, -57), (57, 57), jobs)
        jd = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
        torch.cuda.synchronize()
        A.me_search(8, cur.t, cur.stride, cur.org, ref.t, ref.stride, ref.org, METHODS["hex"], 2, 16, cq_d, qoff, (-57, -57), (57, 57), jd, 48, stream=s.cuda_stream)
        s.synchronize()
        got = jd.cpu().numpy().view(A.me_search_job_dtype())
        for f in ("out_cost", "out_qmvx", "out_qmvy"):
            assert np.array_equal(got[f], exp[f]), (round_, f)
        assert L.x265hip_stream_release(s.cuda_stream) >= 1          # the stream's scratch went back to the pool
        assert L.x265hip_stream_release(s.cuda_stream) == 0          # nothing left under this stream
        del s
