"""GPU parity: the lookahead P-frame cost estimate (x265hip_lowres_cost) vs the oracle's restatement of
CostEstimateGroup::estimateFrameCost / estimateCUCost (slicetype.cpp:3189-3388), which tests/test_oracle_classes_vs_reference.py pins
against the real reference classes."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _pair(width, height, depth, seed):
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    rng = np.random.default_rng([17, depth, width, seed])
    y0 = clip[0][0]
    y1 = np.roll(y0, (3, -5), axis=(0, 1)).copy()
    y1[: height // 3] = clip[1][0][: height // 3]
    y1[-32:, -64:] = y0[-32:, -64:]                      # zero-residual blocks
    noise = rng.integers(-2, 3, size=y1.shape) << (depth - 8)
    y1[:, : width // 2] = np.clip(y1[:, : width // 2].astype(np.int32) + noise[:, : width // 2], 0, (1 << depth) - 1).astype(y1.dtype)
    return y1, y0


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 208, 144), (10, 192, 128), (8, 1920, 1080), (10, 640, 360), (8, 48, 32), (12, 208, 144)])
def test_lowres_cost_matches_oracle(depth, width, height):
    import torch
    dev = torch.device("cuda:0")
    y1, y0 = _pair(width, height, depth, 91)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(y0, dev)
    lc, lr = S.Lookahead(width, height, depth, dev, intra_penalty=5 if depth == 8 else 80), S.Lookahead(width, height, depth, dev)
    lc.run(cur)
    lr.run(ref)
    st = S.LookaheadCost(lc, dev)
    st.run(lc, lr)
    torch.cuda.synchronize()
    O = _oracle()
    dt = y1.dtype
    cp = lc.planes[0].cpu().numpy().view(dt)
    rp = [p.cpu().numpy().view(dt) for p in lr.planes]
    cq = st.cost_q.cpu().numpy().view(np.uint16)
    mvs, mvc, lcost, rows, frame = O.lowres_cost(depth, cp, rp, lc.stride, lc.org, lc.wcu, lc.hcu, cq, st.qoff, lc.intra_cost.cpu().numpy())
    gm = st.mvs.cpu().numpy().reshape(-1, 2)
    bad = np.flatnonzero((gm != mvs).any(axis=1))
    assert bad.size == 0, f"{bad.size} mvs differ, first blocks {bad[:6]} (of {lc.wcu} x {lc.hcu}): gpu {gm[bad[:3]]} oracle {mvs[bad[:3]]}"
    assert np.array_equal(st.mv_costs.cpu().numpy(), mvc)
    assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), lcost)
    assert np.array_equal(st.row_satds.cpu().numpy(), rows)
    assert np.array_equal(st.frame.cpu().numpy()[:3], frame) and st.frame.cpu().numpy()[3] == frame[0]
    if width >= 192:
        assert (mvs != 0).any() and ((lcost >> 14) == 0).any() and ((lcost >> 14) == 1).any()


@pytest.mark.parametrize("depth,width,height,bias", [(8, 256, 128, 0), (8, 208, 144, 20), (10, 192, 128, 0), (8, 1280, 720, 0), (12, 192, 128, 0)])
def test_lowres_b_cost_matches_oracle(depth, width, height, bias):
    """B pictures: two lists, the skip shortcut, the two bi-directional candidates and the scaled score; then a second estimate
    that reuses list 0 (bDoSearch[0] == false)."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(width, height, 3, depth=depth, seed=92)
    rng = np.random.default_rng([18, depth, width])
    y0, y2 = clip[0][0], clip[2][0]
    y1 = np.roll(y0, (2, -3), axis=(0, 1)).copy()
    y1[: height // 3] = np.roll(y2, (-1, 2), axis=(0, 1))[: height // 3]
    y1[-32:, -64:] = y0[-32:, -64:]
    y1[-64:-32, :64] = ((y0[-64:-32, :64].astype(np.int32) + y2[-64:-32, :64] + 1) >> 1).astype(y1.dtype)
    noise = rng.integers(-1, 2, size=y1.shape) << (depth - 8)
    y1[:, width // 2:] = np.clip(y1[:, width // 2:].astype(np.int32) + noise[:, width // 2:], 0, (1 << depth) - 1).astype(y1.dtype)
    pics = [P.DevicePicture(y, dev) for y in (y1, y0, y2)]
    las = [S.Lookahead(width, height, depth, dev, intra_penalty=5 if depth == 8 else 80) for _ in range(3)]
    for la, pic in zip(las, pics):
        la.run(pic)
    lc, l0, l1 = las
    st = S.LookaheadCost(lc, dev, bidir=True)
    st.run(lc, l0, l1, bframe_bias=bias)
    torch.cuda.synchronize()
    O = _oracle()
    dt = y1.dtype
    cp = lc.planes[0].cpu().numpy().view(dt)
    r0 = [p.cpu().numpy().view(dt) for p in l0.planes]
    r1 = [p.cpu().numpy().view(dt) for p in l1.planes]
    cq = st.cost_q.cpu().numpy().view(np.uint16)
    ic = lc.intra_cost.cpu().numpy()
    mvs, mvc, lcost, rows, frame = O.lowres_cost(depth, cp, r0, lc.stride, lc.org, lc.wcu, lc.hcu, cq, st.qoff, ic, ref1_planes=r1, bframe_bias=bias)
    g0, g1 = st.mvs.cpu().numpy().reshape(-1, 2), st.mvs1.cpu().numpy().reshape(-1, 2)
    assert np.array_equal(g0, mvs[0]) and np.array_equal(g1, mvs[1]), "mvs differ"
    assert np.array_equal(st.mv_costs.cpu().numpy(), mvc[0]) and np.array_equal(st.mv_costs1.cpu().numpy(), mvc[1])
    assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), lcost)
    assert np.array_equal(st.row_satds.cpu().numpy(), rows)
    assert np.array_equal(st.frame.cpu().numpy(), frame)
    used = lcost >> 14
    assert (used == 1).any() and (used == 2).any() and (used == 3).any()
    # second estimate of the same picture against another list-1 reference: list 0 is not searched again
    keep0, keepc0 = g0.copy(), st.mv_costs.cpu().numpy().copy()
    st.run(lc, l0, l0, do_search=(0, 1), bframe_bias=bias)
    torch.cuda.synchronize()
    mvs2, mvc2, lcost2, rows2, frame2 = O.lowres_cost(depth, cp, r0, lc.stride, lc.org, lc.wcu, lc.hcu, cq, st.qoff, ic, ref1_planes=r0,
                                                      bframe_bias=bias, do_search=(0, 1), mvs_in=(keep0, None), mv_costs_in=(keepc0, None))
    assert np.array_equal(st.mvs.cpu().numpy().reshape(-1, 2), keep0) and np.array_equal(st.mvs1.cpu().numpy().reshape(-1, 2), mvs2[1])
    assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), lcost2) and np.array_equal(st.frame.cpu().numpy(), frame2)
    # third estimate, NEITHER list searched again (a third of the slice-type decision's triples): the dependency-free launch (round 6) - the vectors are
    # inputs, everything else must equal the oracle's answer and what the lock-step walk gives for the same request
    keep1, keepc1 = st.mvs1.cpu().numpy().reshape(-1, 2).copy(), st.mv_costs1.cpu().numpy().copy()
    st.lowres_costs.zero_(); st.row_satds.zero_(); st.frame.zero_()
    st.run(lc, l0, l0, do_search=(0, 0), bframe_bias=bias)
    torch.cuda.synchronize()
    mvs3, mvc3, lcost3, rows3, frame3 = O.lowres_cost(depth, cp, r0, lc.stride, lc.org, lc.wcu, lc.hcu, cq, st.qoff, ic, ref1_planes=r0, bframe_bias=bias,
                                                      do_search=(0, 0), mvs_in=(keep0, keep1), mv_costs_in=(keepc0, keepc1))
    flat = (st.lowres_costs.cpu().numpy().view(np.uint16).copy(), st.row_satds.cpu().numpy().copy(), st.frame.cpu().numpy().copy())
    assert np.array_equal(st.mvs.cpu().numpy().reshape(-1, 2), keep0) and np.array_equal(st.mvs1.cpu().numpy().reshape(-1, 2), keep1)
    assert np.array_equal(flat[0], lcost3) and np.array_equal(flat[1], rows3) and np.array_equal(flat[2], frame3)
    os.environ["X265HIP_LOWRES_COST_FLAT_OFF"] = "1"
    try:
        st.lowres_costs.zero_(); st.row_satds.zero_(); st.frame.zero_()
        st.run(lc, l0, l0, do_search=(0, 0), bframe_bias=bias)
        torch.cuda.synchronize()
    finally:
        del os.environ["X265HIP_LOWRES_COST_FLAT_OFF"]
    assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), flat[0]) and np.array_equal(st.row_satds.cpu().numpy(), flat[1]) and np.array_equal(st.frame.cpu().numpy(), flat[2])


def test_lowres_cost_rejects_bad_geometry():
    import torch
    H = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    dev = torch.device("cuda:0")
    t = torch.zeros(4096, dtype=torch.uint8, device=dev)
    i32 = torch.zeros(4096, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError):
        # 600 blocks per row cannot be walked by the 16 rows in flight of a 2-row picture
        H.lowres_cost(8, 64, 600, 2, i32, 0, [H.lowres_cost_pair(8, 0, t, [t] * 4, i32, i32, i32, i32, i32, i32)])


@pytest.mark.parametrize("depth,width,height,bidir", [(8, 208, 144, False), (8, 640, 360, True), (10, 1280, 720, False), (8, 1920, 1080, True)])
def test_lowres_cost_split_into_bands_equals_one_workgroup(monkeypatch, depth, width, height, bidir):
    """One estimate alone is walked by several workgroups (bands of block rows, mvs of a band's top row handed up through L2): every band
    count must give the integers of the one-workgroup walk, which the tests above hold against the oracle (and at 720p / 1080p those tests
    run the split form themselves: it is the default from 32 block rows up)."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(width, height, 3, depth=depth, seed=93)
    y1 = np.roll(clip[0][0], (2, -3), axis=(0, 1)).copy()
    y1[: height // 3] = clip[1][0][: height // 3]
    pics = [P.DevicePicture(y, dev) for y in (y1, clip[0][0], clip[2][0])]
    las = [S.Lookahead(width, height, depth, dev, intra_penalty=5 if depth == 8 else 80) for _ in range(3)]
    for la, pic in zip(las, pics):
        la.run(pic)
    lc, l0, l1 = las

    refresh = importlib.import_module("x265-yuuki-asuna_amd.hipabi").lib().x265hip_lowres_cost_env_refresh      # the switches are read when the library loads

    def run(bands, side_by_side=True):
        monkeypatch.setenv("X265HIP_LOWRES_COST_SPLIT", str(bands))
        if side_by_side:
            monkeypatch.delenv("X265HIP_LOWRES_COST_SO_OFF", raising=False)
        else:
            monkeypatch.setenv("X265HIP_LOWRES_COST_SO_OFF", "1")
        refresh()
        st = S.LookaheadCost(lc, dev, bidir=bidir)
        if bidir:
            st.run(lc, l0, l1, bframe_bias=10)
        else:
            st.run(lc, l0)
        torch.cuda.synchronize()
        out = [st.mvs, st.mv_costs, st.lowres_costs, st.row_satds, st.frame] + ([st.mvs1, st.mv_costs1] if bidir else [])
        return [o.cpu().numpy().copy() for o in out]

    try:
        one = run(1)
        assert (one[0] != 0).any()
        for bands in (2, 3, 5, 16):
            got = run(bands)
            for k, (a, b) in enumerate(zip(one, got)):
                assert np.array_equal(a, b), f"{bands} bands: output {k} differs from the one-workgroup walk"
            if bidir:
                # round 6: a split B estimate walks its two lists side by side and a flat launch finishes it (the default above); the one-walk split form stays as the A/B
                got = run(bands, side_by_side=False)
                for k, (a, b) in enumerate(zip(one, got)):
                    assert np.array_equal(a, b), f"{bands} bands, lists in one walk: output {k} differs from the one-workgroup walk"
    finally:
        monkeypatch.undo()
        refresh()
