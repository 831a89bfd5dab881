"""GPU parity: the lookahead P-frame cost estimate (x265hip_lowres_cost) vs the oracle's restatement of
CostEstimateGroup::estimateFrameCost / estimateCUCost (slicetype.cpp:3189-3388), which tests/test_oracle_me_vs_reference.py pins
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


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 208, 144), (10, 192, 128), (8, 1920, 1080), (10, 640, 360), (8, 48, 32)])
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
    assert np.array_equal(st.frame.cpu().numpy(), frame)
    if width >= 192:
        assert (mvs != 0).any() and ((lcost >> 14) == 0).any() and ((lcost >> 14) == 1).any()


def test_lowres_cost_rejects_bad_geometry():
    import torch
    H = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    dev = torch.device("cuda:0")
    t = torch.zeros(4096, dtype=torch.uint8, device=dev)
    i32 = torch.zeros(4096, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError):
        # 600 blocks per row cannot be walked by the 16 rows in flight of a 2-row picture
        H.lowres_cost(8, 64, 600, 2, i32, 0, [H.lowres_cost_pair(8, 0, t, [t] * 4, i32, i32, i32, i32, i32, i32)])
