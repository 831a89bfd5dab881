"""GPU parity: the lookahead's weighted-reference analysis (x265hip_lowres_weight_cost / x265hip_lowres_weight_apply + the host-side
float guess of stages.WeightAnalysis) vs the oracle's restatement of LookaheadTLD::weightCostLuma / weightsAnalyse
(oracle/x265_oracle_pipeline3.c), which tests/test_oracle_classes_vs_reference.py pins against the real LookaheadTLD."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _fade(depth, width, height, gain, lift, seed):
    y0 = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0][0]
    pmax = (1 << depth) - 1
    noise = np.random.default_rng(seed).integers(-1, 2, size=y0.shape) * (1 << (depth - 8))
    y1 = np.clip(np.rint(y0.astype(np.float64) * gain + lift * (1 << (depth - 8))) + noise, 0, pmax).astype(y0.dtype)
    return y0, y1


def _stats(plane, la):
    a = plane.reshape(-1, la.stride)[la.my:la.my + la.lines, la.mx:la.mx + la.width].astype(np.int64)
    sm = int(a.sum())
    return int((a * a).sum()) - sm * sm // a.size, sm


@pytest.mark.parametrize("depth,width,height,gain,lift", [(8, 512, 256, 0.75, 6), (8, 416, 288, 1.0, 0), (8, 512, 256, 1.3, -20), (8, 384, 256, 0.5, 40),
                                                        (8, 512, 288, 1.0, 9), (10, 384, 256, 0.8, 12), (10, 512, 256, 1.15, -6), (8, 384, 256, 0.25, 150)])
def test_weight_analysis_matches_oracle(depth, width, height, gain, lift, seed=95, check_expectation=True):
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    y0, y1 = _fade(depth, width, height, gain, lift, seed=seed)
    cur_pic, ref_pic = P.DevicePicture(y1, dev), P.DevicePicture(y0, dev)
    cur, ref = S.Lookahead(width, height, depth, dev, intra_penalty=5 if depth == 8 else 80), S.Lookahead(width, height, depth, dev)
    cur.run(cur_pic); ref.run(ref_pic)
    torch.cuda.synchronize()
    dt = y0.dtype
    cpl = [p.cpu().numpy().view(dt) for p in cur.planes]
    rpl = [p.cpu().numpy().view(dt) for p in ref.planes]
    icost = cur.intra_cost.cpu().numpy()
    ssd_c, sum_c = _stats(cpl[0], cur)
    ssd_r, sum_r = _stats(rpl[0], ref)
    # the cost kernel alone, four candidates in one launch (unweighted, identity weight, two real ones)
    cands = [None, (64, 6, 0), (3, 2, 6), (83, 6, -19)]
    cost = torch.full((4,), -1, dtype=torch.int32, device=dev)
    A.lowres_weight_cost(depth, cur.planes[0], ref.planes[0], cur.stride, cur.org, cur.width, cur.lines, cur.intra_cost, cands, cost)
    torch.cuda.synchronize()
    exp = [O.lowres_weight_cost(depth, cpl[0], rpl[0], cur.stride, cur.org, cur.width, cur.lines, icost, c) for c in cands]
    assert cost.cpu().numpy().view(np.uint32).tolist() == exp
    assert exp[0] == exp[1]                                             # scale 64 / 2^6, offset 0 is the identity
    # the whole analysis
    wa = S.WeightAnalysis(cur, dev)
    got = wa.analyse(cur, ref, (ssd_c, ssd_r), (sum_c, sum_r))
    torch.cuda.synchronize()
    want = O.weights_analyse(depth, cpl[0], rpl[0], cur.stride, cur.org, cur.width, cur.lines, icost, (ssd_c, ssd_r), (sum_c, sum_r))
    assert got == want, f"analysis: device {got}, oracle {want}"
    if check_expectation:
        assert (want[0] is not None) == (not (gain == 1.0 and lift == 0))
    if want[0] is not None:
        for i in range(4):
            assert np.array_equal(wa.weighted[i].cpu().numpy().view(dt), O.weight_plane(depth, rpl[i], want[0])), f"weighted plane {i} differs"


@pytest.mark.parametrize("depth,width,height,gain,lift", [(8, 512, 256, 0.75, 6), (10, 384, 256, 1.2, -10)])
def test_weighted_p_frame_cost_matches_oracle(depth, width, height, gain, lift):
    """--weightp end to end on the device: analysis -> weighted planes -> the frame cost estimate searching them (the oracle chain
    is pinned against the real CostEstimateGroup::singleCost with bEnableWeightedPred)."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    y0, y1 = _fade(depth, width, height, gain, lift, seed=98)
    y1 = np.roll(y1, (2, -2), axis=(0, 1)).copy()
    cur, ref = S.Lookahead(width, height, depth, dev, intra_penalty=5 if depth == 8 else 80), S.Lookahead(width, height, depth, dev)
    cur.run(P.DevicePicture(y1, dev)); ref.run(P.DevicePicture(y0, dev))
    torch.cuda.synchronize()
    dt = y0.dtype
    cp = cur.planes[0].cpu().numpy().view(dt)
    rpl = [p.cpu().numpy().view(dt) for p in ref.planes]
    ssd_c, sum_c = _stats(cp, cur)
    ssd_r, sum_r = _stats(rpl[0], ref)
    wa = S.WeightAnalysis(cur, dev)
    weight, _, _ = wa.analyse(cur, ref, (ssd_c, ssd_r), (sum_c, sum_r))
    assert weight is not None
    st = S.LookaheadCost(cur, dev)
    st.run(cur, wa.weighted_ref)
    torch.cuda.synchronize()
    icost = cur.intra_cost.cpu().numpy()
    assert weight == O.weights_analyse(depth, cp, rpl[0], cur.stride, cur.org, cur.width, cur.lines, icost, (ssd_c, ssd_r), (sum_c, sum_r))[0]
    wpl = [O.weight_plane(depth, p, weight) for p in rpl]
    cq = st.cost_q.cpu().numpy().view(np.uint16)
    mvs, mvc, lcost, rows, frame = O.lowres_cost(depth, cp, wpl, cur.stride, cur.org, cur.wcu, cur.hcu, cq, st.qoff, icost)
    assert np.array_equal(st.mvs.cpu().numpy().reshape(-1, 2), mvs) and np.array_equal(st.mv_costs.cpu().numpy(), mvc)
    assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), lcost) and np.array_equal(st.row_satds.cpu().numpy(), rows)
    assert np.array_equal(st.frame.cpu().numpy()[:3], frame)
    plain = O.lowres_cost(depth, cp, rpl, cur.stride, cur.org, cur.wcu, cur.hcu, cq, st.qoff, icost)
    assert plain[4][0] != frame[0]


@pytest.mark.parametrize("depth,width,height,gain,lift", [(8, 512, 256, 0.8, 8), (10, 384, 256, 0.7, 12)])
def test_weighted_b_frame_cost_matches_oracle(depth, width, height, gain, lift):
    """--weightp on a B picture: weighted list-0 planes for the search, the unweighted ones for the bi-directional candidates."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    y0, y1 = _fade(depth, width, height, gain, lift, seed=104)
    y1 = np.roll(y1, (2, -2), axis=(0, 1)).copy()
    y2 = np.roll(y1, (-2, 4), axis=(0, 1)).copy()
    y1[-64:-32, :64] = ((y0[-64:-32, :64].astype(np.int32) + y2[-64:-32, :64] + 1) >> 1).astype(y1.dtype)
    cur, l0, l1 = (S.Lookahead(width, height, depth, dev, intra_penalty=5 if depth == 8 else 80) for _ in range(3))
    cur.run(P.DevicePicture(y1, dev)); l0.run(P.DevicePicture(y0, dev)); l1.run(P.DevicePicture(y2, dev))
    torch.cuda.synchronize()
    dt = y0.dtype
    cp = cur.planes[0].cpu().numpy().view(dt)
    p0 = [p.cpu().numpy().view(dt) for p in l0.planes]
    p1 = [p.cpu().numpy().view(dt) for p in l1.planes]
    ssd_c, sum_c = _stats(cp, cur)
    ssd_r, sum_r = _stats(p0[0], l0)
    wa = S.WeightAnalysis(cur, dev)
    weight, _, _ = wa.analyse(cur, l0, (ssd_c, ssd_r), (sum_c, sum_r))
    assert weight is not None
    st = S.LookaheadCost(cur, dev, bidir=True)
    st.run(cur, wa.weighted_ref, l1, ref_bi=l0)
    torch.cuda.synchronize()
    icost = cur.intra_cost.cpu().numpy()
    w0 = [O.weight_plane(depth, p, weight) for p in p0]
    cq = st.cost_q.cpu().numpy().view(np.uint16)
    mvs, mvc, lcost, rows, frame = O.lowres_cost(depth, cp, w0, cur.stride, cur.org, cur.wcu, cur.hcu, cq, st.qoff, icost, ref1_planes=p1, ref_bi_planes=p0)
    assert np.array_equal(st.mvs.cpu().numpy().reshape(-1, 2), mvs[0]) and np.array_equal(st.mvs1.cpu().numpy().reshape(-1, 2), mvs[1])
    assert np.array_equal(st.mv_costs.cpu().numpy(), mvc[0]) and np.array_equal(st.mv_costs1.cpu().numpy(), mvc[1])
    assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), lcost) and np.array_equal(st.row_satds.cpu().numpy(), rows)
    assert np.array_equal(st.frame.cpu().numpy(), frame)
    allw = O.lowres_cost(depth, cp, w0, cur.stride, cur.org, cur.wcu, cur.hcu, cq, st.qoff, icost, ref1_planes=p1)
    assert not np.array_equal(allw[2], lcost)                       # the unweighted planes of the bi-directional candidates matter here


def test_weight_cost_rejects_bad_candidates():
    import torch
    dev = torch.device("cuda:0")
    y0, y1 = _fade(8, 128, 128, 0.9, 3, seed=96)
    la = S.Lookahead(128, 128, 8, dev)
    la.run(P.DevicePicture(y1, dev))
    cost = torch.zeros(4, dtype=torch.int32, device=dev)
    for bad in ([(200, 6, 0)], [(64, 9, 0)], [(64, 6, 300)], []):
        with pytest.raises(A.X265HipError):
            A.lowres_weight_cost(8, la.planes[0], la.planes[0], la.stride, la.org, la.width, la.lines, la.intra_cost, bad, cost)
