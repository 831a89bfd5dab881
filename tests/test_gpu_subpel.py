"""GPU parity: sub-pel refinement stage (x265hip_subpel_refine) vs the oracle's restatement of
motion.cpp:1448-1664 driven through the oracle's luma_hpp/vpp/hvpp + sad/satd primitives."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("depth,subme", [(8, 2), (8, 0), (8, 1), (8, 3), (8, 5), (8, 7), (10, 2), (10, 7), (12, 3)])
def test_subpel_refine_matches_oracle(depth, subme, planes):
    """planes: the candidates are read from the reference picture's phase planes (x265hip_phase_planes) instead of being interpolated
    per candidate tile - the same decisions either way."""
    import torch
    dev = torch.device("cuda:0")
    # sub-pel motion: frame 1 is frame 0 shifted by a non-integer amount (bilinear mix) plus noise
    rng = np.random.default_rng([41, depth, subme])
    clip = F.synth_clip(256, 128, 2, depth=depth, seed=40 + subme)
    y0 = clip[0][0].astype(np.float32)
    sh = np.roll(y0, (1, 2), axis=(0, 1))
    y1 = np.clip(np.rint(0.6 * y0 + 0.4 * sh + rng.normal(0, 1.0, size=y0.shape)), 0, (1 << depth) - 1).astype(clip[0][0].dtype)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, 8, depth, dev, want_surf=False)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, subme, dev, phase_planes=planes)
    sp.run(cur, ref)
    torch.cuda.synchronize()
    O = _oracle()
    best = ms.best.cpu().numpy().view(np.uint64)
    exp = O.subpel_refine(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, 8,
                          0, ms.nctu, best, sp.cost_q_host, sp.qoff, subme)
    got = sp.out.cpu().numpy().reshape(-1, 2)
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {got.shape[0]} PUs differ, first {bad[:5]}: {got[bad[:3]]} vs {exp[bad[:3]]}"
    # the refinement must actually move some vectors off the integer grid
    frac = (exp[:, 1] & 3) | ((exp[:, 1] >> 16) & 3)
    assert np.count_nonzero(frac) > exp.shape[0] // 10
