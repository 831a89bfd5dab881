"""GPU parity: lookahead picture preparation (x265hip_lowres_init) and intra cost estimate (x265hip_lowres_intra) vs the
oracle's restatement of Lowres::init (lowres.cpp:294-306) and LookaheadTLD::lowresIntraEstimate (slicetype.cpp:696-772)."""
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


def _run(width, height, depth, seed, extreme=None, penalty=5):
    import torch
    dev = torch.device("cuda:0")
    y = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0][0]
    if extreme == "max":
        y = np.full_like(y, (1 << depth) - 1)
    elif extreme == "noise":
        y = np.random.default_rng(seed).integers(0, 1 << depth, size=y.shape).astype(y.dtype)
    pic = P.DevicePicture(y, dev)
    la = S.Lookahead(width, height, depth, dev, intra_penalty=penalty)
    la.run(pic)
    torch.cuda.synchronize()
    O = _oracle()
    rows = la.lines + 2 * la.my
    planes = O.lowres_init(depth, pic.host, pic.stride, pic.org, la.stride, la.org, rows, la.width, la.lines, la.mx, la.my)
    for i in range(4):
        got = la.planes[i].cpu().numpy().view(pic.host.dtype)
        assert np.array_equal(got, planes[i]), f"lowres plane {i} differs ({np.count_nonzero(got != planes[i])} samples)"
    cost, mode, lc = O.lowres_intra(depth, planes[0], la.stride, la.org, la.wcu, la.hcu, penalty)
    assert np.array_equal(la.intra_cost.cpu().numpy(), cost), "intraCost differs"
    assert np.array_equal(la.intra_mode.cpu().numpy(), mode), "intraMode differs"
    assert np.array_equal(la.lowres_costs.cpu().numpy().view(np.uint16), lc), "lowresCosts differs"
    return cost, mode


@pytest.mark.parametrize("depth", [8, 10])
def test_lookahead_small(depth):
    cost, mode = _run(256, 128, depth, seed=3)
    assert len(set(mode.tolist())) > 4          # the scan really picks different modes on textured content


def test_lookahead_non_multiple_size():
    _run(200, 136, 8, seed=4)                   # lowres 100x68 -> 13x9 blocks, reads the source margin


def test_lookahead_extremes():
    _run(128, 64, 8, seed=5, extreme="max")
    _run(128, 64, 10, seed=5, extreme="noise")
    _run(128, 64, 8, seed=6, extreme="noise", penalty=0)


def test_lookahead_1080p():
    _run(1920, 1080, 8, seed=7)
