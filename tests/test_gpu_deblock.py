"""GPU parity: in-loop luma deblocking (x265hip_deblock_bs_inter + x265hip_deblock_luma) vs the oracle's restatement of
Deblock::getBoundaryStrength / edgeFilterLuma (oracle/x265_oracle_pipeline4.c)."""
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


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("level", [0, 1, 2])
def test_deblock_after_reconstruction(depth, level):
    """The real use: boundary strengths from the sub-pel / reconstruction stages' outputs, then the filter, on a
    reconstruction produced by the pipeline itself (so strong, normal and no-filter decisions all occur)."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    qp = 32 + 12 * (depth == 10)
    clip = F.synth_clip(256, 192, 2, depth=depth, seed=51 + level)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    pipe = S.FramePipeline(cur.w64, cur.h64, depth, dev, rng=8, subme=2, level=level, qp=qp, want_surf=False)
    rec = pipe.run(cur, ref)
    torch.cuda.synchronize()
    rec_h = rec.cpu().numpy().view(cur.host.dtype).copy()
    mv_h = pipe.sp.out.cpu().numpy()
    ns_h = pipe.rc.num_sig.cpu().numpy().view(np.uint32)
    db = S.Deblock(cur.w64, cur.h64, depth, level, qp, dev)
    db.run(rec, cur, pipe.sp.out, pipe.rc.num_sig)
    torch.cuda.synchronize()
    bv, bh = O.deblock_bs_inter(depth, cur.w64, cur.h64, level, mv_h, ns_h)
    assert np.array_equal(db.bs_ver.cpu().numpy(), bv) and np.array_equal(db.bs_hor.cpu().numpy(), bh)
    assert bv.any() and bh.any()
    exp = O.deblock_luma(depth, rec_h, cur.stride, cur.org, cur.w64, cur.h64, bv, bh, qp)
    got = rec.cpu().numpy().view(cur.host.dtype)
    assert np.array_equal(got, exp), f"{np.count_nonzero(got != exp)} samples differ"
    assert np.count_nonzero(exp != rec_h) > 100          # the filter really changed the picture


@pytest.mark.parametrize("depth", [8, 10])
def test_deblock_random_strengths_qp_map_and_offsets(depth):
    """Arbitrary Bs maps (0 / 1 / 2), a per-block QP map and non-zero beta / tc offsets on noisy and smooth content."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    rng = np.random.default_rng([61, depth])
    w, h = 192, 128
    maxv = (1 << depth) - 1
    for kind in ("smooth", "noise", "steps"):
        if kind == "noise":
            img = rng.integers(0, maxv + 1, size=(h, w))
        elif kind == "smooth":
            yy, xx = np.mgrid[0:h, 0:w]
            img = (xx + yy) * maxv // (w + h) + rng.integers(-2, 3, size=(h, w))
        else:
            img = (np.indices((h, w)).sum(axis=0) // 8 % 2) * (maxv // 6) + maxv // 3 + rng.integers(-1, 2, size=(h, w))
        img = np.clip(img, 0, maxv).astype(np.uint8 if depth == 8 else np.uint16)
        pic = P.DevicePicture(img, dev)
        bv = rng.integers(0, 3, size=(pic.h64 // 4) * (pic.w64 // 8)).astype(np.uint8)
        bh = rng.integers(0, 3, size=(pic.h64 // 8) * (pic.w64 // 4)).astype(np.uint8)
        bv.reshape(pic.h64 // 4, pic.w64 // 8)[:, 0] = 0
        bh.reshape(pic.h64 // 8, pic.w64 // 4)[0, :] = 0
        qmap = rng.integers(20, 45, size=(pic.h64 // 8) * (pic.w64 // 8)).astype(np.int8)
        for bo, to in ((0, 0), (2, -1), (-3, 3)):
            plane = pic.t.clone()
            A.deblock_luma(depth, plane, pic.stride, pic.org, pic.w64, pic.h64, torch.from_numpy(bv).to(dev), torch.from_numpy(bh).to(dev),
                           30, qp_map=torch.from_numpy(qmap).to(dev), beta_offset_div2=bo, tc_offset_div2=to)
            torch.cuda.synchronize()
            exp = O.deblock_luma(depth, pic.host, pic.stride, pic.org, pic.w64, pic.h64, bv, bh, 30, qp_map=qmap, beta_offset_div2=bo, tc_offset_div2=to)
            got = plane.cpu().numpy().view(pic.host.dtype)
            assert np.array_equal(got, exp), f"{kind} offsets {(bo, to)}: {np.count_nonzero(got != exp)} samples differ"
