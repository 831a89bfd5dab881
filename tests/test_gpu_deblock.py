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


@pytest.mark.parametrize("depth,level,qp,cq", [(8, 2, 32, (0, 0)), (8, 1, 36, (3, -4)), (8, 0, 30, (-2, 6)), (10, 1, 33, (1, 1)), (12, 2, 40, (0, 5))])
def test_deblock_with_intra_blocks_and_chroma(depth, level, qp, cq):
    """Mixed intra / inter 4:2:0 pictures: Bs 2 on intra CU edges (luma tc index + the only edges the chroma filter touches), Cb / Cr
    planes filtered on the 8-sample chroma grid with the chroma QP mapping; vs the oracle restatement pinned against the real class."""
    import torch
    H = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    dev = torch.device("cuda:0")
    O = _oracle()
    clip = F.synth_clip(256, 192, 2, depth=depth, seed=57 + level)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    pipe = S.FramePipeline(cur.w64, cur.h64, depth, dev, rng=8, subme=2, level=level, qp=qp + 6 * (depth - 8), want_surf=False)
    rec = pipe.run(cur, ref)
    torch.cuda.synchronize()
    rng = np.random.default_rng([43, depth, level])
    nctu, npu = pipe.ms.nctu, (64 >> (3 + level)) ** 2
    intra = (rng.random((nctu, npu)) < 0.3).astype(np.uint8)
    qmap = rng.integers(max(qp - 6, 0), min(qp + 7, 52), size=(cur.h64 // 8) * (cur.w64 // 8)).astype(np.int8)
    cw, ch = cur.w64 // 2, cur.h64 // 2
    dt = cur.host.dtype
    margin = 16
    cst, corg = cw + 2 * margin, margin * (cw + 2 * margin) + margin
    planes = []
    for c in (1, 2):
        pl = np.zeros((ch + 2 * margin, cst), dtype=dt)
        src = clip[1][c]
        steps = rng.integers(-6, 7, size=(ch // 8, cw // 8)) << (depth - 8)
        body = np.zeros((ch, cw), np.int32)
        body[:src.shape[0], :src.shape[1]] = src
        body[src.shape[0]:, :] = body[src.shape[0] - 1]
        body[:, src.shape[1]:] = body[:, src.shape[1] - 1:src.shape[1]]
        pl[margin:margin + ch, margin:margin + cw] = np.clip(body + np.kron(steps, np.ones((8, 8), np.int32)), 0, (1 << depth) - 1)
        planes.append(np.ascontiguousarray(pl).reshape(-1))
    mv_h = pipe.sp.out.cpu().numpy()
    ns_h = pipe.rc.num_sig.cpu().numpy().view(np.uint32)
    rec_h = rec.cpu().numpy().view(dt).copy()
    bv, bh = O.deblock_bs_inter(depth, cur.w64, cur.h64, level, mv_h, ns_h, intra=intra)
    exp_y = O.deblock_luma(depth, rec_h, cur.stride, cur.org, cur.w64, cur.h64, bv, bh, qp, qp_map=qmap, tc_offset_div2=1)
    exp_cb, exp_cr = O.deblock_chroma(depth, planes[0], planes[1], cst, corg, cur.w64, cur.h64, bv, bh, qp, qp_map=qmap, cb_qp_offset=cq[0],
                                      cr_qp_offset=cq[1], tc_offset_div2=1)
    d_bv = torch.zeros(bv.size, dtype=torch.uint8, device=dev)
    d_bh = torch.zeros(bh.size, dtype=torch.uint8, device=dev)
    H.deblock_bs_inter(cur.w64, cur.h64, level, pipe.sp.out, pipe.rc.num_sig, d_bv, d_bh, intra=torch.from_numpy(intra.reshape(-1)).to(dev))
    d_q = torch.from_numpy(qmap).to(dev)
    H.deblock_luma(depth, rec, cur.stride, cur.org, cur.w64, cur.h64, d_bv, d_bh, qp, qp_map=d_q, tc_offset_div2=1)
    d_cb = torch.from_numpy(planes[0].view(np.uint8)).to(dev)
    d_cr = torch.from_numpy(planes[1].view(np.uint8)).to(dev)
    H.deblock_chroma(depth, d_cb, d_cr, cst, corg, cur.w64, cur.h64, d_bv, d_bh, qp, qp_map=d_q, cb_qp_offset=cq[0], cr_qp_offset=cq[1], tc_offset_div2=1)
    torch.cuda.synchronize()
    assert np.array_equal(d_bv.cpu().numpy(), bv) and np.array_equal(d_bh.cpu().numpy(), bh) and (bv == 2).any() and (bh == 2).any()
    assert np.array_equal(rec.cpu().numpy().view(dt), exp_y), "luma differs"
    assert np.array_equal(d_cb.cpu().numpy().view(dt), exp_cb) and np.array_equal(d_cr.cpu().numpy().view(dt), exp_cr), "chroma differs"
    assert np.count_nonzero(exp_cb != planes[0]) > 20


@pytest.mark.parametrize("level", [0, 1, 2, 3])
@pytest.mark.parametrize("slice_b,lists", [(1, "both"), (1, "noref1"), (0, "both"), (0, "ref0only"), (1, "none")])
def test_boundary_strengths_multi_reference_and_b_pictures(level, slice_b, lists, seed=0):
    """getBoundaryStrength in full (deblock.cpp:217-247): reference picture ids per list, list-1 mvs, the B-picture four-way
    comparison - random coherent motion fields over a 1024x576 picture against the oracle restatement (pinned against the real
    Deblock class in tests/test_oracle_classes_vs_reference.py), every optional operand combination of the ABI."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    w, h = 1024, 576
    nctu, npu = (w // 64) * (h // 64), (64 >> (3 + level)) ** 2
    rng = np.random.default_rng([47, level, slice_b, len(lists), seed])

    def field():
        m = np.zeros((nctu * 85, 2), np.int32)
        x = rng.integers(-3, 4, size=nctu * 85) * 8 + rng.integers(-4, 5, size=nctu * 85) * (rng.random(nctu * 85) < 0.5)
        y = rng.integers(-2, 3, size=nctu * 85) * 8 + rng.integers(-4, 5, size=nctu * 85) * (rng.random(nctu * 85) < 0.5)
        m[:, 0] = rng.integers(0, 1 << 20, size=nctu * 85)
        m[:, 1] = (x & 0xffff) | (y << 16)
        return m

    mv0, mv1 = field(), field()
    same = rng.random(nctu * 85) < 0.4
    mv1[same] = mv0[same]
    combos = np.array([(0, 1), (1, 0), (0, 0), (0, -1), (-1, 0), (1, 1), (2, 1), (-1, 2)], dtype=np.int8)
    pick = rng.choice(len(combos), size=(nctu, npu), p=[0.3, 0.2, 0.2, 0.1, 0.05, 0.05, 0.05, 0.05])
    ref0, ref1 = np.ascontiguousarray(combos[pick, 0]), np.ascontiguousarray(combos[pick, 1])
    if not slice_b:
        ref0 = np.maximum(ref0, 0)
    if lists == "noref1":
        ref1 = None
    elif lists == "ref0only":
        ref1, mv1 = None, None
    elif lists == "none":
        ref0, ref1, mv1 = None, None, None
    ns = (rng.random((nctu, npu)) < 0.1).astype(np.uint32) * 3
    intra = (rng.random((nctu, npu)) < 0.05).astype(np.uint8)
    bv, bh = O.deblock_bs_b(8, w, h, level, mv0, mv1, ref0, ref1, ns, slice_b=slice_b, intra=intra)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)
    d_bv = torch.full((bv.size,), 9, dtype=torch.uint8, device=dev)
    d_bh = torch.full((bh.size,), 9, dtype=torch.uint8, device=dev)
    A.deblock_bs_inter(w, h, level, t(mv0), t(ns), d_bv, d_bh, intra=t(intra), slice_b=slice_b, mv1=t(mv1), ref0=t(ref0), ref1=t(ref1))
    torch.cuda.synchronize()
    assert np.array_equal(d_bv.cpu().numpy(), bv), f"vertical: {np.count_nonzero(d_bv.cpu().numpy() != bv)} units differ"
    assert np.array_equal(d_bh.cpu().numpy(), bh), f"horizontal: {np.count_nonzero(d_bh.cpu().numpy() != bh)} units differ"
    if level < 3:
        both = np.concatenate([bv, bh])
        assert (both == 0).any() and (both == 1).any() and (both == 2).any()
    if lists == "none" and not slice_b:
        pv, ph = O.deblock_bs_inter(8, w, h, level, mv0, ns, intra=intra)
        assert np.array_equal(pv, bv) and np.array_equal(ph, bh)          # the single-reference form is the special case
