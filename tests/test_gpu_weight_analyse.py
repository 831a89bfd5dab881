"""GPU parity: x265hip_weight_analyse_host (the frame encoder's weightAnalyse, encoder/weightPrediction.cpp:222-497 - compensated reference planes,
weightCost of every scale / offset pair, the decision logic) against the oracle's restatement (oracle/x265_oracle_pipeline7.c), which
tests/test_seam_cpu.py pins in situ against the reference's own function on fading clips."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _smooth(rng, h, w, maxv, dtype):
    """a picture with structure at several scales (a weighted copy of it is a plausible faded frame)"""
    img = np.zeros((h, w), np.float64)
    for s in (32, 8, 2):
        c = rng.random((h // s + 2, w // s + 2))
        img += np.kron(c, np.ones((s, s)))[:h, :w] * s
    img = (img - img.min()) / (img.max() - img.min())
    return np.clip(img * 0.7 * maxv + 0.1 * maxv + rng.normal(0, 1.5, (h, w)), 0, maxv).astype(dtype)


def _pad(img, m):
    buf = np.ascontiguousarray(np.pad(img, m, mode="edge"))
    return buf.reshape(-1), buf.shape[1], m * buf.shape[1] + m


def _case(depth, width, height, nlists, scale, offset, with_mvs, seed, chroma_fade=True):
    rng = np.random.default_rng(seed)
    maxv, dt = (1 << depth) - 1, np.uint8 if depth == 8 else np.uint16
    lw, lh, cw, ch = width // 2, height // 2, width // 2, height // 2
    LM, CM = 40, 32
    npx = (((width + 15) >> 4) << 4) * (((height + 15) >> 4) << 4)

    def stats(luma, cb, cr):           # the figures calcAdaptiveQuantFrame would leave (sum, ssd - sum^2 / n), scaled to the picture area
        out_s, out_q = [], []
        for pl, n in ((luma, npx), (cb, npx // 4), (cr, npx // 4)):
            v = pl.astype(np.float64)
            k = n / v.size
            s, q = v.sum() * k, (v * v).sum() * k
            out_s.append(int(s)); out_q.append(int(max(0.0, q - s * s / n)))
        return out_q, out_s

    def fade(pl, sc, of):
        return np.clip(pl.astype(np.float64) * sc + of * (1 << (depth - 8)) + rng.normal(0, 0.7, pl.shape), 0, maxv).astype(dt)

    refs = []
    base_l = [_smooth(rng, lh, lw, maxv, dt) for _ in range(4)]
    base_cb, base_cr = _smooth(rng, ch, cw, maxv, dt), _smooth(rng, ch, cw, maxv, dt)
    nblk = (lw // 8) * (lh // 8)
    for i in range(nlists):
        l4 = [np.roll(p, i * 3, axis=1) for p in base_l]
        cb, cr = np.roll(base_cb, i, axis=0), np.roll(base_cr, i, axis=1)
        mvs = None
        if with_mvs:
            mvs = rng.integers(-40, 41, (nblk, 2)).astype(np.int32)
            mvs[rng.random(nblk) < 0.05] = rng.integers(-3000, 3000, 2)           # far outside: the clip to 8 samples beyond the plane
            mvs[rng.random(nblk) < 0.2] &= ~3                                     # whole-sample vectors
            mvs[rng.random(nblk) < 0.2, 0] &= ~7                                  # the chroma filter's horizontal-only / vertical-only branches
            mvs[rng.random(nblk) < 0.2, 1] &= ~7
        q, s = stats(l4[0], cb, cr)
        lp = [_pad(p, LM) for p in l4]
        cbp, crp = _pad(cb, CM), _pad(cr, CM)
        refs.append(dict(lowres=[(a, o) for a, _, o in lp], cb=(cbp[0], cbp[2]), cr=(crp[0], crp[2]), mvs=mvs, wp_ssd=q, wp_sum=s, _stride=lp[0][1], _stride_c=cbp[1]))
    cur_l = fade(base_l[0], scale, offset)
    cur_cb = fade(base_cb, scale if chroma_fade else 1.0, 0)
    cur_cr = fade(base_cr, scale if chroma_fade else 1.0, 0)
    q, s = stats(cur_l, cur_cb, cur_cr)
    lp, cbp, crp = _pad(cur_l, LM), _pad(cur_cb, CM), _pad(cur_cr, CM)
    cur = dict(lowres=(lp[0], lp[2]), lowres_stride=lp[1], lowres_width=lw, lowres_lines=lh, cb=(cbp[0], cbp[2]), cr=(crp[0], crp[2]), stride_c=cbp[1], wp_ssd=q, wp_sum=s)
    intra = rng.integers(150 << (depth - 8), 4000 << (depth - 8), nblk).astype(np.int32)
    return cur, refs, intra, (LM, LM), (CM, CM)


@pytest.mark.parametrize("depth,width,height,nlists,scale,offset,with_mvs", [(8, 256, 192, 1, 0.8, 6, True), (8, 256, 192, 2, 0.65, -4, True), (8, 320, 256, 1, 1.25, 10, False),
                                                                         (10, 256, 192, 2, 0.8, 6, True), (12, 192, 128, 1, 0.7, 3, True), (8, 256, 192, 1, 1.0, 0, True),
                                                                         (8, 1920, 1088, 2, 0.85, 5, True), (8, 3840, 2176, 1, 0.9, 4, True)])
def test_weight_analyse_host_returns_the_table_of_the_restatement(depth, width, height, nlists, scale, offset, with_mvs):
    O = _oracle()
    cur, refs, intra, lm, cm = _case(depth, width, height, nlists, scale, offset, with_mvs, seed=width + depth + nlists)
    want_w, want_d = O.weight_analyse(depth, cur, refs, width, height, intra)
    got_w, got_d = A.weight_analyse_host(depth, cur, refs, width, height, intra, lm, cm)
    assert np.array_equal(got_w, want_w), (got_w.tolist(), want_w.tolist())
    assert np.array_equal(got_d, want_d)
    if scale != 1.0:
        assert want_w[0, 0, 0] == 1, "the case was meant to find a luma weight"
    else:
        assert not want_w[:, :, 0].any()


def test_weight_analyse_host_with_keyed_planes_gives_the_same_answer_twice():
    """plane keys: the lowres planes stay on the device between calls (shared with x265hip_lowres_cost_host's cache); x265hip_lowres_planes_forget drops them"""
    O = _oracle()
    cur, refs, intra, lm, cm = _case(8, 256, 192, 2, 0.75, 5, True, seed=9)
    want = O.weight_analyse(8, cur, refs, 256, 192, intra)
    keys = [(77 << 32) | 1, (77 << 32) | 2, (77 << 32) | 3]
    for _ in range(2):
        got = A.weight_analyse_host(8, cur, refs, 256, 192, intra, lm, cm, plane_keys=keys)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    A.lib().x265hip_lowres_planes_forget()


def test_weight_analyse_host_rejects_bad_geometry():
    cur, refs, intra, lm, cm = _case(8, 256, 192, 1, 0.8, 6, False, seed=3)
    with pytest.raises(A.X265HipError):
        A.weight_analyse_host(8, cur, refs, 256, 192, intra, (8, 8), cm)               # lowres margins the compensation would leave
    with pytest.raises(A.X265HipError):
        A.weight_analyse_host(8, cur, refs, 256, 192, intra, lm, (8, 8))
    bad = dict(cur); bad["lowres_width"] = 124
    with pytest.raises(A.X265HipError):
        A.weight_analyse_host(8, bad, refs, 256, 192, intra, lm, cm)
    with pytest.raises(A.X265HipError):
        A.weight_analyse_host(9, cur, refs, 256, 192, intra, lm, cm)


@pytest.mark.parametrize("depth", [8, 10])
def test_weight_analyse_host_reproduces_the_weights_the_reference_chose(depth):
    """tests/golden/weight_analyse_d*.npz: what x265's own weightAnalyse was handed inside real encodes of fading clips and what it answered (tools/gen_weight_golden.py;
    weighted P slices, two-list B slices, slices that keep weight 1).  The service must choose the same weights - no oracle in between."""
    import weight_fixture as WF
    for i, c in enumerate(WF.cases(depth)):
        got, den = A.weight_analyse_host(depth, c["cur"], c["refs"], c["pic"][0], c["pic"][1], c["intra"], c["lowres_margin"], c["chroma_margin"])
        assert np.array_equal(got[:c["nlists"]], c["expected"][:c["nlists"]]), (i, got.tolist(), c["expected"].tolist())
        assert [int(den[l, 0]) for l in range(c["nlists"])] == [int(c["expected"][l, 0, 2]) for l in range(c["nlists"])]
