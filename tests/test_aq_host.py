"""CPU: the host-side half of the adaptive-quantisation pass (x265hip_aq_offsets in libx265hip.so - the reference's double-precision
QP offsets, no device work) against the oracle's restatement of calcAdaptiveQuantFrame, which
tests/test_oracle_classes_vs_reference.py pins against the real class."""
import importlib
import os
import sys

import numpy as np
import pytest

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


@pytest.mark.parametrize("depth,mode,strength,qg", [(8, 2, 1.0, 16), (8, 3, 0.8, 16), (8, 1, 1.0, 16), (10, 2, 1.0, 16), (8, 2, 1.3, 8), (10, 3, 1.0, 8),
                                                   (10, 1, 0.7, 8), (8, 0, 1.0, 16), (8, 2, 0.0, 16)])
def test_aq_offsets_equal_oracle(depth, mode, strength, qg):
    import oracle_api as O
    clip = F.synth_clip(320, 176, 1, depth=depth, seed=7)
    yp, stride, org, w64, h64 = F.pad_plane(clip[0][0])
    energy, qp, inv, _, _ = O.aq_frame(depth, yp, stride, org, 320, 176, qg_size=qg, aq_mode=mode, aq_strength=strength, weightp=True)
    if mode == 0 or strength == 0:                                  # no energies are needed then; feed arbitrary ones
        energy = np.arange(len(qp), dtype=np.uint32)
    got_qp, got_inv = A.aq_offsets(depth, qg, mode, strength, energy)
    assert np.array_equal(got_qp, qp) and np.array_equal(got_inv, inv)
    if mode and strength:
        assert len(np.unique(got_inv)) > 4


def test_aq_offsets_reject_bad_arguments():
    e = np.ones(4, np.uint32)
    for args in ((8, 32, 2, 1.0), (9, 16, 2, 1.0), (8, 16, 4, 1.0)):
        with pytest.raises(A.X265HipError):
            A.aq_offsets(*args, e)


def test_host_side_double_math_equals_oracle_on_random_parameters():
    """The library's host-side functions are compiled by another compiler (clang, inside the .hip files) than the oracle and the
    reference (gcc): a libcall rewrite or a contraction there is a one-ulp difference that round parameter values hide (one such case:
    pow(2.0, x) -> exp2(x) in the --hevc-aq offsets).  Random strengths / ranges / costs through x265hip_aq_offsets and
    x265hip_aq_hevc_offsets against the oracle's (reference-pinned) restatements."""
    import oracle_api as O
    rng = np.random.default_rng(2027)
    for it in range(24):
        depth = int(rng.choice([8, 10, 12]))
        w, h = 16 * int(rng.integers(4, 24)), 16 * int(rng.integers(3, 14))
        y = F.synth_clip(w, h, 1, depth=depth, seed=int(rng.integers(1, 1 << 30)))[0][0]
        yp, stride, org, _, _ = F.pad_plane(y)
        mode, qg, strength = int(rng.integers(1, 4)), int(rng.choice([8, 16])), float(rng.uniform(0.05, 3.0))
        energy, qp, inv, _, _ = O.aq_frame(depth, yp, stride, org, w, h, qg_size=qg, aq_mode=mode, aq_strength=strength, weightp=False)
        got_qp, got_inv = A.aq_offsets(depth, qg, mode, strength, energy)
        assert np.array_equal(got_qp, qp) and np.array_equal(got_inv, inv), f"aq_offsets: depth {depth} mode {mode} qg {qg} strength {strength!r}"
        w2, h2, qg2, span = 2 * int(rng.integers(20, 160)), 2 * int(rng.integers(20, 120)), int(rng.choice([8, 16, 32, 64])), float(rng.uniform(1.0, 6.0))
        y2 = F.synth_clip(w2, h2, 1, depth=depth, seed=int(rng.integers(1, 1 << 30)))[0][0]
        yp2, stride2, org2, _, _ = F.pad_plane(y2)
        parts, act, qpo, avg, einv, _, _ = O.aq_hevc_frame(depth, yp2, stride2, org2, w2, h2, qg_size=qg2, qp_adaptation_range=span, weightp=False)
        at = 0
        for d in range(4):
            if not parts[d]:
                continue
            a, q, g, iv = A.aq_hevc_offsets(w2, h2, 64 >> d, span, O.aq_hevc_quadrants(depth, yp2, stride2, org2, w2, h2, 64 >> d))
            assert np.array_equal(a, act[at:at + parts[d]]) and np.array_equal(q, qpo[at:at + parts[d]]) and g == avg[d], \
                f"aq_hevc_offsets: {w2}x{h2} depth {depth} qg {qg2} layer {d} range {span!r}"
            at += parts[d]


def test_cutree_host_side_equals_oracle_on_random_parameters():
    """x265hip_cutree_finish / x265hip_frame_cost_recalculate (host-side, clang-compiled) against the oracle on random costs, strengths,
    weight deltas and frame-rate factors."""
    import oracle_api as O
    rng = np.random.default_rng(2028)
    for it in range(40):
        wcu, hcu = int(rng.integers(2, 40)), int(rng.integers(2, 30))
        n = wcu * hcu
        intra = rng.integers(0, 4000, size=n).astype(np.int32)
        invq = rng.integers(1, 1024, size=n).astype(np.int32)
        prop = rng.integers(0, 65536, size=n).astype(np.uint16)
        qpaq = rng.uniform(-3, 3, size=n)
        preset = rng.uniform(-5, 5, size=n)
        fps_q8, wdelta, strength = int(rng.integers(1, 2000)), float(rng.uniform(0, 1)) * int(rng.integers(0, 2)), float(rng.uniform(0.1, 6.0))
        want = O.cutree_finish(8, intra, invq, prop, qpaq, fps_q8, wdelta, strength, preset.copy())
        got = A.cutree_finish(intra, invq, prop, qpaq, fps_q8, wdelta, strength, preset.copy())
        assert np.array_equal(got, want), f"cutree_finish: {wcu}x{hcu} strength {strength!r} delta {wdelta!r}"
        lc = rng.integers(0, 65536, size=n).astype(np.uint16)
        score_w, rows_w = O.frame_cost_recalculate(8, wcu, hcu, lc, got)
        score_g, rows_g = A.frame_cost_recalculate(wcu, hcu, lc, got)
        assert score_g == score_w and np.array_equal(rows_g, rows_w), f"frame_cost_recalculate: {wcu}x{hcu}"


def test_cutree_qg8_and_hevc_aq_host_side_equal_oracle_on_random_parameters():
    """The --qg-size 8 and --hevc-aq flavours of the cuTree host functions against the oracle, random costs / strengths / deltas / sizes."""
    import oracle_api as O
    rng = np.random.default_rng(2029)
    for it in range(40):
        wcu, hcu = int(rng.integers(2, 40)), int(rng.integers(2, 30))
        n = wcu * hcu
        intra = rng.integers(0, 4000, size=n).astype(np.int32)
        invq = rng.integers(1, 1024, size=n).astype(np.int32)
        prop = rng.integers(0, 65536, size=n).astype(np.uint16)
        fps_q8, wdelta, strength = int(rng.integers(1, 2000)), float(rng.uniform(0, 1)) * int(rng.integers(0, 2)), float(rng.uniform(0.1, 6.0))
        qpaq, preset = rng.uniform(-3, 3, size=4 * n), rng.uniform(-5, 5, size=4 * n)
        want = O.cutree_finish_qg8(8, wcu, hcu, intra, invq, prop, qpaq, fps_q8, wdelta, strength, preset)
        got = A.cutree_finish_qg8(wcu, hcu, intra, invq, prop, qpaq, fps_q8, wdelta, strength, preset)
        assert np.array_equal(got, want), f"cutree_finish_qg8: {wcu}x{hcu} strength {strength!r}"
        lc = rng.integers(0, 65536, size=n).astype(np.uint16)
        sw, rw = O.frame_cost_recalculate_qg8(8, wcu, hcu, lc, got)
        sg, rg = A.frame_cost_recalculate_qg8(wcu, hcu, lc, got)
        assert sg == sw and np.array_equal(rg, rw)
        part = int(rng.choice([16, 32, 64]))
        width, height = 16 * wcu - int(rng.integers(0, 15)), 16 * hcu - int(rng.integers(0, 15))
        cnt = ((width + part - 1) // part) * ((height + part - 1) // part)
        qo = rng.uniform(-4, 4, size=cnt)
        intra1 = np.maximum(intra, 1)
        args = (width, height, part, wcu, intra1, np.maximum(invq, 256), prop, fps_q8, wdelta, strength, qo)
        assert np.array_equal(A.cutree_finish_hevc_aq(*args), O.cutree_finish_hevc_aq(8, *args), equal_nan=True), f"cutree_finish_hevc_aq: {width}x{height} part {part}"
