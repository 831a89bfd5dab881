"""GPU parity: the adaptive-quantisation pass (x265hip_aq_energy + the host-side double-precision offsets of stages.AdaptiveQuant) vs
the oracle's restatement of LookaheadTLD::calcAdaptiveQuantFrame (oracle/x265_oracle_pipeline3.c), which
tests/test_oracle_classes_vs_reference.py pins against the real class."""
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


def _pad(img, margin=16):
    h, w = img.shape
    buf = np.pad(img, margin, mode="edge")
    return np.ascontiguousarray(buf).reshape(-1), w + 2 * margin, margin * (w + 2 * margin) + margin


@pytest.mark.parametrize("depth,width,height,qg,mode,strength,chroma", [(8, 640, 360, 16, 2, 1.0, True), (8, 640, 360, 16, 1, 1.0, True), (8, 416, 240, 16, 3, 0.8, True),
                                                                    (8, 640, 352, 8, 2, 1.0, True), (8, 250, 138, 16, 2, 1.0, False), (10, 384, 256, 16, 2, 1.0, True),
                                                                    (10, 384, 256, 8, 1, 0.6, False), (8, 3840, 2160, 16, 2, 1.0, True), (8, 256, 128, 16, 0, 1.0, True)])
def test_aq_pass_matches_oracle(depth, width, height, qg, mode, strength, chroma, seed=101):
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    yimg, cbimg, crimg = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0]
    pic = P.DevicePicture(yimg, dev)
    kw, okw = {}, {}
    if chroma:
        cpad = [_pad(np.ascontiguousarray(c)) for c in (cbimg, crimg)]
        to_dev = lambda a: torch.from_numpy(a.view(np.uint8) if depth == 8 else a.view(np.int16)).to(dev)
        kw = dict(cb=to_dev(cpad[0][0]), cr=to_dev(cpad[1][0]), stride_c=cpad[0][1], org_c=cpad[0][2])
        okw = dict(cb=cpad[0][0], cr=cpad[1][0], stride_c=cpad[0][1], org_c=cpad[0][2])
    aq = S.AdaptiveQuant(width, height, depth, dev, qg_size=qg, aq_mode=mode, aq_strength=strength, weightp=True)
    qp, inv, wp_sum, wp_ssd = aq.run(pic, **kw)
    energy, eqp, einv, esum, essd = O.aq_frame(depth, pic.host.reshape(-1), pic.stride, pic.org, width, height, qg_size=qg, aq_mode=mode,
                                               aq_strength=strength, weightp=True, **okw)
    assert np.array_equal(aq.energy.cpu().numpy().view(np.uint32), energy), "block energies differ"
    assert wp_sum == [int(v) for v in esum] and wp_ssd == [int(v) for v in essd], "wp statistics differ"
    assert np.array_equal(qp, eqp), f"{np.count_nonzero(qp != eqp)} QP offsets differ"
    assert np.array_equal(inv, einv)
    if mode:
        assert len(np.unique(inv)) > 4


def test_aq_energy_rejects_bad_arguments():
    import torch
    dev = torch.device("cuda:0")
    pic = P.DevicePicture(F.synth_clip(64, 64, 1, depth=8, seed=1)[0][0], dev)
    energy, wp = torch.zeros(64, dtype=torch.int32, device=dev), torch.zeros(6, dtype=torch.int64, device=dev)
    with pytest.raises(A.X265HipError):
        A.aq_energy(8, pic.t, pic.stride, pic.org, 64, 64, 32, energy, wp)
    with pytest.raises(A.X265HipError):
        A.aq_energy(8, pic.t, pic.stride, pic.org, 64, 64, 16, energy, wp, cb=pic.t)


@pytest.mark.parametrize("depth,width,height,qg,rng,chroma", [(8, 640, 360, 16, 1.0, True), (8, 416, 240, 32, 2.5, True), (8, 250, 138, 64, 1.0, False),
                                                              (8, 640, 352, 8, 6.0, True), (10, 384, 256, 16, 1.0, True), (10, 232, 120, 8, 3.0, False),
                                                              (8, 3840, 2160, 16, 1.0, True), (10, 3840, 2160, 8, 2.0, False), (12, 136, 72, 64, 2.0, False)])
def test_hevc_aq_pass_matches_oracle(depth, width, height, qg, rng, chroma, seed=103):
    """--hevc-aq (stages.HevcAq: x265hip_aq_hevc_quadrants per layer + the host-side x265hip_aq_hevc_offsets + the wp statistics of
    x265hip_aq_energy) vs the oracle's restatement of xPreanalyze / xPreanalyzeQp, pinned against the real class on the CPU."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    yimg, cbimg, crimg = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0]
    pic = P.DevicePicture(yimg, dev)
    kw, okw = {}, {}
    if chroma:
        cpad = [_pad(np.ascontiguousarray(c)) for c in (cbimg, crimg)]
        to_dev = lambda a: torch.from_numpy(a.view(np.uint8) if depth == 8 else a.view(np.int16)).to(dev)
        kw = dict(cb=to_dev(cpad[0][0]), cr=to_dev(cpad[1][0]), stride_c=cpad[0][1], org_c=cpad[0][2])
        okw = dict(cb=cpad[0][0], cr=cpad[1][0], stride_c=cpad[0][1], org_c=cpad[0][2])
    aq = S.HevcAq(width, height, depth, dev, qg_size=qg, qp_adaptation_range=rng, weightp=True)
    layers, inv, wp_sum, wp_ssd = aq.run(pic, **kw)
    parts, act, qp, avg, einv, esum, essd = O.aq_hevc_frame(depth, pic.host.reshape(-1), pic.stride, pic.org, width, height, qg_size=qg,
                                                            qp_adaptation_range=rng, weightp=True, **okw)
    at = 0
    for d in range(4):
        if not parts[d]:
            assert (64 >> d) not in layers
            continue
        part = 64 >> d
        sums = aq.sums[aq.parts.index(part)].cpu().numpy().view(np.uint64).reshape(-1, 4, 2)
        assert np.array_equal(sums, O.aq_hevc_quadrants(depth, pic.host.reshape(-1), pic.stride, pic.org, width, height, part)), f"layer {d}: quadrant sums differ"
        a, q, g = layers[part]
        assert np.array_equal(a, act[at:at + parts[d]]) and np.array_equal(q, qp[at:at + parts[d]]) and g == avg[d], f"layer {d}: activities / offsets differ"
        at += parts[d]
    assert np.array_equal(inv, einv) and len(np.unique(inv)) > 3
    assert wp_sum == [int(v) for v in esum] and wp_ssd == [int(v) for v in essd], "wp statistics differ"


def test_hevc_aq_rejects_bad_arguments():
    import torch
    dev = torch.device("cuda:0")
    pic = P.DevicePicture(F.synth_clip(64, 64, 1, depth=8, seed=1)[0][0], dev)
    sums = torch.zeros(64, dtype=torch.int64, device=dev)
    with pytest.raises(A.X265HipError):
        A.aq_hevc_quadrants(8, pic.t, pic.stride, pic.org, 64, 64, 24, sums)
    with pytest.raises(A.X265HipError):
        A.aq_hevc_offsets(64, 64, 16, 0.5, np.zeros((16, 4, 2), np.uint64))


# ---- round 4: the same pass behind host pointers, as the pre-lookahead calls it ------------------------------------------------------------
@pytest.mark.parametrize("depth,width,height,qg,mode,strength,chroma,weightp", [(8, 640, 360, 16, 2, 1.0, True, True), (8, 416, 240, 16, 1, 1.0, True, False),
                                                                            (8, 640, 352, 8, 3, 1.3, True, True), (8, 250, 138, 16, 2, 1.0, False, True),
                                                                            (10, 384, 256, 8, 2, 0.6, True, True), (12, 256, 128, 16, 3, 1.0, True, True),
                                                                            (8, 3840, 2160, 16, 2, 1.0, True, True), (10, 3840, 2160, 16, 2, 1.0, True, True)])
def test_aq_frame_host_fills_the_lowres_arrays_like_the_reference_loop(depth, width, height, qg, mode, strength, chroma, weightp, seed=202):
    """x265hip_aq_frame_host: planes in host memory in (strided, padded, as PicYuv holds them), the Lowres arrays out - block energies, both QP offset
    arrays, x265_exp2fix8 factors, the 2x2 averages of --qg-size 8, wp_sum and the normalised (or raw) wp_ssd - against the oracle's
    calcAdaptiveQuantFrame, which tests/test_oracle_classes_vs_reference.py pins to the real class."""
    O = _oracle()
    yimg, cbimg, crimg = F.synth_clip(width, height, 1, depth=depth, seed=seed)[0]
    ypad = _pad(np.ascontiguousarray(yimg), 32)
    kw = {}
    if chroma:
        cpad = [_pad(np.ascontiguousarray(c)) for c in (cbimg, crimg)]
        kw = dict(cb=cpad[0][0], cr=cpad[1][0], stride_c=cpad[0][1], org_c=cpad[0][2])
    grid = (((width // 2) + 7) // 8, ((height // 2) + 7) // 8)
    use8 = qg == 8 and 2 * grid[0] <= (width + 7) // 8 and 2 * grid[1] <= (height + 7) // 8
    got = A.aq_frame_host(depth, ypad[0], ypad[1], ypad[2], width, height, qg, mode, strength, normalise_wp=weightp, lowres_grid=grid if use8 else None, **kw)
    energy, eqp, einv, esum, essd = O.aq_frame(depth, ypad[0], ypad[1], ypad[2], width, height, qg_size=qg, aq_mode=mode, aq_strength=strength, weightp=weightp, **kw)
    assert np.array_equal(got["energy"], energy), "block energies differ"
    assert np.array_equal(got["wp_sum"], esum) and np.array_equal(got["wp_ssd"], essd), (got["wp_sum"], esum, got["wp_ssd"], essd)
    assert np.array_equal(got["qp_aq_offset"], eqp) and np.array_equal(got["qp_cutree_offset"], eqp), f"{np.count_nonzero(got['qp_aq_offset'] != eqp)} QP offsets differ"
    assert np.array_equal(got["inv_qscale"], einv) and len(np.unique(einv)) > 4
    if use8:
        bw = (width + 7) // 8
        f = einv.reshape(-1, bw)
        want = (f[0:2 * grid[1]:2, 0:2 * grid[0]:2] + f[0:2 * grid[1]:2, 1:2 * grid[0]:2] + f[1:2 * grid[1]:2, 0:2 * grid[0]:2] + f[1:2 * grid[1]:2, 1:2 * grid[0]:2]) // 4
        assert np.array_equal(got["inv_qscale_8x8"].reshape(grid[1], grid[0]), want)


def test_aq_frame_host_rejects_what_it_does_not_cover():
    y = np.zeros(64 * 64, np.uint8)
    for bad in (dict(qg_size=32), dict(aq_mode=0), dict(aq_mode=4), dict(aq_strength=0.0), dict(depth=9)):
        kw = dict(depth=8, qg_size=16, aq_mode=2, aq_strength=1.0)
        kw.update(bad)
        with pytest.raises(A.X265HipError):
            A.aq_frame_host(kw["depth"], y, 64, 0, 64, 64, kw["qg_size"], kw["aq_mode"], kw["aq_strength"])
    with pytest.raises(A.X265HipError):
        A.aq_frame_host(8, y, 64, 0, 64, 64, 16, 2, 1.0, cb=y)          # cb without cr


@pytest.mark.parametrize("depth", [8, 10])
def test_aq_frame_host_reproduces_what_the_reference_left_in_lowres(depth):
    """tests/golden/aq_frame_d*.npz: the picture x265's own calcAdaptiveQuantFrame was handed in a real encode and the arrays IT filled (tools/gen_weight_golden.py);
    the service against them directly, doubles bit for bit - no oracle in between."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import weight_fixture as WF
    for i, c in enumerate(WF.aq_cases(depth)):
        got = A.aq_frame_host(depth, c["y"], c["stride"], c["org"], c["width"], c["height"], c["qg"], c["mode"], c["strength"], cb=c["cb"], cr=c["cr"], stride_c=c["stride_c"],
                              org_c=c["org_c"], normalise_wp=c["weightp"], lowres_grid=c["grid"] if c["qg"] == 8 else None)
        assert np.array_equal(got["qp_aq_offset"], c["qp_aq_offset"]) and np.array_equal(got["qp_cutree_offset"], c["qp_cutree_offset"]), i
        assert np.array_equal(got["inv_qscale"], c["inv_qscale"]) and np.array_equal(got["wp_sum"], c["wp_sum"]) and np.array_equal(got["wp_ssd"], c["wp_ssd"]), i
        if c["qg"] == 8:
            assert np.array_equal(got["inv_qscale_8x8"], c["inv_qscale_8x8"]), i
