"""GPU parity: fused inter prediction + residual round trip (x265hip_inter_recon) vs the oracle's restatement of
predict.cpp:245-265 + quant.cpp:397-480,543-605 + search.cpp:357-375 driven through the oracle primitives."""
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


@pytest.mark.parametrize("flags", [0, 2, 3])          # X265HIP_TU_SIGN_HIDE = 2 (the x265 default), | X265HIP_TU_INTRA_SLICE
@pytest.mark.parametrize("depth,level,qp", [(8, 2, 22), (8, 1, 30), (8, 0, 12), (8, 2, 45), (10, 2, 34), (10, 1, 20), (8, 2, 0), (12, 2, 40), (12, 0, 30)])
def test_inter_recon_matches_oracle(depth, level, qp, flags):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([51, depth, level, qp])
    clip = F.synth_clip(256, 128, 2, depth=depth, seed=50 + level)
    y0 = clip[0][0].astype(np.float32)
    sh = np.roll(y0, (1, 1), axis=(0, 1))
    y1 = np.clip(np.rint(0.5 * y0 + 0.5 * sh + rng.normal(0, 2.0 * (1 << (depth - 8)), size=y0.shape)), 0, (1 << depth) - 1).astype(clip[0][0].dtype)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, 8, depth, dev, want_surf=False)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, 2, dev)
    sp.run(cur, ref)
    st = S.InterRecon(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=flags)
    recon = torch.zeros_like(cur.t)
    st.run(cur, ref, recon, sp.out)
    torch.cuda.synchronize()
    O = _oracle()
    mv = sp.out.cpu().numpy().reshape(-1, 2)
    erec, elev, ens, edist = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org,
                                           cur.w64, cur.h64, level, mv, qp, intra_slice=flags)
    if flags & 2 and qp in (12, 20, 22, 30):              # sign hiding really changed levels
        plain = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, level, mv, qp, intra_slice=flags & 1)[1]
        assert np.count_nonzero(plain != elev) > 0
    assert np.array_equal(st.num_sig.cpu().numpy().view(np.uint32), ens), "numSig differs"
    assert np.array_equal(st.levels.cpu().numpy(), elev), "quantised levels differ"
    grec = recon.cpu().numpy().view(cur.host.dtype).reshape(cur.host.shape)
    assert np.array_equal(grec, erec), f"recon differs at {np.count_nonzero(grec != erec)} samples"
    assert np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist), "SSE differs"
    # the case set must exercise every branch of the inverse path: full idct, DC-only shortcut, no coefficients
    if qp == 22:
        assert (ens > 1).any()
    if qp == 45:
        assert (ens == 0).any()


@pytest.mark.parametrize("depth,level,qp", [(8, 2, 24), (8, 1, 30), (8, 0, 20), (10, 2, 36), (10, 0, 30), (12, 1, 44)])
def test_inter_recon_chroma_matches_oracle(depth, level, qp):
    """The chroma planes of a 4:2:0 picture through the same stage: 1/8-sample 4-tap prediction from the luma mvs, residual round
    trip on half-size blocks (4x4 ... 16x16)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([53, depth, level])
    W, Hh = 256, 128
    clip = F.synth_clip(W, Hh, 2, depth=depth, seed=60 + level)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, 8, depth, dev, want_surf=False)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, 3, dev)
    sp.run(cur, ref)
    torch.cuda.synchronize()
    mv = sp.out.cpu().numpy().reshape(-1, 2).copy()
    # spread the mvs so that every 1/8 phase and the integer / h-only / v-only paths occur
    q = mv[:, 1]
    jit = rng.integers(-9, 10, size=(q.size, 2))
    qx = (((q & 0xffff) ^ 0x8000) - 0x8000) + jit[:, 0]
    qy = (q >> 16) + jit[:, 1]
    qx[::5] &= ~7; qy[::7] &= ~7
    mv[:, 1] = (qx & 0xffff) | (qy << 16)
    d_mv = torch.from_numpy(mv.reshape(-1)).to(dev)
    cw, ch, margin = cur.w64 // 2, cur.h64 // 2, 24
    stride = cw + 2 * margin
    org = margin * stride + margin
    O = _oracle()
    dt = cur.host.dtype
    for c in (1, 2):
        planes = []
        for k in (1, 0):                                      # current, reference
            src = clip[k][c]
            body = np.zeros((ch, cw), dt)
            body[:src.shape[0], :src.shape[1]] = src
            planes.append(np.ascontiguousarray(np.pad(body, margin, mode="edge")).reshape(-1))
        fenc_h, fref_h = planes
        flags = 2 if (level == 1 or depth == 10) else 0        # X265HIP_TU_SIGN_HIDE on some of the cases
        st = S.InterReconChroma(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=flags)
        d_f = torch.from_numpy(fenc_h.view(np.uint8)).to(dev)
        d_r = torch.from_numpy(fref_h.view(np.uint8)).to(dev)
        d_o = torch.zeros_like(d_f)
        st.run(d_f, d_r, d_o, stride, org, d_mv)
        torch.cuda.synchronize()
        erec, elev, ens, edist = O.inter_recon_chroma(depth, fenc_h, fref_h, stride, org, cur.w64, cur.h64, level, mv, qp, intra_slice=flags)
        assert np.array_equal(st.num_sig.cpu().numpy().view(np.uint32), ens), f"plane {c}: numSig differs"
        assert np.array_equal(st.levels.cpu().numpy(), elev), f"plane {c}: levels differ"
        assert np.array_equal(d_o.cpu().numpy().view(dt), erec), f"plane {c}: reconstruction differs"
        assert np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist), f"plane {c}: SSE differs"
        assert (ens > 0).any()


@pytest.mark.parametrize("depth,level,qp", [(8, 2, 26), (8, 1, 30), (8, 0, 22), (10, 2, 38), (12, 1, 46)])
def test_inter_recon_bi_matches_oracle(depth, level, qp):
    """B pictures: per block list 0, list 1 or both (predInterLumaShort of each list + addAvg), every fractional phase combination."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([55, depth, level])
    clip = F.synth_clip(256, 128, 3, depth=depth, seed=64 + level)
    cur, r0, r1 = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev), P.DevicePicture(clip[2][0], dev)
    nctu = (cur.w64 // 64) * (cur.h64 // 64)
    mvs = []
    for _ in range(2):
        qx, qy = rng.integers(-30, 31, size=nctu * 85), rng.integers(-30, 31, size=nctu * 85)
        qx[::4] &= ~3; qy[::3] &= ~3                           # integer / h-only / v-only phases too
        m = np.zeros((nctu * 85, 2), np.int32)
        m[:, 1] = (qx & 0xffff) | (qy << 16)
        mvs.append(m)
    nblk = (64 >> (3 + level)) ** 2
    dirs = rng.integers(1, 4, size=nctu * nblk).astype(np.uint8)
    flags = 2 if level == 1 else 0                          # X265HIP_TU_SIGN_HIDE on some of the cases
    st = S.InterReconBi(nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=flags)
    recon = torch.zeros_like(cur.t)
    st.run(cur, r0, r1, recon, torch.from_numpy(mvs[0].reshape(-1)).to(dev), torch.from_numpy(mvs[1].reshape(-1)).to(dev),
           dir_flags=torch.from_numpy(dirs).to(dev))
    torch.cuda.synchronize()
    O = _oracle()
    erec, elev, ens, edist = O.inter_recon_bi(depth, cur.host.reshape(-1), cur.stride, cur.org, r0.host.reshape(-1), r1.host.reshape(-1),
                                              cur.w64, cur.h64, level, mvs[0], mvs[1], qp, dir_flags=dirs, intra_slice=flags)
    assert np.array_equal(st.num_sig.cpu().numpy().view(np.uint32), ens), "numSig differs"
    assert np.array_equal(st.levels.cpu().numpy(), elev), "levels differ"
    assert np.array_equal(recon.cpu().numpy().view(cur.host.dtype).reshape(-1), erec.reshape(-1)), "reconstruction differs"
    assert np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist), "SSE differs"
    assert (dirs == 3).any() and (dirs == 1).any() and (dirs == 2).any()


@pytest.mark.parametrize("depth,level,qp,w0,w1", [(8, 2, 26, (1, 45, 6, 6), (1, 70, -9, 6)), (8, 1, 30, (1, 120, 20, 7), (0, 64, 0, 7)),
                                                  (8, 0, 22, (0, 32, 0, 5), (1, 29, -3, 5)), (10, 2, 38, (1, 61, 4, 6), None),
                                                  (10, 1, 34, None, (1, 127, -128, 7)), (12, 1, 46, (1, -20, 100, 4), (1, 90, 7, 3)),
                                                  (8, 2, 28, (1, 1, 0, 0), (1, 3, 1, 0))])
def test_inter_recon_bi_with_explicit_weights_matches_oracle(depth, level, qp, w0, w1):
    """Explicit weighted prediction in the B / weighted-P stage: blocks of one list -> predInterLumaShort + addWeightUni (weight_sp),
    blocks of both -> addWeightBi when both lists carry a table and one is present (list 0's denominator for both), else addAvg; a list
    without a table (None) or with wtPresent 0 keeps the unweighted paths."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([57, depth, level])
    clip = F.synth_clip(256, 128, 3, depth=depth, seed=68 + level)
    cur, r0, r1 = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev), P.DevicePicture(clip[2][0], dev)
    nctu = (cur.w64 // 64) * (cur.h64 // 64)
    mvs = []
    for _ in range(2):
        qx, qy = rng.integers(-30, 31, size=nctu * 85), rng.integers(-30, 31, size=nctu * 85)
        qx[::4] &= ~3; qy[::3] &= ~3
        m = np.zeros((nctu * 85, 2), np.int32)
        m[:, 1] = (qx & 0xffff) | (qy << 16)
        mvs.append(m)
    nblk = (64 >> (3 + level)) ** 2
    dirs = rng.integers(1, 4, size=nctu * nblk).astype(np.uint8)
    st = S.InterReconBi(nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=2)
    plain = S.InterReconBi(nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=2)
    recon, recon_plain = torch.zeros_like(cur.t), torch.zeros_like(cur.t)
    d_mv = [torch.from_numpy(m.reshape(-1)).to(dev) for m in mvs]
    d_dir = torch.from_numpy(dirs).to(dev)
    st.run(cur, r0, r1, recon, d_mv[0], d_mv[1], dir_flags=d_dir, weights=(w0, w1))
    plain.run(cur, r0, r1, recon_plain, d_mv[0], d_mv[1], dir_flags=d_dir)
    torch.cuda.synchronize()
    O = _oracle()
    erec, elev, ens, edist = O.inter_recon_bi(depth, cur.host.reshape(-1), cur.stride, cur.org, r0.host.reshape(-1), r1.host.reshape(-1),
                                              cur.w64, cur.h64, level, mvs[0], mvs[1], qp, dir_flags=dirs, intra_slice=2, weights=(w0, w1))
    assert np.array_equal(st.num_sig.cpu().numpy().view(np.uint32), ens), "numSig differs"
    assert np.array_equal(st.levels.cpu().numpy(), elev), "levels differ"
    assert np.array_equal(recon.cpu().numpy().view(cur.host.dtype).reshape(-1), erec.reshape(-1)), "reconstruction differs"
    assert np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist), "SSE differs"
    assert not torch.equal(st.levels, plain.levels), "the weights changed nothing"


@pytest.mark.parametrize("depth,level,qp,w0,w1", [(8, 2, 26, None, None), (8, 1, 30, (1, 45, 6, 6), (1, 70, -9, 6)), (8, 0, 22, (1, 120, 20, 7), (0, 64, 0, 7)),
                                                  (10, 2, 38, (1, 61, 4, 6), None), (10, 0, 34, None, (1, 127, -128, 7)), (12, 1, 46, (1, -20, 100, 4), (1, 90, 7, 3))])
def test_inter_recon_chroma_bi_matches_oracle(depth, level, qp, w0, w1):
    """The chroma planes of a B picture (x265hip_inter_recon_chroma_bi): 1/8-sample 4-tap prediction of one list or both
    (predInterChromaShort + addAvg), with and without explicit weights (addWeightUni / addWeightBi with the plane's own table)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([59, depth, level])
    clip = F.synth_clip(256, 128, 3, depth=depth, seed=70 + level)
    pics = [P.DevicePicture(c[0], dev, c[1], c[2]) for c in clip]
    cur, r0, r1 = pics[1], pics[0], pics[2]
    nctu = (cur.w64 // 64) * (cur.h64 // 64)
    mvs = []
    for _ in range(2):
        qx, qy = rng.integers(-40, 41, size=nctu * 85), rng.integers(-40, 41, size=nctu * 85)
        qx[::4] &= ~7; qy[::3] &= ~7                           # integer / h-only / v-only phases too
        m = np.zeros((nctu * 85, 2), np.int32)
        m[:, 1] = (qx & 0xffff) | (qy << 16)
        mvs.append(m)
    nblk = (64 >> (3 + level)) ** 2
    dirs = rng.integers(1, 4, size=nctu * nblk).astype(np.uint8)
    d_mv = [torch.from_numpy(m.reshape(-1)).to(dev) for m in mvs]
    d_dir = torch.from_numpy(dirs).to(dev)
    O = _oracle()
    weights = None if (w0 is None and w1 is None) else (w0, w1)
    for c in range(2):
        st = S.InterReconChromaBi(nctu, cur.w64, cur.h64, depth, level, qp - c, dev, intra_slice=2)
        recon = torch.zeros_like(cur.c[c])
        st.run(cur.c[c], r0.c[c], r1.c[c], recon, cur.stride_c, cur.org_c, d_mv[0], d_mv[1], dir_flags=d_dir, weights=weights)
        torch.cuda.synchronize()
        erec, elev, ens, edist = O.inter_recon_chroma_bi(depth, cur.c_host[c].reshape(-1), r0.c_host[c].reshape(-1), r1.c_host[c].reshape(-1), cur.stride_c, cur.org_c,
                                                         cur.w64, cur.h64, level, mvs[0], mvs[1], qp - c, dir_flags=dirs, intra_slice=2, weights=weights)
        assert np.array_equal(st.num_sig.cpu().numpy().view(np.uint32), ens), f"plane {c}: numSig differs"
        assert np.array_equal(st.levels.cpu().numpy(), elev), f"plane {c}: levels differ"
        assert np.array_equal(recon.cpu().numpy().view(cur.host.dtype).reshape(-1), erec.reshape(-1)), f"plane {c}: reconstruction differs"
        assert np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist), f"plane {c}: SSE differs"
        assert (ens > 0).any()
        if weights is None:                                    # blocks of one list: exactly the uni-directional chroma stage
            uni = S.InterReconChroma(nctu, cur.w64, cur.h64, depth, level, qp - c, dev, intra_slice=2)
            rec_u = torch.zeros_like(cur.c[c])
            uni.run(cur.c[c], r0.c[c], rec_u, cur.stride_c, cur.org_c, d_mv[0])
            n = 4 << level
            only0 = torch.from_numpy(np.repeat(dirs == 1, n * n)).to(dev)
            assert torch.equal(uni.levels[only0], st.levels[only0])


def test_inter_recon_bi_rejects_bad_weights():
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(128, 64, 2, depth=8, seed=3)
    cur, r0 = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    st = S.InterReconBi(2, cur.w64, cur.h64, 8, 2, 26, dev)
    mv = torch.zeros(2 * 85 * 2, dtype=torch.int32, device=dev)
    with pytest.raises(A.X265HipError):
        st.run(cur, r0, r0, torch.zeros_like(cur.t), mv, mv, weights=((1, 64, 0, 9), None))


QUANT_SCALES, INV_QUANT_SCALES = (26214, 23302, 20560, 18396, 16384, 14564), (40, 45, 51, 57, 64, 72)      # scalinglist.cpp:129-130


def _tables(rng, n, qp, depth, dev, lists=True, nr=True):
    """A scaling matrix of the kind ScalingList::setupQuantMatrices turns into coefficient tables (quantCoef = (scale << 4) / m,
    dequantCoef = invScale * m, scalinglist.cpp) + denoiser offsets; returns (host arrays, device record, device residual sums)."""
    import torch
    rem = qp % 6
    m = rng.integers(8, 64, size=n * n)
    qc = ((QUANT_SCALES[rem] << 4) // m).astype(np.int32) if lists else None
    dqc = (INV_QUANT_SCALES[rem] * m).astype(np.int32) if lists else None
    off = rng.integers(0, 5 << (depth - 8), size=n * n).astype(np.uint16) if nr else None
    d = lambda a: None if a is None else torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to(dev)
    d_sum = torch.zeros(n * n, dtype=torch.int32, device=dev) if nr else None
    rec = A.tu_tables(d(qc), d(dqc), d(off), d_sum)
    return (qc, dqc, off), rec, d_sum


@pytest.mark.parametrize("depth,level,qp,with_tables", [(8, 2, 24, False), (8, 1, 30, True), (8, 0, 22, False), (10, 2, 34, True), (12, 1, 40, False)])
def test_inter_recon_captures_transform_coefficients_and_delta_u(depth, level, qp, with_tables):
    """x265hip_tu_tables.dct_coeff_out / delta_u_out: what Quant::transformNxN hands to the quantiser (m_resiDctCoeff, after the denoiser)
    and primitives.quant's deltaU, for a host-side RDOQ pass on device-produced transforms - luma and one chroma plane."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([58, depth, level, qp])
    clip = F.synth_clip(192, 128, 2, depth=depth, seed=80 + level)
    cur, ref = P.DevicePicture(clip[1][0], dev, clip[1][1], clip[1][2]), P.DevicePicture(clip[0][0], dev, clip[0][1], clip[0][2])
    ms = P.MotionSearch(cur.w64, cur.h64, 8, depth, dev, want_surf=False)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, 3, dev)
    sp.run(cur, ref)
    torch.cuda.synchronize()
    mv = sp.out.cpu().numpy().reshape(-1, 2)
    O = _oracle()
    for chroma in (False, True):
        n = (8 << level) >> (1 if chroma else 0)
        (qc, dqc, off), rec0, d_sum = _tables(rng, n, qp, depth, dev, with_tables, with_tables)
        st = (S.InterReconChroma if chroma else S.InterRecon)(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=2)
        d_dct = torch.full_like(st.levels, 0x5a5a)
        d_du = torch.full((st.levels.numel(),), 0x5a5a5a5a, dtype=torch.int32, device=dev)
        t = lambda a: None if a is None else torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to(dev)
        st.tables = A.tu_tables(t(qc), t(dqc), t(off), d_sum, d_dct, d_du)
        e_dct, e_du = np.zeros(st.levels.numel(), np.int16), np.zeros(st.levels.numel(), np.int32)
        osum = np.zeros(n * n, np.uint32)
        O.set_tu_tables(depth, qc, dqc, off, osum if with_tables else None)
        O.set_tu_capture(depth, e_dct, e_du)
        try:
            if chroma:
                out = torch.zeros_like(cur.c[0])
                st.run(cur.c[0], ref.c[0], out, cur.stride_c, cur.org_c, sp.out)
                elev = O.inter_recon_chroma(depth, cur.c_host[0].reshape(-1), ref.c_host[0].reshape(-1), cur.stride_c, cur.org_c, cur.w64, cur.h64, level, mv, qp,
                                            intra_slice=2)[1]
            else:
                recon = torch.zeros_like(cur.t)
                st.run(cur, ref, recon, sp.out)
                elev = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, level, mv, qp, intra_slice=2)[1]
        finally:
            O.set_tu_tables(depth)
            O.set_tu_capture(depth)
        torch.cuda.synchronize()
        assert np.array_equal(st.levels.cpu().numpy(), elev)
        assert np.array_equal(d_dct.cpu().numpy(), e_dct), f"{'chroma' if chroma else 'luma'}: captured coefficients differ"
        assert np.array_equal(d_du.cpu().numpy(), e_du), f"{'chroma' if chroma else 'luma'}: captured deltaU differs"
        assert np.count_nonzero(e_dct) > 100


@pytest.mark.parametrize("depth,level,qp,lists,nr", [(8, 2, 24, True, True), (8, 1, 30, True, False), (8, 0, 20, False, True), (10, 2, 36, True, True), (12, 1, 44, True, True)])
def test_inter_recon_with_scaling_lists_and_denoiser(depth, level, qp, lists, nr):
    """x265hip_tu_tables: scaling-list quantiser / dequantiser coefficients and the denoiser (running residual sums) in the fused inter TU
    stage, luma and both chroma planes, sign hiding on - against the oracle, whose table paths are pinned against the real Quant."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([57, depth, level, qp])
    clip = F.synth_clip(256, 128, 2, depth=depth, seed=70 + level)
    cur, ref = P.DevicePicture(clip[1][0], dev, clip[1][1], clip[1][2]), P.DevicePicture(clip[0][0], dev, clip[0][1], clip[0][2])
    ms = P.MotionSearch(cur.w64, cur.h64, 8, depth, dev, want_surf=False)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, 3, dev)
    sp.run(cur, ref)
    torch.cuda.synchronize()
    mv = sp.out.cpu().numpy().reshape(-1, 2)
    O = _oracle()
    flags = 2
    # luma
    n = 8 << level
    (qc, dqc, off), rec, d_sum = _tables(rng, n, qp, depth, dev, lists, nr)
    st = S.InterRecon(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=flags)
    st.tables = rec
    recon = torch.zeros_like(cur.t)
    st.run(cur, ref, recon, sp.out)
    torch.cuda.synchronize()
    osum = np.zeros(n * n, np.uint32)
    O.set_tu_tables(depth, qc, dqc, off, osum if nr else None)
    try:
        erec, elev, ens, edist = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, level, mv, qp, intra_slice=flags)
    finally:
        O.set_tu_tables(depth)
    assert np.array_equal(st.levels.cpu().numpy(), elev), "luma levels differ"
    assert np.array_equal(st.num_sig.cpu().numpy().view(np.uint32), ens) and np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist)
    assert np.array_equal(recon.cpu().numpy().view(cur.host.dtype).reshape(cur.host.shape), erec)
    if nr:
        assert np.array_equal(d_sum.cpu().numpy().view(np.uint32), osum) and osum.sum() > 0
    plain = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, level, mv, qp, intra_slice=flags)[1]
    assert not np.array_equal(plain, elev), "the tables changed nothing: the case does not exercise them"
    # one chroma plane
    nc = n // 2
    (qc, dqc, off), rec, d_sum = _tables(rng, nc, qp, depth, dev, lists, nr)
    stc = S.InterReconChroma(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=flags)
    stc.tables = rec
    out = torch.zeros_like(cur.c[0])
    stc.run(cur.c[0], ref.c[0], out, cur.stride_c, cur.org_c, sp.out)
    torch.cuda.synchronize()
    osum = np.zeros(nc * nc, np.uint32)
    O.set_tu_tables(depth, qc, dqc, off, osum if nr else None)
    try:
        erec, elev, ens, edist = O.inter_recon_chroma(depth, cur.c_host[0].reshape(-1), ref.c_host[0].reshape(-1), cur.stride_c, cur.org_c, cur.w64, cur.h64, level, mv, qp,
                                                      intra_slice=flags)
    finally:
        O.set_tu_tables(depth)
    assert np.array_equal(stc.levels.cpu().numpy(), elev) and np.array_equal(stc.num_sig.cpu().numpy().view(np.uint32), ens)
    assert np.array_equal(out.cpu().numpy().view(cur.host.dtype), erec.reshape(-1))
    if nr:
        assert np.array_equal(d_sum.cpu().numpy().view(np.uint32), osum)


@pytest.mark.parametrize("depth,level,qp", [(8, 2, 24), (8, 0, 20), (10, 2, 36), (10, 1, 30)])
def test_inter_recon_chroma_pair_equals_two_single_plane_calls(depth, level, qp):
    """x265hip_inter_recon_chroma_pair: Cb and Cr in ONE launch, each plane with its own QP / flags / outputs, against the single-plane
    entry the test above pins on the oracle."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(256, 192, 2, depth=depth, seed=66 + level)
    cur, ref = P.DevicePicture(clip[1][0], dev, clip[1][1], clip[1][2]), P.DevicePicture(clip[0][0], dev, clip[0][1], clip[0][2])
    ms = P.MotionSearch(cur.w64, cur.h64, 8, depth, dev, want_surf=False)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, 3, dev)
    sp.run(cur, ref)
    single = [S.InterReconChroma(ms.nctu, cur.w64, cur.h64, depth, level, qp - 2 * i, dev, intra_slice=2 * i) for i in range(2)]
    pair = [S.InterReconChroma(ms.nctu, cur.w64, cur.h64, depth, level, qp - 2 * i, dev, intra_slice=2 * i) for i in range(2)]
    rec_single = [torch.zeros_like(cur.c[0]) for _ in range(2)]
    rec_pair = [torch.zeros_like(cur.c[0]) for _ in range(2)]
    for i in range(2):
        single[i].run(cur.c[i], ref.c[i], rec_single[i], cur.stride_c, cur.org_c, sp.out)
    S.InterReconChroma.run_pair(pair, cur.c, ref.c, rec_pair, cur.stride_c, cur.org_c, sp.out)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(rec_single[i], rec_pair[i]), f"plane {i}: reconstruction differs"
        assert torch.equal(single[i].levels, pair[i].levels) and torch.equal(single[i].num_sig, pair[i].num_sig) and torch.equal(single[i].dist, pair[i].dist), f"plane {i}"
        assert int(pair[i].num_sig.sum()) > 0
    assert not torch.equal(pair[0].levels, pair[1].levels)
