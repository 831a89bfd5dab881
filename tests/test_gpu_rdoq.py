"""GPU parity of the RDOQ pre-passes emitted by the fused TU stages (x265hip_tu_tables.rdoq_*, round 3): the data-parallel half of
Quant::rdoQuant (quant.cpp:609+) - primitives.nquant's levels / count and, per coefficient and per 4x4 coefficient group, what the
slots cu[].nonPsyRdoQuant / cu[].psyRdoQuant (= _1p + _2p, dct.cpp:986-1069) compute - produced next to the transform.

Expected side: the ORACLE TABLE's own slots (oracle/x265_oracle_host.c, pinned slot by slot against the real reference table in
tests/test_oracle_vs_reference.py) called the way rdoQuant calls them - once per coefficient group on m_resiDctCoeff / m_fencDctCoeff -
on the transform coefficients the oracle's TU chain captured (pinned against the real Quant::transformNxN,
tests/test_oracle_classes_vs_reference.py); the source block's transform through the table's copy_ps + dct like quant.cpp:436-441."""
import ctypes
import importlib
import os
import sys

import numpy as np
import pytest

import harness as Hn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
QUANT_SCALES = (26214, 23302, 20560, 18396, 16384, 14564)      # scalinglist.cpp:129


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api
    return oracle_api


def expected_prepasses(orc, depth, n, coef, src_blocks, qtable, qp, psy_scale):
    """coef: int16 [blocks, n*n] = m_resiDctCoeff per block; src_blocks: [blocks, n, n] source samples (psy) or None.
    Returns costUncoded int64 [blocks, n*n], cg sums int64 [blocks, n*n/16, 2] ([0]: what nonPsyRdoQuant / _1p add to the totals, [1]: what
    psyRdoQuant / _2p add), nquant levels int16, numSig uint32, fencDct int16."""
    log2n = n.bit_length() - 1
    cu = f"cu[{log2n - 2}]"
    nb, nn, ncg = coef.shape[0], n * n, n * n // 16
    cost = np.zeros((nb, nn), np.int64)
    cg = np.zeros((nb, ncg, 2), np.int64)
    lev = np.zeros((nb, nn), np.int16)
    ns = np.zeros(nb, np.uint32)
    fdct = np.zeros((nb, nn), np.int16)
    per, rem = qp // 6, qp % 6
    tshift = 15 - depth - log2n
    qbits = 14 + per + tshift
    qt = np.ascontiguousarray(qtable if qtable is not None else np.full(nn, QUANT_SCALES[rem], np.int32), dtype=np.int32)
    f_non, f_p1, f_p2, f_nq = orc.fn(f"{cu}.nonPsyRdoQuant"), orc.fn(f"{cu}.psyRdoQuant_1p"), orc.fn(f"{cu}.psyRdoQuant_2p"), orc.fn("nquant")
    f_psy = orc.fn(f"{cu}.psyRdoQuant")
    f_cps, f_dct = orc.fn(f"{cu}.copy_ps"), orc.fn(f"{cu}.dct")
    psy = ctypes.c_int64(psy_scale)
    for b in range(nb):
        c = np.ascontiguousarray(coef[b])
        if psy_scale:
            src = np.ascontiguousarray(src_blocks[b])
            short = np.zeros(nn, np.int16)
            f_cps(short.ctypes.data, n, src.ctypes.data, n)
            f_dct(short.ctypes.data, fdct[b].ctypes.data, n)
            cost2 = np.zeros(nn, np.int64)
        for g in range(ncg):
            blk = (g // (n // 4)) * 4 * n + (g % (n // 4)) * 4
            tu, tr = ctypes.c_int64(0), ctypes.c_int64(0)
            if psy_scale:
                f_p1(c.ctypes.data, cost[b].ctypes.data, ctypes.byref(tu), ctypes.byref(tr), blk)
                assert tu.value == tr.value
                cg[b, g, 0] = tu.value
                tu.value = tr.value = 0           # _2p adds the finished values on top (quant.cpp:716-717 calls both on the same totals)
                f_p2(c.ctypes.data, fdct[b].ctypes.data, cost[b].ctypes.data, ctypes.byref(tu), ctypes.byref(tr), ctypes.byref(psy), blk)
                # the one-call form must agree with the two-pass form
                t2, r2 = ctypes.c_int64(0), ctypes.c_int64(0)
                f_psy(c.ctypes.data, fdct[b].ctypes.data, cost2.ctypes.data, ctypes.byref(t2), ctypes.byref(r2), ctypes.byref(psy), blk)
                assert t2.value == tu.value
            else:
                f_non(c.ctypes.data, cost[b].ctypes.data, ctypes.byref(tu), ctypes.byref(tr), blk)
                cg[b, g, 0] = tu.value
            assert tu.value == tr.value
            cg[b, g, 1] = tu.value
        if psy_scale:
            assert np.array_equal(cost2, cost[b])
        ns[b] = f_nq(c.ctypes.data, qt.ctypes.data, lev[b].ctypes.data, qbits, 1 << (qbits - 1), nn)
    return cost, cg, lev, ns, fdct


def _zorder_xy(z):
    return (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)


@pytest.mark.parametrize("depth,level,qp,psy,lists", [(8, 2, 24, 0, False), (8, 2, 27, 9731, False), (8, 1, 30, 12345, True), (8, 0, 22, 0, False),
                                                      (8, 0, 25, 777, False), (10, 2, 34, 200001, True), (10, 1, 30, 0, False), (12, 1, 40, 3000017, False),
                                                      (12, 2, 44, 0, True)])
def test_inter_stage_emits_the_rdoq_prepasses(depth, level, qp, psy, lists):
    """Luma with / without the psy term (quant.cpp:411: psy-rdoq is luma only), one chroma plane without; scaling lists reach nquant."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([59, depth, level, qp])
    clip = F.synth_clip(192, 128, 2, depth=depth, seed=90 + level)
    cur, ref = P.DevicePicture(clip[1][0], dev, clip[1][1], clip[1][2]), P.DevicePicture(clip[0][0], dev, clip[0][1], clip[0][2])
    ms = P.MotionSearch(cur.w64, cur.h64, 8, depth, dev, want_surf=False)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, 3, dev)
    sp.run(cur, ref)
    torch.cuda.synchronize()
    mv = sp.out.cpu().numpy().reshape(-1, 2)
    O = _oracle()
    orc = Hn.load_oracle(depth, ROOT, host=True)
    ctus_w = cur.w64 // 64
    for chroma in (False, True):
        n = (8 << level) >> (1 if chroma else 0)
        nn = n * n
        psy_here = 0 if chroma else psy
        st = (S.InterReconChroma if chroma else S.InterRecon)(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=2)
        nblocks = st.levels.numel() // nn
        qc = None
        if lists:
            m = rng.integers(8, 64, size=nn)
            qc = ((QUANT_SCALES[qp % 6] << 4) // m).astype(np.int32)
        d = lambda a: None if a is None else torch.from_numpy(a).to(dev)
        d_dct = torch.zeros_like(st.levels)
        d_cost = torch.full((nblocks * nn,), 0x5a5a5a5a5a5a, dtype=torch.int64, device=dev)
        d_cg = torch.full((nblocks * nn // 8,), 0x5a5a5a5a5a5a, dtype=torch.int64, device=dev)
        d_lev = torch.full_like(st.levels, 0x5a5a)
        d_ns = torch.full((nblocks,), 0x5a5a, dtype=torch.int32, device=dev)
        d_fd = torch.zeros_like(st.levels)
        st.tables = A.tu_tables(quant_coeff=d(qc), dct_coeff_out=d_dct, rdoq_cost_uncoded=d_cost, rdoq_cg_cost=d_cg, rdoq_levels=d_lev, rdoq_num_sig=d_ns,
                                fenc_dct_out=d_fd, psy_scale=psy_here)
        e_dct, e_du = np.zeros(st.levels.numel(), np.int16), np.zeros(st.levels.numel(), np.int32)
        O.set_tu_tables(depth, qc, None, None, None)
        O.set_tu_capture(depth, e_dct, e_du)
        try:
            if chroma:
                out = torch.zeros_like(cur.c[0])
                st.run(cur.c[0], ref.c[0], out, cur.stride_c, cur.org_c, sp.out)
                elev = O.inter_recon_chroma(depth, cur.c_host[0].reshape(-1), ref.c_host[0].reshape(-1), cur.stride_c, cur.org_c, cur.w64, cur.h64, level, mv, qp,
                                            intra_slice=2)[1]
                plane, stride, org, ctu_px = cur.c_host[0].reshape(-1), cur.stride_c, cur.org_c, 32
            else:
                recon = torch.zeros_like(cur.t)
                st.run(cur, ref, recon, sp.out)
                elev = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, level, mv, qp, intra_slice=2)[1]
                plane, stride, org, ctu_px = cur.host.reshape(-1), cur.stride, cur.org, 64
        finally:
            O.set_tu_tables(depth)
            O.set_tu_capture(depth)
        torch.cuda.synchronize()
        assert np.array_equal(st.levels.cpu().numpy(), elev), "the stage's own levels changed"
        assert np.array_equal(d_dct.cpu().numpy(), e_dct)
        npu = (ctu_px // n) ** 2
        src = np.zeros((nblocks, n, n), plane.dtype)
        for b in range(nblocks):
            ctu, z = divmod(b, npu)
            bx, by = _zorder_xy(z)
            px, py = (ctu % ctus_w) * ctu_px + bx * n, (ctu // ctus_w) * ctu_px + by * n
            src[b] = np.lib.stride_tricks.as_strided(plane[org + py * stride + px:], (n, n), (stride * plane.itemsize, plane.itemsize))
        cost, cg, lev, ns, fdct = expected_prepasses(orc, depth, n, e_dct.reshape(nblocks, nn), src, qc, qp, psy_here)
        which = f"{'chroma' if chroma else 'luma'} {n}x{n}"
        if psy_here:
            assert np.array_equal(d_fd.cpu().numpy().reshape(nblocks, nn), fdct), which + ": the source block's transform differs"
        assert np.array_equal(d_lev.cpu().numpy().reshape(nblocks, nn), lev), which + ": nquant levels differ"
        assert np.array_equal(d_ns.cpu().numpy().view(np.uint32), ns), which + ": nquant count differs"
        assert np.array_equal(d_cost.cpu().numpy().reshape(nblocks, nn), cost), which + ": costUncoded differs"
        assert np.array_equal(d_cg.cpu().numpy().reshape(nblocks, -1, 2), cg), which + ": coefficient-group sums differ"
        assert ns.sum() > 20 and (cost != 0).sum() > 100
        if psy_here:
            assert (cost < 0).any()                                  # the psy term really bit


@pytest.mark.parametrize("depth,n,qp,psy", [(8, 4, 22, 4099), (8, 4, 27, 0), (8, 8, 27, 12001), (8, 16, 30, 0), (8, 32, 24, 70001), (10, 4, 30, 50021), (10, 32, 33, 0),
                                            (12, 8, 38, 900001), (12, 16, 40, 0)])
def test_intra_stage_emits_the_rdoq_prepasses(depth, n, qp, psy):
    """The intra candidate stage, incl. the 4x4 luma TU whose residual takes the DST while the source block's transform stays the DCT."""
    import torch
    import test_gpu_intra_recon as TI
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([60, depth, n, qp])
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    ntu, modes = 10, (0, 1, 2, 10, 18, 26, 34)
    W = n * ntu
    yy, xx = np.mgrid[0:n, 0:W]
    src = np.clip(np.rint((0.5 + 0.35 * np.sin(xx / 9.0) * np.cos(yy / 5.0)) * pmax + rng.normal(0, 3.0 * (1 << (depth - 8)), (n, W))), 0, pmax).astype(dt)
    fenc_stride = W + 16
    fenc = np.zeros((n, fenc_stride), dtype=dt)
    fenc[:, :W] = src
    nbw = 4 * n + 1
    nb = np.zeros((ntu, 2, nbw + 3), dtype=dt)
    for t in range(ntu):
        base = int(src[:, t * n:(t + 1) * n].mean())
        a = np.clip(base + rng.integers(-12 << (depth - 8), 13 << (depth - 8), nbw), 0, pmax).astype(dt)
        nb[t, 0, :nbw] = a
        nb[t, 1, :nbw] = TI._smooth(a)
    jobs = np.zeros(ntu * len(modes), dtype=A.job_dtype())
    recon_stride = n + 5
    for t in range(ntu):
        for k, m in enumerate(modes):
            j = t * len(modes) + k
            jobs["off"][j] = (t * n, (t * 2) * (nbw + 3), (t * 2 + 1) * (nbw + 3), j * n * recon_stride)
            jobs["arg"][j, 0] = m
    njobs, nn = len(jobs), n * n
    recon_len = njobs * n * recon_stride
    O = _oracle()
    orc = Hn.load_oracle(depth, ROOT, host=True)
    e_dct, e_du = np.zeros(njobs * nn, np.int16), np.zeros(njobs * nn, np.int32)
    O.set_tu_capture(depth, e_dct, e_du)
    try:
        elev = O.intra_recon(depth, n, fenc.reshape(-1), fenc_stride, nb.reshape(-1), recon_len, recon_stride, qp, 3, jobs)[1]
    finally:
        O.set_tu_capture(depth)
    d_dct = torch.zeros(njobs * nn, dtype=torch.int16, device=dev)
    d_cost = torch.full((njobs * nn,), 0x5a5a5a5a5a5a, dtype=torch.int64, device=dev)
    d_cg = torch.full((njobs * nn // 8,), 0x5a5a5a5a5a5a, dtype=torch.int64, device=dev)
    d_nq = torch.full((njobs * nn,), 0x5a5a, dtype=torch.int16, device=dev)
    d_nqs = torch.full((njobs,), 0x5a5a, dtype=torch.int32, device=dev)
    d_fd = torch.zeros(njobs * nn, dtype=torch.int16, device=dev)
    rec = A.tu_tables(dct_coeff_out=d_dct, rdoq_cost_uncoded=d_cost, rdoq_cg_cost=d_cg, rdoq_levels=d_nq, rdoq_num_sig=d_nqs, fenc_dct_out=d_fd, psy_scale=psy)
    d_fenc = torch.from_numpy(fenc.reshape(-1).view(np.uint8)).to(dev)
    d_nb = torch.from_numpy(nb.reshape(-1).view(np.uint8)).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_rec = torch.zeros(recon_len * dt().itemsize, dtype=torch.uint8, device=dev)
    d_lev = torch.zeros(njobs * nn, dtype=torch.int16, device=dev)
    d_ns = torch.zeros(njobs, dtype=torch.int32, device=dev)
    d_dist = torch.zeros(njobs, dtype=torch.int64, device=dev)
    A.intra_recon_batch(depth, n, d_fenc, fenc_stride, d_nb, d_rec, recon_stride, qp, 3, d_jobs, njobs, d_lev, d_ns, d_dist, tables=rec)
    torch.cuda.synchronize()
    assert np.array_equal(d_lev.cpu().numpy(), elev) and np.array_equal(d_dct.cpu().numpy(), e_dct)
    srcb = np.stack([src[:, (j // len(modes)) * n:(j // len(modes) + 1) * n] for j in range(njobs)])
    cost, cg, lev, ns, fdct = expected_prepasses(orc, depth, n, e_dct.reshape(njobs, nn), srcb, None, qp, psy)
    if psy:
        assert np.array_equal(d_fd.cpu().numpy().reshape(njobs, nn), fdct), "the source block's transform differs (always the DCT, quant.cpp:436-441)"
    assert np.array_equal(d_nq.cpu().numpy().reshape(njobs, nn), lev) and np.array_equal(d_nqs.cpu().numpy().view(np.uint32), ns)
    assert np.array_equal(d_cost.cpu().numpy().reshape(njobs, nn), cost), "costUncoded differs"
    assert np.array_equal(d_cg.cpu().numpy().reshape(njobs, -1, 2), cg), "coefficient-group sums differ"
    assert ns.sum() > 20
