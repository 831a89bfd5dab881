"""GPU parity: the intra TU candidate set (x265hip_intra_recon_batch) vs the oracle's restatement of Search::codeIntraLumaQT's
pixel work (search.cpp:335-373: predIntraLumaAng, calcresidual, transformNxN / DST-VII for 4x4, invtransformNxN, add_ps / copy_pp,
sse_pp) driven through the oracle primitives."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _smooth(a):
    """[1 2 1] smoothing along the left-bottom .. corner .. above-right path (any filtered contents exercise the stage the same
    way; the filter primitive has its own parity test)."""
    n4 = len(a) - 1
    n2 = n4 // 2
    path = np.concatenate([a[n4:n2:-1], a[0:1], a[1:n2 + 1]]).astype(np.int32)
    f = path.copy()
    f[1:-1] = (path[:-2] + 2 * path[1:-1] + path[2:] + 2) >> 2
    out = a.copy()
    out[n4:n2:-1] = f[:n2]
    out[0] = f[n2]
    out[1:n2 + 1] = f[n2 + 1:]
    return out


@pytest.mark.parametrize("depth,n,qp,islice", [(8, 4, 22, 1), (8, 8, 27, 1), (8, 16, 32, 0), (8, 32, 22, 1), (8, 32, 44, 0),
                                               (10, 4, 30, 0), (10, 8, 12, 1), (10, 16, 24, 1), (10, 32, 37, 1), (8, 16, 0, 1),
                                               (12, 4, 40, 1), (12, 16, 33, 0), (12, 32, 50, 1),
                                               # islice >= 2: X265HIP_TU_SIGN_HIDE (Quant::signBitHidingHDQ after the quantiser; all 35 modes, so
                                               # the horizontal / vertical scans of the 4x4 and luma 8x8 TUs are walked as well)
                                               (8, 4, 22, 3), (8, 8, 27, 2), (8, 16, 30, 3), (8, 32, 22, 2), (8, 32, 36, 3), (10, 4, 30, 2), (10, 8, 24, 3),
                                               (10, 32, 30, 2), (12, 16, 33, 3)])
@pytest.mark.parametrize("chroma", [False, True])
def test_intra_recon_matches_oracle(depth, n, qp, islice, chroma):
    _run(depth, n, qp, islice, chroma, False)


QUANT_SCALES, INV_QUANT_SCALES = (26214, 23302, 20560, 18396, 16384, 14564), (40, 45, 51, 57, 64, 72)      # scalinglist.cpp:129-130


@pytest.mark.parametrize("depth,n,qp,islice", [(8, 4, 22, 3), (8, 8, 27, 2), (8, 16, 30, 1), (8, 32, 24, 3), (10, 8, 30, 2), (10, 32, 33, 3), (12, 16, 40, 2)])
@pytest.mark.parametrize("chroma", [False, True])
def test_intra_recon_with_scaling_lists_and_denoiser(depth, n, qp, islice, chroma):
    """x265hip_tu_tables in the intra candidate stage (scaling-list coefficient tables + the denoiser's offsets and running sums)."""
    _run(depth, n, qp, islice, chroma, True)


def _run(depth, n, qp, islice, chroma, tabs):
    """chroma: the 4:2:0 chroma flavour (predIntraChromaAng: unfiltered neighbours, bFilter 0; DCT for 4x4 - the oracle side of both
    flavours is pinned against the real Quant / the table primitive in tests/test_oracle_classes_vs_reference.py)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng([61, depth, n, qp])
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    ntu = 24
    # source: smooth texture + noise so that some candidates quantise to nothing, some to DC only, most to many levels
    W = n * ntu
    yy, xx = np.mgrid[0:n, 0:W]
    src = np.clip(np.rint((0.5 + 0.35 * np.sin(xx / 9.0) * np.cos(yy / 5.0)) * pmax + rng.normal(0, 3.0 * (1 << (depth - 8)), (n, W))), 0, pmax).astype(dt)
    src[:, :n] = src[0, 0]                                      # a flat TU: DC / empty residuals
    fenc_stride = W + 16
    fenc = np.zeros((n, fenc_stride), dtype=dt)
    fenc[:, :W] = src
    nbw = 4 * n + 1
    nb = np.zeros((ntu, 2, nbw + 3), dtype=dt)                  # odd record size: unaligned neighbour rows
    for t in range(ntu):
        base = int(src[:, t * n:(t + 1) * n].mean())
        a = np.clip(base + rng.integers(-12 << (depth - 8), 13 << (depth - 8), nbw), 0, pmax).astype(dt)
        if t == 0:
            a[:] = src[0, 0]
        nb[t, 0, :nbw] = a
        nb[t, 1, :nbw] = _smooth(a)
    jobs = np.zeros(ntu * 35, dtype=H.job_dtype())
    recon_stride = n + 5
    for t in range(ntu):
        for m in range(35):
            j = t * 35 + m
            jobs["off"][j] = (t * n, (t * 2) * (nbw + 3), (t * 2 + 1) * (nbw + 3), j * n * recon_stride)
            jobs["arg"][j, 0] = m
    njobs = len(jobs)
    recon_len = njobs * n * recon_stride
    O = _oracle()
    rec = d_sum = osum = None
    if tabs:
        m = rng.integers(8, 64, size=n * n)
        qc, dqc = ((QUANT_SCALES[qp % 6] << 4) // m).astype(np.int32), (INV_QUANT_SCALES[qp % 6] * m).astype(np.int32)
        off = rng.integers(0, 5 << (depth - 8), size=n * n).astype(np.uint16)
        osum = np.zeros(n * n, np.uint32)
        d_sum = torch.zeros(n * n, dtype=torch.int32, device=dev)
        d_cap = (torch.zeros(njobs * n * n, dtype=torch.int16, device=dev), torch.zeros(njobs * n * n, dtype=torch.int32, device=dev))
        e_cap = (np.zeros(njobs * n * n, np.int16), np.zeros(njobs * n * n, np.int32))
        rec = H.tu_tables(torch.from_numpy(qc).to(dev), torch.from_numpy(dqc).to(dev), torch.from_numpy(off.view(np.int16)).to(dev), d_sum, *d_cap)
        O.set_tu_tables(depth, qc, dqc, off, osum)
        O.set_tu_capture(depth, *e_cap)
    try:
        erec, elev, ens, edist = O.intra_recon(depth, n, fenc.reshape(-1), fenc_stride, nb.reshape(-1), recon_len, recon_stride, qp, islice, jobs, chroma=chroma)
    finally:
        O.set_tu_tables(depth)
        O.set_tu_capture(depth)

    d_fenc = torch.from_numpy(fenc.reshape(-1).view(np.uint8)).to(dev)
    d_nb = torch.from_numpy(nb.reshape(-1).view(np.uint8)).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_rec = torch.zeros(recon_len * dt().itemsize, dtype=torch.uint8, device=dev)
    d_lev = torch.full((njobs * n * n,), 0x5a5a, dtype=torch.int16, device=dev)
    d_ns = torch.zeros(njobs, dtype=torch.int32, device=dev)
    d_dist = torch.zeros(njobs, dtype=torch.int64, device=dev)
    H.intra_recon_batch(depth, n, d_fenc, fenc_stride, d_nb, d_rec, recon_stride, qp, islice, d_jobs, njobs, d_lev, d_ns, d_dist, chroma=chroma, tables=rec)
    torch.cuda.synchronize()
    if tabs:
        assert np.array_equal(d_sum.cpu().numpy().view(np.uint32), osum) and osum.sum() > 0, "denoiser residual sums differ"
        assert np.array_equal(d_cap[0].cpu().numpy(), e_cap[0]), "captured transform coefficients differ"
        assert np.array_equal(d_cap[1].cpu().numpy(), e_cap[1]), "captured deltaU differs"
        flat = O.intra_recon(depth, n, fenc.reshape(-1), fenc_stride, nb.reshape(-1), recon_len, recon_stride, qp, islice, jobs, chroma=chroma)[1]
        assert not np.array_equal(flat, elev), "the tables changed nothing"
    assert np.array_equal(d_ns.cpu().numpy().view(np.uint32), ens), "numSig differs"
    assert np.array_equal(d_lev.cpu().numpy(), elev), "quantised levels differ"
    grec = d_rec.cpu().numpy().view(dt)
    # only the n x n block of each candidate is defined (the row padding of recon_stride stays untouched on both sides)
    assert np.array_equal(grec, erec), f"recon differs at {np.count_nonzero(grec != erec)} samples"
    assert np.array_equal(d_dist.cpu().numpy().view(np.uint64), edist), "SSE differs"
    # every branch of the inverse path must be hit somewhere in the case set
    if qp in (22, 27, 12):
        assert (ens > 1).any()
    if qp == 44:
        assert (ens == 0).any()
    if islice & 2:                                              # sign hiding really changed levels
        plain = O.intra_recon(depth, n, fenc.reshape(-1), fenc_stride, nb.reshape(-1), recon_len, recon_stride, qp, islice & 1, jobs, chroma=chroma)[1]
        assert np.count_nonzero(plain != elev) > 0
    if chroma and n <= 16:                                      # the two flavours really differ (edge smoothing / filtered neighbours / DST)
        other = O.intra_recon(depth, n, fenc.reshape(-1), fenc_stride, nb.reshape(-1), recon_len, recon_stride, qp, islice, jobs)[0]
        assert not np.array_equal(other, erec)
