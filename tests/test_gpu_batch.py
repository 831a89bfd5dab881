"""GPU parity, BATCH LAYER: the job-list entry points (many blocks per launch, device planes)
vs the oracle called block by block on host copies of the same planes.  Complements the
table-layer test (which drives the same kernels one block at a time through the reference's
signatures): here offsets, strides and per-job arguments vary inside one launch."""
import importlib

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
spec = H.spec


def dev_of(arr):
    import torch
    a = arr.view(np.int16) if arr.dtype == np.uint16 else (arr.view(np.int32) if arr.dtype == np.uint32 else arr)
    return torch.from_numpy(a).to("cuda:0")


def back(t, dtype):
    return t.cpu().numpy().view(dtype)


INTERP = [("hpp", A.IP_HPP), ("hps", A.IP_HPS), ("vpp", A.IP_VPP), ("vps", A.IP_VPS), ("vsp", A.IP_VSP),
          ("vss", A.IP_VSS), ("hvpp", A.IP_HVPP)]


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("luma", [True, False])
def test_interp_batch(depth, luma, repo_root):
    orc = H.load_oracle(depth, repo_root)
    rng = np.random.default_rng([21, depth, int(luma)])
    dt, m = H.pix_dtype(depth), H.pixel_max(depth)
    pus = [2, 4, 14, 19] if luma else [1, 4, 13, 6]          # 16x16, 64x64, 12x16, 32x8  /  4x4, 32x32, 8x6, 2x4 (4:2:0)
    sizes = [(pu, *(spec.pu_dims(pu) if luma else spec.chroma_pu_dims(1, pu))) for pu in pus]
    for name, kind in INTERP:
        if not luma and name == "hvpp":
            continue
        src_short = name in ("vsp", "vss")
        dst_short = name in ("hps", "vps", "vss")
        for pu, w, h in sizes:
            field = (f"pu[{pu}].luma_{name}" if luma else f"chroma[1].pu[{pu}].filter_{name}")
            fn = orc.fn(field)
            sst, rows = 300, 200
            src = (rng.integers(-8192, 8192, size=sst * rows).astype(np.int16) if src_short
                   else rng.integers(0, m + 1, size=sst * rows).astype(dt))
            njobs = 23
            dpitch = w * (h + 8) + 16
            ddt = np.int16 if dst_short else dt
            dst0 = np.full(njobs * dpitch + 64, 7, dtype=ddt)
            exp = dst0.copy()
            jobs = []
            for j in range(njobs):
                so = int(rng.integers(8, rows - h - 16)) * sst + int(rng.integers(8, sst - w - 16))
                do = j * dpitch + 5
                idx = int(rng.integers(0, 4 if luma else 8))
                a1 = int(rng.integers(0, 2)) if name == "hps" else (int(rng.integers(0, 4)) if name == "hvpp" else 0)
                jobs.append(([so, do], [idx, a1]))
                args = (idx, a1) if name in ("hps", "hvpp") else (idx,)
                fn(H.ptr(src, so), sst, H.ptr(exp, do), w, *args)
            ts, td = dev_of(src), dev_of(dst0)
            A.interp_batch(kind, depth, 8 if luma else 4, w, h, A.plane(ts, sst), A.plane(td, w), A.make_jobs(jobs, "cuda:0"), njobs)
            got = back(td, ddt)
            assert np.array_equal(got, exp), f"{field} depth {depth}: {np.count_nonzero(got != exp)} samples differ"


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("use_mfma", [0, 1])
def test_transform_batch(depth, use_mfma, repo_root):
    orc = H.load_oracle(depth, repo_root)
    rng = np.random.default_rng([22, depth, use_mfma])
    m = H.pixel_max(depth)
    cases = [("dct", A.TR_DCT, n) for n in (4, 8, 16, 32)] + [("idct", A.TR_IDCT, n) for n in (4, 8, 16, 32)]
    cases += [("dst4x4", A.TR_DST4, 4), ("idst4x4", A.TR_IDST4, 4)] + [("lowpass_dct", A.TR_LOWPASS_DCT, n) for n in (8, 16, 32)]
    for name, kind, n in cases:
        field = name if name.endswith("4x4") else f"cu[{spec.LUMA_CU.index(n)}].{name}"
        fn = orc.fn(field)
        inverse = kind in (A.TR_IDCT, A.TR_IDST4)
        njobs = 29
        st = n + 11
        if inverse:
            src = rng.integers(-32768, 32768, size=njobs * n * n).astype(np.int16)
            src[:n * n] = 32767; src[n * n:2 * n * n] = -32768          # saturating extremes
            dst0 = np.full(njobs * st * n + 64, 3, dtype=np.int16)
            exp = dst0.copy()
            jobs = []
            for j in range(njobs):
                jobs.append(([j * n * n, j * st * n + 2], []))
                fn(H.ptr(src, j * n * n), H.ptr(exp, j * st * n + 2), st)
            ts, td = dev_of(src), dev_of(dst0)
            A.transform_batch(kind, depth, n, A.plane(ts, n), A.plane(td, st), A.make_jobs(jobs, "cuda:0"), njobs, use_mfma)
        else:
            src = rng.integers(-m, m + 1, size=njobs * st * n + 64).astype(np.int16)
            src[:st * n] = m; src[st * n:2 * st * n] = -m                # TestBench all-max / all-min residuals
            dst0 = np.full(njobs * n * n, 3, dtype=np.int16)
            exp = dst0.copy()
            jobs = []
            for j in range(njobs):
                jobs.append(([j * st * n + 1, j * n * n], []))
                fn(H.ptr(src, j * st * n + 1), H.ptr(exp, j * n * n), st)
            ts, td = dev_of(src), dev_of(dst0)
            A.transform_batch(kind, depth, n, A.plane(ts, st), A.plane(td, n), A.make_jobs(jobs, "cuda:0"), njobs, use_mfma)
        got = back(td, np.int16)
        assert np.array_equal(got, exp), f"{field} depth {depth} mfma={use_mfma}: {np.count_nonzero(got != exp)} coefficients differ"


@pytest.mark.parametrize("depth", [8, 10])
def test_intra_batch_all_modes(depth, repo_root):
    orc = H.load_oracle(depth, repo_root)
    rng = np.random.default_rng([23, depth])
    dt, m = H.pix_dtype(depth), H.pixel_max(depth)
    for ci, n in enumerate((4, 8, 16, 32)):
        nb_pitch = 4 * 32 + 16
        ntu = 3
        nb = rng.integers(0, m + 1, size=ntu * nb_pitch).astype(dt)
        nb[:nb_pitch] = m                                               # one TU with flat-max neighbours
        st = n + 5
        jobs, njobs = [], ntu * 35 * 2
        dst0 = np.full(njobs * st * n + 64, 1, dtype=dt)
        exp = dst0.copy()
        j = 0
        for t in range(ntu):
            for mode in range(35):
                for bf in (0, 1):
                    do = j * st * n + 3
                    jobs.append(([t * nb_pitch + 2, do], [mode, bf]))
                    orc.fn(f"cu[{ci}].intra_pred[{mode}]")(H.ptr(exp, do), st, H.ptr(nb, t * nb_pitch + 2), mode, bf)
                    j += 1
        tn, td = dev_of(nb), dev_of(dst0)
        A.intra_batch(A.INTRA_PRED, depth, n, A.plane(tn, 0), A.plane(td, st), A.make_jobs(jobs, "cuda:0"), njobs)
        got = back(td, dt)
        assert np.array_equal(got, exp), f"intra {n}x{n} depth {depth}: {np.count_nonzero(got != exp)} samples differ"


def test_quant_batch(repo_root):
    orc = H.load_oracle(8, repo_root)
    rng = np.random.default_rng(24)
    njobs, n = 17, 1024
    coef = rng.integers(-255, 256, size=njobs * n).astype(np.int16)
    qc = rng.integers(1, 256, size=njobs * n).astype(np.int32)
    du0 = np.zeros(njobs * n, dtype=np.int32)
    q0 = np.zeros(njobs * n, dtype=np.int16)
    edu, eq = du0.copy(), q0.copy()
    jobs, eret = [], []
    for j in range(njobs):
        num = int(rng.choice([16, 64, 256, 1024]))
        bits = int(rng.integers(9, 22))
        add = int((171 if j & 1 else 85) << (bits - 9))
        o = j * n
        jobs.append(([o, o, o, o], [bits, add, num]))
        eret.append(orc.fn("quant")(H.ptr(coef, o), H.ptr(qc, o), H.ptr(edu, o), H.ptr(eq, o), bits, add, num))
    import torch
    tc, tq, tdu, tqo = dev_of(coef), dev_of(qc), dev_of(du0), dev_of(q0)
    res = torch.zeros(njobs, dtype=torch.int32, device="cuda:0")
    A.quant_batch(A.Q_QUANT, [A.plane(tc), A.plane(tq), A.plane(tdu), A.plane(tqo)], A.make_jobs(jobs, "cuda:0"), njobs, res)
    assert res.cpu().tolist() == eret
    assert np.array_equal(back(tdu, np.int32), edu) and np.array_equal(back(tqo, np.int16), eq)


def test_blockop_batch_addavg_and_var(repo_root):
    orc = H.load_oracle(10, repo_root)
    rng = np.random.default_rng(25)
    import torch
    w = h = 32
    njobs, st = 19, 40
    a = rng.integers(-8192, 8192, size=njobs * st * h).astype(np.int16)
    b = rng.integers(-8192, 8192, size=njobs * st * h).astype(np.int16)
    d0 = np.zeros(njobs * st * h, dtype=np.uint16)
    exp = d0.copy()
    jobs = []
    for j in range(njobs):
        o = j * st * h
        jobs.append(([o, o, o], []))
        orc.fn("pu[3].addAvg[0]")(H.ptr(a, o), H.ptr(b, o), H.ptr(exp, o), st, st, st)
    ta, tb, td = dev_of(a), dev_of(b), dev_of(d0)
    A.blockop_batch(A.OP_ADDAVG, 10, w, h, [A.plane(td, st), A.plane(ta, st), A.plane(tb, st)], A.make_jobs(jobs, "cuda:0"), njobs)
    assert np.array_equal(back(td, np.uint16), exp)
    res = torch.zeros(njobs, dtype=torch.int64, device="cuda:0")
    A.blockop_batch(A.OP_VAR, 10, w, h, [A.plane(td, st), None, None], A.make_jobs(jobs, "cuda:0"), njobs, res)
    ev = [orc.fn("cu[3].var")(H.ptr(exp, j * st * h), st) for j in range(njobs)]
    assert back(res, np.uint64).tolist() == ev
