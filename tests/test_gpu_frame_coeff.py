"""GPU parity, BATCH LAYER of the last two slot families (csrc/frame_coeff_kernels.hip): thousands of calls of
scanPosLast / findPosFirstLast / costCoeffNxN / costCoeffRemain / costC1C2Flag / the uncoded-cost pre-passes per launch, whole planes through
planecopy_* / planeClipAndMax / frameInitLowres, whole rows of SSIM windows and cutree rows - each against the oracle's slot called job by job
on host copies (oracle/x265_oracle_host.c, itself pinned against the reference build in tests/test_oracle_vs_reference.py).  Input recipes
follow the reference's harnesses (pixelharness.cpp:1705-2080) like tests/harness_host.py."""
import ctypes
import importlib

import numpy as np
import pytest

import harness as H
import harness_host as HH

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
ptr = H.ptr


def dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to("cuda:0")


def back(t, dtype):
    return t.cpu().numpy().view(dtype)


@pytest.fixture(scope="module")
def tables(repo_root):
    t = H.host_tables(repo_root)
    A.set_entropy_bits(t["entropy_state_bits"])           # either form is accepted: the top byte is ignored
    return t


def oracle(depth, repo_root):
    return H.load_oracle(depth, repo_root, host=True)


def test_scan_pos_last_batch(repo_root, tables):
    orc = oracle(8, repo_root)
    f = orc.fn("scanPosLast")
    rng = np.random.default_rng(901)
    scans, coeffs, jobs, want = [], [], [], []
    so = co = 0
    for j in range(700):
        tr = 4 << int(rng.integers(0, 4))
        scan, inner = HH.block_scan(rng, tr, int(rng.integers(0, 3)))
        coeff = HH.sparse_coeffs(rng, "random", tr * tr)
        if j % 7 == 0:
            coeff[:] = 0                                  # one non-zero anywhere, first / last position included
            coeff[int(scan[[0, -1, int(rng.integers(0, tr * tr))][j % 3]])] = -5
        if not np.any(coeff):
            coeff[int(rng.integers(0, tr * tr))] = 9
        nsig = int(np.count_nonzero(coeff))
        if j % 11 == 0 and nsig > 1:
            nsig = int(rng.integers(1, nsig))             # the walk stops at the numSig-th non-zero, not at the block's last
        sign, flag, num = np.zeros(64, np.uint16), np.zeros(64, np.uint16), np.zeros(64, np.uint8)
        r = f(ptr(scan), ptr(coeff), ptr(sign), ptr(flag), ptr(num), nsig, ptr(inner), tr)
        want.append((r, sign, flag, num))
        jobs.append(([so, co, 64 * j, 64 * j, 64 * j], [nsig, tr]))
        scans.append(scan); coeffs.append(coeff)
        so += scan.size; co += coeff.size
    n = len(jobs)
    import torch
    d_scan, d_coeff = dev(np.concatenate(scans)), dev(np.concatenate(coeffs))
    d_sign, d_flag = torch.full((n * 64,), 0x4d4d, dtype=torch.int16, device="cuda:0"), torch.full((n * 64,), 0x4d4d, dtype=torch.int16, device="cuda:0")
    d_num = torch.full((n * 64,), 0x4d, dtype=torch.uint8, device="cuda:0")
    res = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    A.coeff_batch(A.CF_SCAN_POS_LAST, 8, [d_scan, d_coeff, d_sign, d_flag, d_num], A.make_coeff_jobs(jobs, "cuda:0"), n, res)
    torch.cuda.synchronize()
    got_r, gs, gf, gn = res.cpu().numpy(), back(d_sign, np.uint16).reshape(n, 64), back(d_flag, np.uint16).reshape(n, 64), back(d_num, np.uint8).reshape(n, 64)
    for j, (r, sign, flag, num) in enumerate(want):
        assert got_r[j] == r and np.array_equal(gs[j], sign) and np.array_equal(gf[j], flag) and np.array_equal(gn[j], num), f"scanPosLast job {j}"


def test_cabac_estimators_batch(repo_root, tables):
    """costCoeffNxN + costC1C2Flag + costCoeffRemain + findPosFirstLast: 3000 coefficient groups each, every job with its own context bytes."""
    import torch
    orc = oracle(8, repo_root)
    rng = np.random.default_rng(902)
    n = 3000
    # ---- costCoeffNxN
    f = orc.fn("costCoeffNxN")
    scan_b, coeff_b, tab_b = np.zeros((n, 16), np.uint16), np.zeros((n, 16), np.int16), np.zeros((n, 16), np.uint8)
    ctx0 = HH._ctx_states(rng, n * 64).reshape(n, 64)
    want_bits, want_ctx, want_abs, jobs = [], ctx0.copy(), np.full((n, 24), 0x4d4d, np.uint16), []
    for j in range(n):
        size_idx = int(rng.integers(0, 4))
        offset = 0 if size_idx == 0 else (9 if size_idx == 1 else 12)
        inner = HH.diag_scan(4) if j % 3 else rng.permutation(16).astype(np.uint16)
        tab = rng.integers(0, 9, size=16).astype(np.uint8)
        coeff = HH.sparse_coeffs(rng, "random", 16)
        if j % 4 == 0:                                    # dense groups: every visited position writes a level (the last entry of absCoeff included)
            coeff = rng.integers(-300, 301, size=16).astype(np.int16)
        off = int(rng.integers(0, 16))
        sub_base = 0 if j % 5 == 0 else 16 * int(rng.integers(1, 4))
        mask = nsig = 0
        for k in range(off + 1):
            c = int(coeff[int(inner[k])])
            mask = mask * 2 + (c != 0); nsig += c != 0
        if nsig == 0:
            coeff[int(inner[off])] = -2; mask |= 1
        nnz0 = 1 if off < 15 else 0
        ctx = want_ctx[j]
        want_bits.append(f(ptr(inner), ptr(coeff), 4, ptr(want_abs[j], 2 + nnz0), ptr(tab), mask, ptr(ctx), offset, off, sub_base))
        scan_b[j], coeff_b[j], tab_b[j] = inner, coeff, tab
        jobs.append(([16 * j, 16 * j, 24 * j + 2 + nnz0, 16 * j, 64 * j], [4, mask, offset, off, sub_base]))
    d_abs = torch.full((n * 24,), 0x4d4d, dtype=torch.int16, device="cuda:0")
    d_ctx = dev(ctx0)
    res = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    A.coeff_batch(A.CF_COST_COEFF_NXN, 8, [dev(scan_b), dev(coeff_b), d_abs, dev(tab_b), d_ctx], A.make_coeff_jobs(jobs, "cuda:0"), n, res)
    torch.cuda.synchronize()
    assert np.array_equal(back(res, np.uint32), np.array(want_bits, np.uint32))
    assert np.array_equal(back(d_ctx, np.uint8).reshape(n, 64), want_ctx)
    assert np.array_equal(back(d_abs, np.uint16).reshape(n, 24), want_abs)
    # ---- costC1C2Flag
    f = orc.fn("costC1C2Flag")
    absb, ctx0 = np.zeros((n, 16), np.uint16), HH._ctx_states(rng, n * 32).reshape(n, 32)
    want, want_ctx, jobs = [], ctx0.copy(), []
    for j in range(n):
        k = int(rng.integers(1, 9))
        v = rng.integers(1, 32768, size=k)
        v = np.where(v < 32767 // 3, 1, np.where(v < 32767 // 2, 2, np.where(v < 32767 * 3 // 4, 3, v)))
        absb[j, :k] = v
        off = int(rng.integers(4, 28))
        want.append(f(ptr(absb[j]), k, ptr(want_ctx[j]), off))
        jobs.append(([0, 0, 16 * j, 0, 32 * j], [k, off]))
    d_ctx = dev(ctx0)
    A.coeff_batch(A.CF_COST_C1C2, 8, [None, None, dev(absb), None, d_ctx], A.make_coeff_jobs(jobs, "cuda:0"), n, res)
    torch.cuda.synchronize()
    assert np.array_equal(back(res, np.uint32), np.array(want, np.uint32))
    assert np.array_equal(back(d_ctx, np.uint8).reshape(n, 32), want_ctx)
    # ---- costCoeffRemain
    f = orc.fn("costCoeffRemain")
    absb = rng.integers(0, 32768, size=(n, 24))
    absb[absb < 32767 * 2 // 3] = rng.integers(1, 40, size=int((absb < 32767 * 2 // 3).sum()))
    absb = absb.astype(np.uint16)
    want, jobs = [], []
    for j in range(n):
        nnz = int(rng.integers(0, 17))
        first = int(rng.integers(0, 9))
        want.append(f(ptr(absb[j]), nnz, first))
        jobs.append(([0, 0, 24 * j], [nnz, first]))
    A.coeff_batch(A.CF_COST_COEFF_REMAIN, 8, [None, None, dev(absb), None, None], A.make_coeff_jobs(jobs, "cuda:0"), n, res)
    torch.cuda.synchronize()
    assert np.array_equal(back(res, np.uint32), np.array(want, np.uint32))
    # ---- findPosFirstLast: groups inside 32x32 blocks
    f = orc.fn("findPosFirstLast")
    blocks = np.stack([HH.sparse_coeffs(rng, "random", 1024) for _ in range(64)])
    tbls = np.stack([HH.diag_scan(4), np.arange(16, dtype=np.uint16), rng.permutation(16).astype(np.uint16)])
    want, jobs = [], []
    for j in range(n):
        b, g, t = int(rng.integers(0, 64)), int(rng.integers(0, 64)), j % 3
        org = (g // 8) * 4 * 32 + (g % 8) * 4
        if not any(blocks[b][org + (int(q) >> 2) * 32 + (int(q) & 3)] for q in range(16)):
            blocks[b][org + 5] = 3
        jobs.append(([16 * t, 1024 * b + org], [32]))
    for (offs, _), j in zip(jobs, range(n)):
        b, org = divmod(offs[1], 1024)
        want.append(f(ptr(blocks[b], org), 32, ptr(tbls[offs[0] // 16])))
    A.coeff_batch(A.CF_FIND_POS_FIRST_LAST, 8, [dev(tbls), dev(blocks), None, None, None], A.make_coeff_jobs(jobs, "cuda:0"), n, res)
    torch.cuda.synchronize()
    assert np.array_equal(back(res, np.uint32), np.array(want, np.uint32))


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_rdoq_uncoded_cost_batch(depth, repo_root, tables):
    """Every coefficient group of 40 TUs per size through the four pre-pass slots, in place on the blocks (row stride = the TU's)."""
    import torch
    orc = oracle(depth, repo_root)
    rng = np.random.default_rng([903, depth])
    for kind, name, has_fenc in ((A.CF_RDOQ_NONPSY, "nonPsyRdoQuant", False), (A.CF_RDOQ_PSY, "psyRdoQuant", True),
                                 (A.CF_RDOQ_PSY_1P, "psyRdoQuant_1p", False), (A.CF_RDOQ_PSY_2P, "psyRdoQuant_2p", True)):
        for log2 in (2, 3, 4, 5):
            f = orc.fn(f"cu[{log2 - 2}].{name}")
            tr, nb = 1 << log2, 40
            resi = rng.integers(-32768, 32768, size=(nb, tr * tr)).astype(np.int16)
            fenc = rng.integers(-32768, 32768, size=(nb, tr * tr)).astype(np.int16)
            cost0 = rng.integers(-(1 << 40), 1 << 40, size=(nb, tr * tr)).astype(np.int64)
            psy = rng.integers(0, 1 << 16, size=nb).astype(np.int64)
            groups = [(b, (g // (tr // 4)) * 4 * tr + (g % (tr // 4)) * 4) for b in range(nb) for g in range((tr // 4) ** 2)]
            tot0 = rng.integers(0, 1 << 40, size=(len(groups), 2)).astype(np.int64)
            want_cost, want_tot, jobs = cost0.copy(), tot0.copy(), []
            for k, (b, blk) in enumerate(groups):
                if has_fenc:
                    f(ptr(resi[b]), ptr(fenc[b]), ptr(want_cost[b]), ptr(want_tot[k], 0), ptr(want_tot[k], 1), ptr(psy, b), blk)
                else:
                    f(ptr(resi[b]), ptr(want_cost[b]), ptr(want_tot[k], 0), ptr(want_tot[k], 1), blk)
                jobs.append(([b * tr * tr, b * tr * tr, b * tr * tr, 2 * k, b], [blk, log2]))
            d_cost, d_tot = dev(cost0), dev(tot0)
            A.coeff_batch(kind, depth, [dev(fenc), dev(resi), d_cost, d_tot, dev(psy)], A.make_coeff_jobs(jobs, "cuda:0"), len(jobs))
            torch.cuda.synchronize()
            assert np.array_equal(back(d_cost, np.int64).reshape(nb, -1), want_cost), (name, log2)
            assert np.array_equal(back(d_tot, np.int64).reshape(-1, 2), want_tot), (name, log2)


@pytest.mark.parametrize("depth", [8, 10])
def test_whole_plane_copies(depth, repo_root):
    """planecopy_cp / _sp / _sp_shl / _pp_shr and planeClipAndMax on 1080p-sized planes with odd widths, two planes per launch."""
    import torch
    orc = oracle(depth, repo_root)
    rng = np.random.default_rng([904, depth])
    pd, pm = H.pix_dtype(depth), H.pixel_max(depth)
    w, h, ss, ds = 1913, 1080, 1984, 1936
    for kind, name, sdt, args in ((A.FR_PLANECOPY_CP, "planecopy_cp", np.uint8, (depth - 8,)), (A.FR_PLANECOPY_SP, "planecopy_sp", np.uint16, (16 - depth, pm)),
                                  (A.FR_PLANECOPY_SP_SHL, "planecopy_sp_shl", np.uint16, (2, pm)), (A.FR_PLANECOPY_PP_SHR, "planecopy_pp_shr", pd, (depth - 8 + 1,))):
        hi = 255 if sdt == np.uint8 else (65535 if sdt == np.uint16 and "sp" in name else pm)
        src = rng.integers(0, hi + 1, size=2 * ss * h).astype(sdt)
        want = np.full(2 * ds * h, 3, pd)
        for k in range(2):
            orc.fn(name)(ptr(src, k * ss * h), ss, ptr(want, k * ds * h), ds, w, h, *args)
        d_dst = dev(np.full(2 * ds * h, 3, pd))
        d_src = dev(src)
        jobs = A.make_jobs([([k * ss * h, k * ds * h], list(args)) for k in range(2)], "cuda:0")
        A.frame_batch(kind, depth, w, h, [A.Plane(d_src.data_ptr(), ss), A.Plane(d_dst.data_ptr(), ds)], jobs, 2)
        torch.cuda.synchronize()
        assert np.array_equal(back(d_dst, pd), want), name
    if depth > 8:
        src = rng.integers(0, pm + 1, size=2 * ss * h).astype(pd)
        want, res = src.copy(), []
        tot = np.zeros(1, np.uint64)
        for k in range(2):
            res.append((int(orc.fn("planeClipAndMax")(ptr(want, k * ss * h), ss, w, h, ptr(tot), 64 + k, 940 - 7 * k)), int(tot[0])))
        d = dev(src)
        out = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        jobs = A.make_jobs([([k * ss * h], [64 + k, 940 - 7 * k]) for k in range(2)], "cuda:0")
        A.frame_batch(A.FR_PLANE_CLIP_MAX, depth, w, h, [A.Plane(d.data_ptr(), ss), A.Plane(d.data_ptr(), ss)], jobs, 2, out)
        torch.cuda.synchronize()
        assert np.array_equal(back(d, pd), want)
        assert [tuple(r) for r in out.cpu().numpy().reshape(2, 2).tolist()] == res


@pytest.mark.parametrize("depth", [8, 10])
def test_ssim_rows_lowres_and_cutree_rows(depth, repo_root):
    """The SSIM of a picture the way the reference walks it (encoder/framefilter.cpp calculateSSIM: rows of 4x4x2 moments, ssim_end_4 over
    groups of four windows), frameInitLowres of an odd-sized plane, propagateCost / fix8 rows."""
    import torch
    orc = oracle(depth, repo_root)
    rng = np.random.default_rng([905, depth])
    pd, pm = H.pix_dtype(depth), H.pixel_max(depth)
    w, h, st = 328, 64, 352
    a = rng.integers(0, pm + 1, size=st * h).astype(pd)
    b = np.clip(a.astype(np.int32) + rng.integers(-12, 13, size=a.size), 0, pm).astype(pd)
    bw, bh = w // 4, h // 4
    pairs = [(y, x) for y in range(bh) for x in range(0, bw - 1, 2)]
    want = np.zeros((len(pairs), 8), np.int32)
    for k, (y, x) in enumerate(pairs):
        orc.fn("ssim_4x4x2_core")(ptr(a, 4 * y * st + 4 * x), st, ptr(b, 4 * y * st + 4 * x), st, ptr(want[k]))
    d_a, d_b = dev(a), dev(b)
    out = torch.zeros(len(pairs) * 8, dtype=torch.int32, device="cuda:0")
    jobs = A.make_jobs([([4 * y * st + 4 * x, 4 * y * st + 4 * x], []) for y, x in pairs], "cuda:0")
    A.frame_batch(A.FR_SSIM_CORE, depth, 8, 4, [A.Plane(d_a.data_ptr(), st), A.Plane(d_b.data_ptr(), st)], jobs, len(pairs), out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(-1, 8), want)
    # ssim_end_4 over the rows of moments: sums[y] = [bw][4]
    per_row = (bw // 2) * 2
    sums = want.reshape(bh, per_row, 4)
    ends = [(y, x, min(4, per_row - x - 1)) for y in range(bh - 1) for x in range(0, per_row - 1, 4)]
    wf = np.array([orc.fn("ssim_end_4")(ptr(sums[y], 4 * x), ptr(sums[y + 1], 4 * x), wd) for y, x, wd in ends], np.float32)
    pad = np.concatenate([sums.reshape(-1), np.zeros(32, np.int32)])        # the last group's [5][4] window reads past a row's end
    d_s = dev(pad)
    outf = torch.zeros(len(ends), dtype=torch.float32, device="cuda:0")
    jobs = A.make_jobs([([(y * per_row + x) * 4, ((y + 1) * per_row + x) * 4], [wd]) for y, x, wd in ends], "cuda:0")
    A.frame_batch(A.FR_SSIM_END4, depth, 0, 0, [A.Plane(d_s.data_ptr(), 0), A.Plane(d_s.data_ptr(), 0)], jobs, len(ends), outf)
    torch.cuda.synchronize()
    assert outf.cpu().numpy().tobytes() == wf.tobytes()
    # frameInitLowres, odd size
    lw, lh, ss, ds = 161, 37, 2 * 161 + 9, 173
    src = rng.integers(0, pm + 1, size=ss * (2 * lh + 2)).astype(pd)
    wd4 = [np.full(ds * lh, 7, pd) for _ in range(4)]
    orc.fn("frameInitLowres")(ptr(src), *[ptr(x) for x in wd4], ss, ds, lw, lh)
    d4 = [dev(np.full(ds * lh, 7, pd)) for _ in range(4)]
    A.frame_init_lowres(depth, dev(src), 0, ss, d4, ds, lw, lh)
    torch.cuda.synchronize()
    for g, x in zip(d4, wd4):
        assert np.array_equal(back(g, pd), x)
    # propagateCost + fix8 rows
    n = 4097
    pin, inter = rng.integers(0, 65536, size=n).astype(np.uint16), rng.integers(0, 65536, size=n).astype(np.uint16)
    intra, invq = rng.integers(1, 1 << 15, size=n).astype(np.int32), rng.integers(1, 1 << 15, size=n).astype(np.int32)
    fps = np.array([float(rng.uniform(2.56, 256.0))], np.float64)
    wantp = np.zeros(n, np.int32)
    orc.fn("propagateCost")(ptr(wantp), ptr(pin), ptr(intra), ptr(inter), ptr(invq), ptr(fps), n)
    dd = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    A.propagate_cost(dd, dev(pin), dev(intra), dev(inter), dev(invq), float(fps[0]), n)
    torch.cuda.synchronize()
    assert np.array_equal(dd.cpu().numpy(), wantp)
    q = rng.uniform(-127.0, 127.0, size=n)
    wantq, wantu = np.zeros(n, np.uint16), np.zeros(n, np.float64)
    orc.fn("fix8Pack")(ptr(wantq), ptr(q), n)
    orc.fn("fix8Unpack")(ptr(wantu), ptr(wantq), n)
    dq, du = torch.zeros(n, dtype=torch.int16, device="cuda:0"), torch.zeros(n, dtype=torch.float64, device="cuda:0")
    one = A.make_jobs([([0, 0], [])], "cuda:0")
    A.frame_batch(A.FR_FIX8_PACK, depth, n, 1, [A.Plane(dev(q).data_ptr(), 0), A.Plane(dq.data_ptr(), 0)], one, 1)
    A.frame_batch(A.FR_FIX8_UNPACK, depth, n, 1, [A.Plane(dq.data_ptr(), 0), A.Plane(du.data_ptr(), 0)], one, 1)
    torch.cuda.synchronize()
    assert np.array_equal(back(dq, np.uint16), wantq) and du.cpu().numpy().tobytes() == wantu.tobytes()


def test_edges_empty_batches_bad_arguments_and_the_missing_bit_cost_table(repo_root):
    """Empty batches are accepted and touch nothing; 1-sample-wide planes; bad kinds / depths / NULL operands are refused with a message; the two estimators
    that price context-coded bins refuse to run before the host handed its table in (and the table filler then leaves those two slots to the host)."""
    import torch
    L = A.lib()
    L.x265hip_last_error.restype = ctypes.c_char_p
    one = A.make_jobs([([0, 0], [0, 255])], "cuda:0")
    src = torch.arange(7, dtype=torch.uint8, device="cuda:0")
    dst = torch.full((7,), 9, dtype=torch.uint8, device="cuda:0")
    pl = [A.Plane(src.data_ptr(), 1), A.Plane(dst.data_ptr(), 1)]
    A.frame_batch(A.FR_PLANECOPY_CP, 8, 1, 7, pl, one, 0)                      # njobs = 0
    torch.cuda.synchronize()
    assert dst.cpu().tolist() == [9] * 7
    A.frame_batch(A.FR_PLANECOPY_CP, 8, 1, 7, pl, one, 1)                      # a column of 7 rows, stride 1
    torch.cuda.synchronize()
    assert dst.cpu().tolist() == list(range(7))
    f = L.x265hip_frame_batch
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(A.Plane), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    arr = (A.Plane * 2)(*pl)
    assert f(99, 8, 1, 7, arr, one.data_ptr(), 1, None, None) < 0 and b"kind" in L.x265hip_last_error()
    assert f(A.FR_PLANECOPY_CP, 9, 1, 7, arr, one.data_ptr(), 1, None, None) < 0 and b"depth" in L.x265hip_last_error()
    assert f(A.FR_PLANECOPY_CP, 8, 0, 7, arr, one.data_ptr(), 1, None, None) < 0
    assert f(A.FR_PLANE_CLIP_MAX, 10, 4, 4, arr, one.data_ptr(), 1, None, None) < 0 and b"out" in L.x265hip_last_error()
    assert f(A.FR_PLANECOPY_CP, 8, 1, 7, None, one.data_ptr(), 1, None, None) < 0
    g = L.x265hip_coeff_batch
    g.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    bufs = (ctypes.c_void_p * 5)(src.data_ptr(), src.data_ptr(), dst.data_ptr(), src.data_ptr(), dst.data_ptr())
    cj = A.make_coeff_jobs([([0] * 5, [1, 0])], "cuda:0")
    res = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    assert g(A.CF_COST_COEFF_REMAIN, 8, bufs, cj.data_ptr(), 0, res.data_ptr(), None) == 0          # empty
    assert g(42, 8, bufs, cj.data_ptr(), 1, res.data_ptr(), None) < 0 and b"kind" in L.x265hip_last_error()
    assert g(A.CF_COST_COEFF_REMAIN, 8, bufs, cj.data_ptr(), 1, None, None) < 0 and b"result" in L.x265hip_last_error()
    assert g(A.CF_COST_COEFF_REMAIN, 8, None, cj.data_ptr(), 1, res.data_ptr(), None) < 0
    L.x265hip_set_entropy_bits.argtypes = [ctypes.c_void_p]
    assert L.x265hip_set_entropy_bits(None) < 0
    # wait policy: accepted values only; spinning (the runtime's default) restored afterwards
    assert L.x265hip_set_wait_policy(7) < 0
    assert L.x265hip_set_wait_policy(0) == 0 and L.x265hip_set_wait_policy(1) == 0
