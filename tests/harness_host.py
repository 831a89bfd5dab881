"""Handlers for the host-side slots: RDOQ cost pre-passes, CABAC bit estimators,
lowres downscale, SSIM moments, plane copies, cutree helpers.  Input recipes follow the reference harnesses
(source/test/pixelharness.cpp:1705-2080, mbdstharness.cpp:286-410) with seeded numpy generators.
Importing this module registers the handlers in harness.HANDLERS."""
from __future__ import annotations

import ctypes

import numpy as np

import harness as H
from harness import dims, fill, parse, pix_dtype, pixel_max, ptr


def diag_scan(n):
    """Up-right diagonal scan of an n x n block as raster positions (H.265 6.5.3)."""
    out = []
    for s in range(2 * n - 1):
        for y in range(min(s, n - 1), -1, -1):
            x = s - y
            if x < n:
                out.append(y * n + x)
    return np.array(out, dtype=np.uint16)


def block_scan(rng, tr, kind):
    """Scan order of a tr x tr block visiting 4x4 groups one after the other (16 consecutive scan positions = one
    group, like g_scanOrder): groups and in-group positions follow `kind`."""
    g = tr // 4
    if kind == 0:
        inner, outer = diag_scan(4), diag_scan(g) if g > 1 else np.zeros(1, np.uint16)
    elif kind == 1:
        inner, outer = np.arange(16, dtype=np.uint16), np.arange(g * g, dtype=np.uint16)
    else:
        inner, outer = rng.permutation(16).astype(np.uint16), rng.permutation(g * g).astype(np.uint16)
    out = []
    for cg in outer:
        cy, cx = divmod(int(cg), g)
        for p in inner:
            y, x = divmod(int(p), 4)
            out.append((cy * 4 + y) * tr + cx * 4 + x)
    return np.array(out, dtype=np.uint16), inner.copy()


def sparse_coeffs(rng, case, n):
    """pixelharness.cpp:1713-1725: mostly zero, mostly negative 15-bit levels."""
    if case == "min":
        return np.full(n, -32768, np.int16)
    if case == "max":
        return np.full(n, 32767, np.int16)
    v = rng.integers(0, 32768, size=n)
    v[v < 32767 * 2 // 3] = 0
    v = np.where(rng.integers(0, 10, size=n) < 8, -v, v)
    return v.astype(np.int16)


def _rdoq(kind):
    def h(fa, fb, path, depth, rng, case):
        n = dims(path)[0]
        lo, hi = -32768, 32767
        resi = fill(rng, case, n * n, np.int16, lo, hi)
        fenc = fill(rng, case, n * n, np.int16, lo, hi)
        cost0 = rng.integers(-(1 << 40), 1 << 40, size=n * n).astype(np.int64)
        tot0 = rng.integers(0, 1 << 40, size=2).astype(np.int64)
        psy = np.array([int(rng.integers(0, 1 << 16))], np.int64)
        g = n // 4
        blk = int(rng.integers(0, g)) * 4 * n + int(rng.integers(0, g)) * 4
        outs = []
        for f in (fa, fb):
            cost, tu, tr = cost0.copy(), tot0[:1].copy(), tot0[1:].copy()
            if kind in ("nonpsy", "psy1"):
                f(ptr(resi), ptr(cost), ptr(tu), ptr(tr), blk)
            else:
                f(ptr(resi), ptr(fenc), ptr(cost), ptr(tu), ptr(tr), ptr(psy), blk)
            outs.append((cost, int(tu[0]), int(tr[0])))
        return [("costUncoded", outs[0][0], outs[1][0]), ("totalUncoded", outs[0][1], outs[1][1]), ("totalRd", outs[0][2], outs[1][2])]
    return h


def _cmp_scan_pos_last(fa, fb, path, depth, rng, case):
    log2 = int(rng.integers(2, 6))
    tr = 1 << log2
    scan, inner = block_scan(rng, tr, int(rng.integers(0, 3)))
    coeff = sparse_coeffs(rng, case, tr * tr)
    if not np.any(coeff):
        coeff[-1] = -1
    nsig = int(np.count_nonzero(coeff))
    outs = []
    for f in (fa, fb):
        sign, flag, num = np.full(64, 0xCDCD, np.uint16), np.full(64, 0xCDCD, np.uint16), np.full(64, 0xCD, np.uint8)
        r = f(ptr(scan), ptr(coeff), ptr(sign), ptr(flag), ptr(num), nsig, ptr(inner), tr)
        outs.append((r, sign, flag, num))
    return [(nm, outs[0][i], outs[1][i]) for i, nm in enumerate(("scanPosLast", "coeffSign", "coeffFlag", "coeffNum"))]


def _cmp_find_pos_first_last(fa, fb, path, depth, rng, case):
    tr = 1 << int(rng.integers(2, 6))
    scan = (diag_scan(4) if rng.integers(0, 2) else rng.permutation(16).astype(np.uint16))
    coeff = sparse_coeffs(rng, case, 4 * tr + 8)
    if not any(coeff[(p >> 2) * tr + (p & 3)] for p in range(16)):
        p = int(rng.integers(0, 16))
        coeff[(p >> 2) * tr + (p & 3)] = 3          # an all-zero group leaves the upper 24 bits undefined
    return [("packed", fa(ptr(coeff), tr, ptr(scan)), fb(ptr(coeff), tr, ptr(scan)))]


def _ctx_states(rng, n):
    return rng.integers(2, 125, size=n).astype(np.uint8)     # pixelharness.cpp:1880


def _cmp_cost_coeff_nxn(fa, fb, path, depth, rng, case):
    size_idx = int(rng.integers(0, 4))
    tr = 4 << size_idx
    offset = 0 if size_idx == 0 else (9 if size_idx == 1 else 12)
    scan, inner = block_scan(rng, tr, int(rng.integers(0, 3)))
    tab = rng.integers(0, 9, size=16).astype(np.uint8)
    coeff = sparse_coeffs(rng, "random" if case != "random" else case, tr * tr + 4 * tr)
    if rng.integers(0, 2):            # dense blocks: every visited position of the group writes a level, up to the last entry of absCoeff
        coeff = np.where(coeff == 0, rng.integers(-9, 10, size=coeff.size), coeff).astype(np.int16)
    ncg = tr * tr // 16
    cg = int(rng.integers(0, ncg))
    sub_base = cg * 16
    off = int(rng.integers(0, 16))
    origin = int(scan[sub_base])                             # raster position of the group's first scan position ...
    origin = (origin // tr // 4 * 4) * tr + (origin % tr) // 4 * 4   # ... -> the group's top-left corner
    mask, nsig = 0, 0
    for k in range(off + 1):
        p = int(inner[k])
        c = int(coeff[origin + (p >> 2) * tr + (p & 3)])
        mask = mask * 2 + (c != 0)
        nsig += c != 0
    if nsig == 0:
        p = int(inner[off])
        coeff[origin + (p >> 2) * tr + (p & 3)] = -2
        mask |= 1
    nnz0 = 1 if off < 15 else 0
    ctx0 = _ctx_states(rng, 64)
    outs = []
    for f in (fa, fb):
        ctx = ctx0.copy()
        absc = np.full(24, 0xCDCD, np.uint16)
        r = f(ptr(inner), ptr(coeff, origin), tr, ptr(absc, 2 + nnz0), ptr(tab), mask, ptr(ctx), offset, off, sub_base)
        outs.append((r, ctx, absc))
    return [("bits", outs[0][0], outs[1][0]), ("contexts", outs[0][1], outs[1][1]), ("absCoeff", outs[0][2], outs[1][2])]


def _cmp_cost_coeff_remain(fa, fb, path, depth, rng, case):
    a = rng.integers(0, 32768, size=40)
    a[a < 32767 * 2 // 3] = 1
    if case == "max":
        a[:] = 32767
    a = a.astype(np.uint16)
    nnz = int(rng.integers(0, 17))
    first = next((k for k in range(8) if a[k] >= 2), 8)
    return [("bits", fa(ptr(a), nnz, first), fb(ptr(a), nnz, first))]


def _cmp_cost_c1c2(fa, fb, path, depth, rng, case):
    vals = []
    for _ in range(8):
        v = int(rng.integers(0, 32768))
        v = 0 if v < 32767 // 3 else (1 if v < 32767 * 2 // 3 else (2 if v < 32767 * 3 // 4 else v))
        if v:
            vals.append(v)
    if not vals:
        vals = [1]
    a = np.zeros(16, np.uint16)
    a[:len(vals)] = vals
    ctx0 = _ctx_states(rng, 8)
    off = int(rng.integers(0, 4))
    outs = []
    for f in (fa, fb):
        ctx = ctx0.copy()
        outs.append((f(ptr(a), len(vals), ptr(ctx), off), ctx))
    return [("packed", outs[0][0], outs[1][0]), ("contexts", outs[0][1], outs[1][1])]


def _cmp_lowres(fa, fb, path, depth, rng, case):
    w, h = int(rng.integers(4, 40)), int(rng.integers(2, 20))
    ss, ds = 2 * w + 8 + int(rng.integers(0, 8)), w + int(rng.integers(0, 8))
    src = fill(rng, case, ss * (2 * h + 2), pix_dtype(depth), 0, pixel_max(depth))
    outs = []
    for f in (fa, fb):
        d = [np.full(ds * h + 8, 7, pix_dtype(depth)) for _ in range(4)]
        f(ptr(src), ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ss, ds, w, h)
        outs.append(np.concatenate(d))
    return [("planes", outs[0], outs[1])]


def _cmp_ssim_core(fa, fb, path, depth, rng, case):
    st1, st2 = 8 + int(rng.integers(0, 9)), 8 + int(rng.integers(0, 9))
    a = fill(rng, case, st1 * 4 + 8, pix_dtype(depth), 0, pixel_max(depth))
    b = fill(rng, "random", st2 * 4 + 8, pix_dtype(depth), 0, pixel_max(depth))
    outs = []
    for f in (fa, fb):
        s = np.zeros(8, np.int32)
        f(ptr(a), st1, ptr(b), st2, ptr(s))
        outs.append(s)
    return [("sums", outs[0], outs[1])]


def _cmp_ssim_end(fa, fb, path, depth, rng, case):
    # moments of real pixel blocks, so the 8-bit build's int arithmetic stays in range (pixel.cpp:661-664)
    def moments():
        m = np.zeros((5, 4), np.int32)
        for i in range(5):
            a = fill(rng, case, 16, np.int64, 0, pixel_max(depth))
            b = np.clip(a + rng.integers(-20, 21, size=16), 0, pixel_max(depth))
            m[i] = [a.sum(), b.sum(), ((a * a + b * b).sum()) & 0xffffffff, (a * b).sum() & 0xffffffff]
        return m
    s0, s1 = moments(), moments()
    width = int(rng.integers(1, 5))
    ra, rb = fa(ptr(s0), ptr(s1), width), fb(ptr(s0), ptr(s1), width)
    return [("ssim_bits", np.float32(ra).tobytes(), np.float32(rb).tobytes())]


def _planecopy(kind):
    def h(fa, fb, path, depth, rng, case):
        w, hh = int(rng.integers(1, 70)), int(rng.integers(1, 12))
        ss, ds = w + int(rng.integers(0, 9)), w + int(rng.integers(0, 9))
        pd = pix_dtype(depth)
        if kind == "cp":
            src = fill(rng, case, ss * hh, np.uint8, 0, 255)
            args = (depth - 8,)
        elif kind == "pp_shr":
            src = fill(rng, case, ss * hh, pd, 0, pixel_max(depth))
            args = (int(rng.integers(0, 5)),)
        else:
            src = fill(rng, case, ss * hh, np.uint16, 0, 65535)
            args = (int(rng.integers(0, 7)), int((1 << depth) - 1))
        outs = []
        for f in (fa, fb):
            d = np.full(ds * hh, 3, pd)
            f(ptr(src), ss, ptr(d), ds, w, hh, *args)
            outs.append(d)
        return [("dst", outs[0], outs[1])]
    return h


def _cmp_propagate(fa, fb, path, depth, rng, case):
    n = int(rng.integers(1, 200))
    pin = fill(rng, case, n, np.uint16, 0, 65535)
    intra = rng.integers(1, 1 << 15, size=n).astype(np.int32)
    inter = fill(rng, case, n, np.uint16, 0, 65535)
    invq = rng.integers(1, 1 << 15, size=n).astype(np.int32)
    fps = np.array([float(rng.uniform(2.56, 256.0))], np.float64)
    outs = []
    for f in (fa, fb):
        d = np.zeros(n, np.int32)
        f(ptr(d), ptr(pin), ptr(intra), ptr(inter), ptr(invq), ptr(fps), n)
        outs.append(d)
    return [("dst", outs[0], outs[1])]


def _cmp_fix8_pack(fa, fb, path, depth, rng, case):
    n = int(rng.integers(1, 100))
    src = rng.uniform(-127.0, 127.0, size=n)
    outs = []
    for f in (fa, fb):
        d = np.zeros(n, np.uint16)
        f(ptr(d), ptr(src), n)
        outs.append(d)
    return [("dst", outs[0], outs[1])]


def _cmp_fix8_unpack(fa, fb, path, depth, rng, case):
    n = int(rng.integers(1, 100))
    src = fill(rng, case, n, np.uint16, 0, 65535)
    outs = []
    for f in (fa, fb):
        d = np.zeros(n, np.float64)
        f(ptr(d), ptr(src), n)
        outs.append(d.tobytes())
    return [("dst", outs[0], outs[1])]


def _cmp_plane_clip_max(fa, fb, path, depth, rng, case):
    w, hh = int(rng.integers(1, 70)), int(rng.integers(1, 12))
    st = w + int(rng.integers(0, 9))
    src0 = fill(rng, case, st * hh, pix_dtype(depth), 0, pixel_max(depth))
    lo, hi = int(rng.integers(0, 100)), int(rng.integers(500, pixel_max(depth) + 1))
    outs = []
    for f in (fa, fb):
        s = src0.copy()
        tot = np.zeros(1, np.uint64)
        r = f(ptr(s), st, w, hh, ptr(tot), lo, hi)
        outs.append((int(r), s, int(tot[0])))
    return [("max", outs[0][0], outs[1][0]), ("plane", outs[0][1], outs[1][1]), ("sum", outs[0][2], outs[1][2])]


H.HANDLERS.update({
    "nonPsyRdoQuant": _rdoq("nonpsy"), "psyRdoQuant": _rdoq("psy"), "psyRdoQuant_1p": _rdoq("psy1"), "psyRdoQuant_2p": _rdoq("psy2"),
    "scanPosLast": _cmp_scan_pos_last, "findPosFirstLast": _cmp_find_pos_first_last, "costCoeffNxN": _cmp_cost_coeff_nxn,
    "costCoeffRemain": _cmp_cost_coeff_remain, "costC1C2Flag": _cmp_cost_c1c2,
    "frameInitLowres": _cmp_lowres, "frameInitLowerRes": _cmp_lowres,
    "ssim_4x4x2_core": _cmp_ssim_core, "ssim_end_4": _cmp_ssim_end,
    "planecopy_cp": _planecopy("cp"), "planecopy_sp": _planecopy("sp"), "planecopy_sp_shl": _planecopy("sp"),
    "planecopy_pp_shr": _planecopy("pp_shr"),
    "propagateCost": _cmp_propagate, "fix8Pack": _cmp_fix8_pack, "fix8Unpack": _cmp_fix8_unpack,
    "planeClipAndMax": _cmp_plane_clip_max,
})
