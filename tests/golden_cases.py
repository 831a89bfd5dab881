"""Golden-vector cases shared by tools/gen_golden.py (records reference outputs) and tests/test_golden.py
(replays them on the oracle table and on the HIP table).  A case = (slot path, tag, argument template):
every array argument is stored explicitly in the .npz, so a fixture is pure data."""
from __future__ import annotations

import numpy as np

import harness as H

spec = H.spec


def _pix(rng, depth, n):
    return rng.integers(0, H.pixel_max(depth) + 1, size=n).astype(H.pix_dtype(depth))


def cases(depth):
    """Yield dicts: path, tag, build(rng) -> list of args where arrays are ('in'|'out'|'inout', ndarray, offset)."""
    m = H.pixel_max(depth)
    out = []

    def add(path, tag, builder):
        out.append({"path": path, "tag": tag, "build": builder})

    # pixel compare: every PU size for sad/satd, every CU for sa8d / sse / psy
    for i in range(25):
        w, h = spec.pu_dims(i)
        for fld in ("sad", "satd"):
            def b(rng, w=w, h=h):
                a, bb = _pix(rng, depth, 64 * h + 8), _pix(rng, depth, (w + 9) * h + 8)
                return [("in", a, 0), 64, ("in", bb, 3), w + 9]
            add(f"pu[{i}].{fld}", "rand", b)
    for i, n in enumerate(spec.LUMA_CU):
        for fld in ("sa8d", "sse_pp", "psy_cost_pp"):
            def b(rng, n=n):
                a, bb = _pix(rng, depth, 64 * n + 8), _pix(rng, depth, (n + 5) * n + 8)
                return [("in", a, 0), 64, ("in", bb, 1), n + 5]
            add(f"cu[{i}].{fld}", "rand", b)
        def bmax(rng, n=n):
            a = np.zeros(64 * n + 8, H.pix_dtype(depth)); bb = np.full((n + 5) * n + 8, m, H.pix_dtype(depth))
            return [("in", a, 0), 64, ("in", bb, 1), n + 5]
        add(f"cu[{i}].sa8d", "minmax", bmax)
    # transforms
    for i, n in enumerate((4, 8, 16, 32)):
        def bd(rng, n=n):
            src = rng.integers(-m, m + 1, size=(n + 3) * n + 8).astype(np.int16)
            return [("in", src, 2), ("out", np.zeros(n * n, np.int16), 0), n + 3]
        add(f"cu[{i}].dct", "rand", bd)
        def bdm(rng, n=n):
            src = np.full((n + 3) * n + 8, m, np.int16)
            return [("in", src, 2), ("out", np.zeros(n * n, np.int16), 0), n + 3]
        add(f"cu[{i}].dct", "max", bdm)
        def bi(rng, n=n):
            src = rng.integers(-32768, 32768, size=n * n).astype(np.int16)
            return [("in", src, 0), ("out", np.zeros((n + 3) * n + 8, np.int16), 1), n + 3]
        add(f"cu[{i}].idct", "rand", bi)
    add("dst4x4", "rand", lambda rng: [("in", rng.integers(-m, m + 1, size=7 * 4 + 8).astype(np.int16), 1), ("out", np.zeros(16, np.int16), 0), 7])
    add("idst4x4", "rand", lambda rng: [("in", rng.integers(-32768, 32768, size=16).astype(np.int16), 0), ("out", np.zeros(7 * 4 + 8, np.int16), 1), 7])
    # quant / dequant
    for n in (16, 256, 1024):
        def bq(rng, n=n):
            return [("in", rng.integers(-m, m + 1, size=n).astype(np.int16), 0), ("in", rng.integers(1, m, size=n).astype(np.int32), 0),
                    ("out", np.zeros(n, np.int32), 0), ("out", np.zeros(n, np.int16), 0), 17, 85 << 8, n]
        add("quant", f"n{n}", bq)
        add("dequant_normal", f"n{n}", lambda rng, n=n: [("in", rng.integers(-32768, 32768, size=n).astype(np.int16), 0),
                                                         ("out", np.zeros(n, np.int16), 0), n, 57 << 3, 4])
    # interpolation: a luma and a chroma size, every coefficient index
    for fld, pu, short_in, short_out in (("luma_hpp", 2, 0, 0), ("luma_vpp", 9, 0, 0), ("luma_hps", 7, 0, 1), ("luma_vsp", 8, 1, 0),
                                         ("luma_vss", 2, 1, 1), ("luma_vps", 13, 0, 1)):
        w, h = spec.pu_dims(pu)
        for idx in range(4):
            def bf(rng, w=w, h=h, idx=idx, short_in=short_in, short_out=short_out, fld=fld):
                ss = w + 16
                n = ss * (h + 16)
                src = rng.integers(-8192, 8192, size=n).astype(np.int16) if short_in else _pix(rng, depth, n)
                dst = np.zeros((w + 2) * (h + 8), np.int16 if short_out else H.pix_dtype(depth))
                args = [("in", src, 8 * ss + 8), ss, ("out", dst, 0), w + 2, idx]
                if fld.endswith("hps"):
                    args.append(1)
                return args
            add(f"pu[{pu}].{fld}", f"idx{idx}", bf)
    for ix, iy in ((1, 2), (3, 3), (2, 0)):
        def bhv(rng, ix=ix, iy=iy):
            ss = 48
            return [("in", _pix(rng, depth, ss * 48), 8 * ss + 8), ss, ("out", np.zeros(18 * 16, H.pix_dtype(depth)), 0), 18, ix, iy]
        add("pu[2].luma_hvpp", f"{ix}{iy}", bhv)
    for idx in range(8):
        def bc(rng, idx=idx):
            ss = 24
            return [("in", _pix(rng, depth, ss * 24), 4 * ss + 4), ss, ("out", np.zeros(10 * 8, H.pix_dtype(depth)), 0), 10, idx]
        add("chroma[1].pu[2].filter_hpp", f"idx{idx}", bc)
        add("chroma[1].pu[2].filter_vpp", f"idx{idx}", bc)
    # intra: every mode at every size, filtered and unfiltered edge
    for i, n in enumerate((4, 8, 16, 32)):
        for mode in range(35):
            for bf in (0, 1):
                def bi2(rng, n=n, mode=mode, bf=bf):
                    return [("out", np.zeros((n + 1) * n, H.pix_dtype(depth)), 0), n + 1, ("in", _pix(rng, depth, 4 * n + 1), 0), mode, bf]
                add(f"cu[{i}].intra_pred[{mode}]", f"bf{bf}", bi2)
        add(f"cu[{i}].intra_filter", "rand", lambda rng, n=n: [("in", _pix(rng, depth, 4 * n + 1), 0), ("out", np.zeros(4 * n + 1, H.pix_dtype(depth)), 0)])
    # SAO statistics (edge classes with carried buffers) and deblocking
    for fld, nbuf in (("saoCuStatsBO", 0), ("saoCuStatsE0", 0), ("saoCuStatsE1", 1), ("saoCuStatsE2", 2), ("saoCuStatsE3", 1)):
        def bs(rng, nbuf=nbuf):
            st = 72
            base = int(rng.integers(8, m - 8))
            rec = np.clip(base + rng.integers(-3, 4, size=st * 70), 0, m).astype(H.pix_dtype(depth))
            diff = rng.integers(-m, m + 1, size=64 * 66).astype(np.int16)
            args = [("in", diff, 0), ("in", rec, st + 2), st]
            for _ in range(nbuf):
                args.append(("inout", rng.integers(-1, 2, size=80).astype(np.int8), 4))
            args += [37, 23, ("inout", rng.integers(-100, 100, size=36).astype(np.int32), 0),
                     ("inout", rng.integers(0, 100, size=36).astype(np.int32), 0)]
            return args
        add(fld, "rand", bs)
    for d in (0, 1):
        def bl(rng, d=d):
            st = 24
            base = int(rng.integers(16, m - 16))
            rec = np.clip(base + rng.integers(-6, 7, size=st * 24), 0, m).astype(H.pix_dtype(depth))
            off, step = (1, st) if d == 0 else (st, 1)
            return [("inout", rec, 8 * st + 8), step, off, 9 << (depth - 8), 6 << (depth - 8)]
        add(f"pelFilterLumaStrong[{d}]", "rand", bl)
        def bch(rng, d=d):
            st = 24
            base = int(rng.integers(16, m - 16))
            rec = np.clip(base + rng.integers(-6, 7, size=st * 24), 0, m).astype(H.pix_dtype(depth))
            off, step = (1, st) if d == 0 else (st, 1)
            return [("inout", rec, 8 * st + 8), step, off, 5 << (depth - 8), -1, -1]
        add(f"pelFilterChroma[{d}]", "rand", bch)
    return out


def run_case(table, case, store, record):
    """record=True: call the slot, save inputs / outputs / return value into store.
    record=False: replay with the stored inputs, return list of mismatch strings."""
    path, tag = case["path"], case["tag"]
    key = f"{path}|{tag}"
    fn = table.fn(path)
    if fn is None:
        return [f"{key}: slot is NULL"]
    rng = np.random.default_rng([abs(hash(key)) % (2 ** 31), table.depth]) if record else None
    if record:
        # deterministic, hash-independent seed
        rng = np.random.default_rng([sum(ord(c) * (i + 1) for i, c in enumerate(key)) % (2 ** 31), table.depth])
        tmpl = case["build"](rng)
    else:
        tmpl = case["build"](np.random.default_rng(0))      # shapes / scalar slots only; arrays are replaced below
    args, arrays = [], []
    scal = []
    for k, a in enumerate(tmpl):
        if isinstance(a, tuple):
            role, arr, off = a
            if not record:
                arr = store[f"{key}|in{k}"].copy()
            else:
                store[f"{key}|in{k}"] = arr.copy()
            arrays.append((k, role, arr))
            args.append(H.ptr(arr, off))
        else:
            scal.append(int(a))
            args.append(int(a))
    if record:
        store[f"{key}|args"] = np.array(scal, dtype=np.int64)
    else:
        saved = store[f"{key}|args"].tolist()
        it = iter(saved)
        args = [a if isinstance(t, tuple) else next(it) for a, t in zip(args, tmpl)]
    ret = fn(*args)
    fails = []
    if ret is not None:
        if record:
            store[f"{key}|ret"] = np.array([int(ret)], dtype=np.uint64)
        elif int(store[f"{key}|ret"][0]) != int(ret) % (1 << 64):
            fails.append(f"{key}: return {ret} != golden {int(store[f'{key}|ret'][0])}")
    for k, role, arr in arrays:
        if role in ("out", "inout"):
            if record:
                store[f"{key}|out{k}"] = arr.copy()
            elif not np.array_equal(store[f"{key}|out{k}"], arr):
                fails.append(f"{key}: output {k} differs from golden")
    return fails
