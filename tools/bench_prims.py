#!/usr/bin/env python3
"""[bench.py --prims: the primitive-level part of bench.py's cpu_baseline leg - the only place besides tests/ and smoke() that
loads oracle/, and only to time the CPU path beside the kernels]

Per-family throughput of the batch-layer kernels on one MI355X, with the CPU path timed beside each
(primitive level).

For every family: N blocks per launch (inputs resident in HBM), HIP-event time per launch, algorithmic bytes per block
as defined in SURVEY.md section 8(d), achieved GB/s and the fraction of the 8 TB/s HBM peak.  The CPU columns run the
SAME job list through (a) the real reference C primitive from oracle/_ref (when that build travelled to this box) and
(b) the oracle's AVX2-compiled restatement, OpenMP over jobs on all host cores (oracle/x265_oracle_bench.c).
For the MFMA transforms also int8 TOPS (4*N^3 MACs per block x 2 limbs) against the 5 POPS dense int8 peak.

    python tools/bench_prims.py > profiles/rNN_prims.txt
"""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
spec = importlib.import_module("x265-yuuki-asuna_amd.table_spec")
import harness as H  # noqa: E402  (TEST INFRASTRUCTURE: table loaders for the CPU columns)

HBM = 8000.0
I8_PEAK_TOPS = 5000.0
(SIG_PIXELCMP, SIG_SAD_X4, SIG_FILTER, SIG_FILTER_HPS, SIG_FILTER_HV, SIG_P2S, SIG_DCT, SIG_QUANT, SIG_NQUANT,
 SIG_DEQUANT_NORMAL, SIG_INTRA_PRED, SIG_INTRA_ALLANGS, SIG_COPY, SIG_SUB_PS, SIG_ADD_PS, SIG_ADDAVG, SIG_SAO_E0,
 SIG_SAO_B0, SIG_DEBLOCK_LUMA, SIG_VAR, SIG_CALCRES) = range(21)


class BPlane(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("stride", ctypes.c_ssize_t), ("elem", ctypes.c_int)]


def timeit(fn, iters=10):
    import torch
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def jobs_np(n, offs=(), args=()):
    """offs / args: sequences of scalars or length-n arrays -> numpy array of x265hip_job records."""
    arr = np.zeros(n, dtype=A.job_dtype())
    for k, o in enumerate(offs):
        arr["off"][:, k] = o
    for k, a in enumerate(args):
        arr["arg"][:, k] = a
    return arr


def to_dev(a, dev):
    import torch
    if a.dtype == np.uint16:
        a = a.view(np.int16)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(dev)


def jobs_dev(arr, dev):
    import torch
    return torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(dev)


class CPU:
    """The CPU columns: the reference build's table (if present) and the oracle's AVX2 flavour, timed by
    x265oracle_time_jobs over the same job list."""

    def __init__(self):
        self.lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle_avx2.so"))
        self.lib.x265oracle_time_jobs.restype = ctypes.c_double
        self.lib.x265oracle_time_jobs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        sys.path.insert(0, ROOT)
        from bench import effective_cpus
        self.threads = effective_cpus()        # cgroup CPU quota, not the socket's thread count
        self.port = H.load_oracle(8, ROOT, avx2=True)
        self.ref = H.load_reference(8, ROOT)

    def time(self, table, path, sig, planes, jobs, reps=2):
        fn = table.ptr(path)
        if not fn:
            return None
        pl = (BPlane * 4)()
        keep = []
        for i, (arr, stride) in enumerate(planes):
            if arr is None:
                continue
            arr = arr.copy()                 # in-place primitives must not disturb the GPU's input
            keep.append(arr)
            pl[i] = BPlane(arr.ctypes.data, stride, arr.itemsize)
        j = np.ascontiguousarray(jobs)
        t1 = self.lib.x265oracle_time_jobs(fn, sig, ctypes.byref(pl), j.ctypes.data, len(j), 1, self.threads, None)
        reps = int(min(200, max(3, 0.25 / max(t1, 1e-6))))            # ~0.25 s of CPU work per measurement
        return self.lib.x265oracle_time_jobs(fn, sig, ctypes.byref(pl), j.ctypes.data, len(j), reps, self.threads, None)


def pu_index(w, h):
    return next(i for i in range(25) if spec.pu_dims(i) == (w, h))


HEADER = (f"{'kernel (8-bit unless noted)':40s} {'blocks':>8s} {'GPU us':>9s} {'B/blk':>7s} {'GPU GB/s':>9s} {'frac':>6s} "
          f"{'ref-C GB/s':>10s} {'port GB/s':>10s} {'GPU/CPU':>8s}  notes")


def report(cpu, name, n, t_gpu, bpj, path, sig, planes, jobs, extra=""):
    gbs = n * bpj / t_gpu / 1e9
    cols = []
    best = None
    for tab in (cpu.ref, cpu.port):
        t = cpu.time(tab, path, sig, planes, jobs) if tab is not None else None
        cols.append(f"{n * bpj / t / 1e9:10.1f}" if t else f"{'-':>10s}")
        if t and (best is None or t < best):
            best = t
    ratio = f"{best / t_gpu:8.1f}" if best else f"{'-':>8s}"
    print(f"{name:40s} {n:8d} {t_gpu * 1e6:9.1f} {bpj:7d} {gbs:9.1f} {gbs / HBM:6.3f} {cols[0]} {cols[1]} {ratio}  {extra}", flush=True)


def main():
    import argparse
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma-separated families to run: compare,interp,transform,quant,intra,intratu,blockop,loopfilter,sao,frame,coeff")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU columns (profiling runs)")
    args = ap.parse_args()
    want = set(x for x in args.only.split(",") if x)
    on = lambda fam: not want or fam in want
    dev = "cuda:0"
    rng = np.random.default_rng(5)
    cpu = CPU()
    if args.no_cpu:
        cpu.ref = cpu.port = None
    print(f"# host CPU columns: {cpu.threads} OpenMP threads (= the container's CPU quota; {os.cpu_count()} hardware threads visible); ref-C = real reference C primitives (oracle/_ref, g++ -O2, "
          f"{'present' if cpu.ref is not None else 'ABSENT on this box'}); port = oracle restatement, gcc -O3 -march=x86-64-v3")
    print(HEADER)

    W, Hh, st = 3840, 2176, 4032
    ref_h = rng.integers(0, 256, size=st * (Hh + 160), dtype=np.uint8)
    ref_d = to_dev(ref_h, dev)

    # ---- a1 / a4 / a5 / a6: block compare on a 4K plane, random candidate positions, fenc blocks at stride 64 ----
    for kind, name, grp in () if not on("compare") else ((A.CMP_SAD, "sad", "pu"), (A.CMP_SATD, "satd", "pu"), (A.CMP_SA8D, "sa8d", "cu"),
                            (A.CMP_SSE_PP, "sse_pp", "cu"), (A.CMP_PSY_COST, "psy_cost_pp", "cu")):
        for w in (8, 16, 32, 64):
            if name in ("sse_pp", "psy_cost_pp") and w not in (16, 32):
                continue
            n = 1 << 18 if w <= 16 else 1 << 16
            fenc_h = rng.integers(0, 256, size=n * 64 + 64 * 64, dtype=np.uint8)
            aoff = (np.arange(n, dtype=np.int64) * 64) % (n * 64 - 64 * 64)
            boff = (rng.integers(80, Hh - 80, size=n) * st + rng.integers(96, W - 96, size=n)).astype(np.int64)
            fenc_d, ao, bo = to_dev(fenc_h, dev), to_dev(aoff, dev), to_dev(boff, dev)
            out = torch.zeros(n, dtype=torch.int64, device=dev)
            t = timeit(lambda: A.pixelcmp_batch(kind, 8, w, w, fenc_d, 64, ref_d, st, n, out, a_off=ao, b_off=bo))
            path = f"pu[{pu_index(w, w)}].{name}" if grp == "pu" else f"cu[{int(np.log2(w)) - 2}].{name}"
            report(cpu, f"{name} {w}x{w}", n, t, 2 * w * w + 8, path, SIG_PIXELCMP, [(fenc_h, 64), (ref_h, st)],
                   jobs_np(n, (aoff, boff)), "random positions: each row of a block is its own DRAM sector" if w == 8 and name == "sad" else "")

    # the same kernels on jobs in PICTURE ORDER (the block grid of a plane, candidates within +-8 samples of the block's own position - what a
    # refinement pass over a picture issues): neighbouring jobs share DRAM lines, so the fetched bytes approach the algorithmic ones
    if on("compare"):
        src_h = rng.integers(0, 256, size=st * (Hh + 160), dtype=np.uint8)
        src_d = to_dev(src_h, dev)
        for kind, name in ((A.CMP_SAD, "sad"), (A.CMP_SATD, "satd")):
            for w in (8, 16):
                bw, bh = (W - 192) // w, (Hh - 160) // w
                by, bx = np.divmod(np.arange(bw * bh, dtype=np.int64), bw)
                n = bw * bh
                aoff = (80 + by * w) * st + 96 + bx * w
                boff = aoff + rng.integers(-8, 9, size=n) * st + rng.integers(-8, 9, size=n)
                ao, bo = to_dev(aoff, dev), to_dev(boff, dev)
                out = torch.zeros(n, dtype=torch.int64, device=dev)
                t = timeit(lambda: A.pixelcmp_batch(kind, 8, w, w, src_d, st, ref_d, st, n, out, a_off=ao, b_off=bo))
                report(cpu, f"{name} {w}x{w} picture order", n, t, 2 * w * w + 8, f"pu[{pu_index(w, w)}].{name}", SIG_PIXELCMP, [(src_h, st), (ref_h, st)],
                       jobs_np(n, (aoff, boff)), "block grid of a 4K plane, candidates within +-8 samples")

    # ---- a11: interpolation ----
    def interp_case(kind, kname, slot, sig, w, taps, dst_dtype, src_short=False, bpj=None):
        n = 1 << 16 if w <= 16 else 1 << 14
        src_h = ref_h if not src_short else rng.integers(-8192, 8192, size=st * (Hh + 160)).astype(np.int16)
        src_d = ref_d if not src_short else to_dev(src_h, dev)
        off0 = (rng.integers(80, Hh - 80, size=n) * st + rng.integers(96, W - 96, size=n)).astype(np.int64)
        rows = w + (taps - 1 if kind == A.IP_HPS else 0)
        off1 = np.arange(n, dtype=np.int64) * w * rows
        nidx = 4 if taps == 8 else 8
        a0, a1 = rng.integers(1, nidx, size=n), (np.ones(n, np.int64) if kind == A.IP_HPS else rng.integers(1, nidx, size=n))
        jb = jobs_np(n, (off0, off1), (a0, a1))
        jd = jobs_dev(jb, dev)
        dst_h = np.zeros(n * w * rows, dtype=dst_dtype)
        dst_d = to_dev(dst_h, dev)
        t = timeit(lambda: A.interp_batch(kind, 8, taps, w, w, A.plane(src_d, st), A.plane(dst_d, w), jd, n))
        pu = pu_index(w, w) if taps == 8 else pu_index(2 * w, 2 * w)
        path = f"pu[{pu}].{slot}" if taps == 8 else f"chroma[1].pu[{pu}].{slot}"
        es, ed = src_h.itemsize, dst_h.itemsize
        if bpj is None:
            ap = taps - 1
            hor = kind in (A.IP_HPP, A.IP_HPS, A.IP_HVPP)
            ver = kind in (A.IP_VPP, A.IP_VPS, A.IP_VSP, A.IP_VSS, A.IP_HVPP) or kind == A.IP_HPS
            bpj = (w + (ap if hor else 0)) * (w + (ap if ver else 0)) * es + w * rows * ed
        report(cpu, f"{'luma' if taps == 8 else 'chroma'} {kname} {w}x{w}", n, t, bpj, path, sig,
               [(src_h, st), (dst_h, w)], jb)

    for w in (16, 64) if on("interp") else ():
        interp_case(A.IP_HPP, "hpp", "luma_hpp", SIG_FILTER, w, 8, np.uint8)
        interp_case(A.IP_VPP, "vpp", "luma_vpp", SIG_FILTER, w, 8, np.uint8)
        interp_case(A.IP_HVPP, "hvpp", "luma_hvpp", SIG_FILTER_HV, w, 8, np.uint8)
    if on("interp"):
        interp_case(A.IP_HPS, "hps(+rows)", "luma_hps", SIG_FILTER_HPS, 16, 8, np.int16)
        interp_case(A.IP_VSP, "vsp", "luma_vsp", SIG_FILTER, 16, 8, np.uint8, src_short=True)
        interp_case(A.IP_VSS, "vss", "luma_vss", SIG_FILTER, 16, 8, np.int16, src_short=True)
        interp_case(A.IP_HPP, "hpp", "filter_hpp", SIG_FILTER, 8, 4, np.uint8)
        interp_case(A.IP_VPP, "vpp", "filter_vpp", SIG_FILTER, 8, 4, np.uint8)
        interp_case(A.IP_P2S, "p2s", "convert_p2s[0]", SIG_P2S, 32, 8, np.int16, bpj=32 * 32 * 3)

    # ---- a7: transforms, VALU vs MFMA ----
    for n_ in (4, 8, 16, 32) if on("transform") else ():
        nb = 1 << 16
        src_h = rng.integers(-255, 256, size=nb * n_ * n_, dtype=np.int16)
        dst_h = np.zeros(nb * n_ * n_, np.int16)
        src_d, dst_d = to_dev(src_h, dev), to_dev(dst_h, dev)
        off = np.arange(nb, dtype=np.int64) * n_ * n_
        for kind, kname in ((A.TR_DCT, "dct"), (A.TR_IDCT, "idct")):
            jb = jobs_np(nb, (off, off), (0, 0, 0, 1 if kind == A.TR_IDCT else 0))
            jd = jobs_dev(jb, dev)
            for mf in ((0, 1) if n_ >= 16 else (0,)):
                t = timeit(lambda: A.transform_batch(kind, 8, n_, A.plane(src_d, n_), A.plane(dst_d, n_), jd, nb, mf))
                tops = nb * 4 * n_ ** 3 * 2 / t / 1e12
                extra = f"MFMA int8 limbs: {tops:6.1f} TOPS = {tops / I8_PEAK_TOPS:.3f} of dense peak" if mf else "VALU"
                report(cpu, f"{kname}{n_} {'mfma' if mf else 'valu'}", nb, t, 2 * n_ * n_ * 2, f"cu[{int(np.log2(n_)) - 2}].{kname}",
                       SIG_DCT, [(src_h, n_), (dst_h, n_)], jb, extra)

    # ---- a8: quant family, 32x32 ----
    if not (on("quant") or on("intra") or on("intratu") or on("blockop") or on("loopfilter") or on("sao") or on("frame") or on("coeff")):
        return
    nb, n2 = 1 << 15, 1024
    coef_h = rng.integers(-255, 256, size=nb * n2, dtype=np.int16)
    qc_h = rng.integers(1, 256, size=nb * n2).astype(np.int32)
    du_h, qo_h = np.zeros(nb * n2, np.int32), np.zeros(nb * n2, np.int16)
    coef_d, qc_d, du_d, qo_d = (to_dev(x, dev) for x in (coef_h, qc_h, du_h, qo_h))
    res = torch.zeros(nb, dtype=torch.int32, device=dev)
    off = np.arange(nb, dtype=np.int64) * n2
    jb = jobs_np(nb, (off, off, off, off), (17, 85 << 8, n2))
    jd = jobs_dev(jb, dev)
    planes_d = [A.plane(coef_d), A.plane(qc_d), A.plane(du_d), A.plane(qo_d)]
    planes_h = [(coef_h, 0), (qc_h, 0), (du_h, 0), (qo_h, 0)]
    t = timeit(lambda: A.quant_batch(A.Q_QUANT, planes_d, jd, nb, res))
    report(cpu, "quant 32x32", nb, t, n2 * (2 + 4) * 2, "quant", SIG_QUANT, planes_h, jb)
    t = timeit(lambda: A.quant_batch(A.Q_NQUANT, planes_d, jd, nb, res))
    report(cpu, "nquant 32x32", nb, t, n2 * (2 + 4) + n2 * 2, "nquant", SIG_NQUANT, planes_h, jb)
    jb2 = jobs_np(nb, (off, off, off, off), (n2, 40, 5))
    jd2 = jobs_dev(jb2, dev)
    t = timeit(lambda: A.quant_batch(A.Q_DEQUANT_NORMAL, planes_d, jd2, nb, res))
    report(cpu, "dequant_normal 32x32", nb, t, n2 * 4, "dequant_normal", SIG_DEQUANT_NORMAL, planes_h, jb2)

    # ---- a12: intra prediction, all 35 modes per TU ----
    for n_ in (8, 16, 32):
        ntu = 1 << 11
        nb_h = rng.integers(0, 256, size=ntu * 160, dtype=np.uint8)
        dst_h = np.zeros(ntu * 35 * n_ * n_, np.uint8)
        nb_d, dst_d = to_dev(nb_h, dev), to_dev(dst_h, dev)
        tu = np.repeat(np.arange(ntu, dtype=np.int64), 35)
        mode = np.tile(np.arange(35, dtype=np.int64), ntu)
        jb = jobs_np(ntu * 35, (tu * 160, (tu * 35 + mode) * n_ * n_), (mode, 1))
        jd = jobs_dev(jb, dev)
        jb_cpu = jobs_np(ntu * 35, (tu * 160, (tu * 35 + mode) * n_ * n_), (np.maximum(mode, 2), 1))   # an angular slot only takes angular modes
        t = timeit(lambda: A.intra_batch(A.INTRA_PRED, 8, n_, A.plane(nb_d), A.plane(dst_d, n_), jd, ntu * 35))
        # one CPU job list per mode class is not needed: slot [mode] differs, time the angular slot 10 on all jobs
        report(cpu, f"intra_pred {n_}x{n_} (35 modes/TU)", ntu * 35, t, 4 * n_ + 1 + n_ * n_, f"cu[{int(np.log2(n_)) - 2}].intra_pred[10]",
               SIG_INTRA_PRED, [(nb_h, 0), (dst_h, n_)], jb_cpu, "CPU column: the angular slot on every job")

    # ---- (f)-2: the intra TU candidate set - prediction + residual round trip of every (TU, mode) candidate in one launch ----
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O  # noqa: E402  (bench.py's cpu_baseline leg)
    import time
    for n_ in (4, 8, 16, 32) if on("intratu") else ():
        ntu = {4: 1 << 13, 8: 1 << 12, 16: 1 << 11, 32: 1 << 9}[n_]
        nbw = 4 * n_ + 1
        fe_h = rng.integers(0, 256, size=ntu * n_ * n_, dtype=np.uint8)
        nb_h = np.clip(128 + rng.integers(-20, 21, size=ntu * 2 * nbw), 0, 255).astype(np.uint8)
        tu = np.repeat(np.arange(ntu, dtype=np.int64), 35)
        mode = np.tile(np.arange(35, dtype=np.int64), ntu)
        jb = jobs_np(ntu * 35, (tu * n_ * n_, tu * 2 * nbw, (tu * 2 + 1) * nbw, (tu * 35 + mode) * n_ * n_), (mode,))
        nj = ntu * 35
        fe_d, nb_d, jd = to_dev(fe_h, dev), to_dev(nb_h, dev), jobs_dev(jb, dev)
        rec_d = torch.zeros(nj * n_ * n_, dtype=torch.uint8, device=dev)
        lev_d = torch.zeros(nj * n_ * n_, dtype=torch.int16, device=dev)
        ns_d = torch.zeros(nj, dtype=torch.int32, device=dev)
        di_d = torch.zeros(nj, dtype=torch.int64, device=dev)
        t = timeit(lambda: A.intra_recon_batch(8, n_, fe_d, n_, nb_d, rec_d, n_, 27, 1, jd, nj, lev_d, ns_d, di_d))
        bpj = n_ * n_ * 3 + 2 * nbw + 12                 # recon + levels written, neighbours read; the source block is shared by 35 jobs
        col = f"{'-':>10s}"
        ratio = f"{'-':>8s}"
        if cpu.port is not None:
            samp = min(nj, 35 * 256)
            O.intra_recon(8, n_, fe_h, n_, nb_h, samp * n_ * n_, n_, 27, 1, jb[:samp], nthreads=cpu.threads, avx2=True)
            t0 = time.perf_counter()
            O.intra_recon(8, n_, fe_h, n_, nb_h, samp * n_ * n_, n_, 27, 1, jb[:samp], nthreads=cpu.threads, avx2=True)
            tc = (time.perf_counter() - t0) * nj / samp
            col, ratio = f"{nj * bpj / tc / 1e9:10.1f}", f"{tc / t:8.1f}"
        gbs = nj * bpj / t / 1e9
        print(f"{f'intra TU candidates {n_}x{n_} (35 modes)':40s} {nj:8d} {t * 1e6:9.1f} {bpj:7d} {gbs:9.1f} {gbs / HBM:6.3f} {'-':>10s} {col} {ratio}  "
              f"pred+DCT/DST+quant+dequant+IDCT+recon+SSE per job; CPU column = oracle restatement on a {35 * 256}-job sample", flush=True)

    # ---- a10: element-wise block ops, 32x32 ----
    nb, w = 1 << 15, 32
    pa_h = rng.integers(0, 256, size=nb * w * w, dtype=np.uint8)
    pb_h = rng.integers(0, 256, size=nb * w * w, dtype=np.uint8)
    s_h = rng.integers(-255, 256, size=nb * w * w, dtype=np.int16)
    s2_h = rng.integers(-8000, 8000, size=nb * w * w).astype(np.int16)
    pa_d, pb_d, s_d, s2_d = (to_dev(x, dev) for x in (pa_h, pb_h, s_h, s2_h))
    off = np.arange(nb, dtype=np.int64) * w * w
    jb = jobs_np(nb, (off, off, off))
    jd = jobs_dev(jb, dev)
    pu32 = pu_index(32, 32)
    cases = [
        ("copy_pp 32x32", A.OP_COPY_PP, [(pa_d, pa_h), (pb_d, pb_h), None], f"pu[{pu32}].copy_pp", SIG_COPY, 2 * w * w),
        ("sub_ps 32x32", A.OP_SUB_PS, [(s_d, s_h), (pa_d, pa_h), (pb_d, pb_h)], "cu[3].sub_ps", SIG_SUB_PS, 4 * w * w),
        ("add_ps 32x32", A.OP_ADD_PS, [(pa_d, pa_h), (pb_d, pb_h), (s_d, s_h)], "cu[3].add_ps[0]", SIG_ADD_PS, 4 * w * w),
        ("addAvg 32x32", A.OP_ADDAVG, [(pa_d, pa_h), (s2_d, s2_h), (s_d, s_h)], f"pu[{pu32}].addAvg[0]", SIG_ADDAVG, 5 * w * w),
    ]
    for name, op, pls, path, sig, bpj in cases:
        pd = [A.plane(p[0], w) if p else None for p in pls]
        t = timeit(lambda: A.blockop_batch(op, 8, w, w, pd, jd, nb))
        report(cpu, name, nb, t, bpj, path, sig, [(p[1], w) if p else (None, 0) for p in pls], jb)

    # ---- a13 / a14: SAO band offset on 64x64 CTUs, strong luma deblocking edges ----
    nb = 1 << 13
    rec_h = rng.integers(0, 256, size=nb * 64 * 64, dtype=np.uint8)
    offs_h = rng.integers(-7, 8, size=32).astype(np.int8)
    rec_d, offs_d = to_dev(rec_h, dev), to_dev(offs_h, dev)
    jb = jobs_np(nb, (np.arange(nb, dtype=np.int64) * 4096, 0), (64, 64))
    jd = jobs_dev(jb, dev)
    t = timeit(lambda: A.loopfilter_batch(A.LF_SAO_B0, 8, [A.plane(rec_d, 64), A.plane(offs_d)], jd, nb))
    report(cpu, "saoCuOrgB0 64x64", nb, t, 2 * 4096, "saoCuOrgB0", SIG_SAO_B0, [(rec_h, 64), (offs_h, 0)], jb)
    ne = 1 << 18
    edge = (rng.integers(8, Hh - 8, size=ne) * st + rng.integers(8, W - 8, size=ne)).astype(np.int64)
    jb = jobs_np(ne, (edge,), (st, 1, 3, 3))           # vertical edge: 4 lines down the plane, taps along x
    jd = jobs_dev(jb, dev)
    t = timeit(lambda: A.loopfilter_batch(A.LF_DEBLOCK_LUMA_STRONG, 8, [A.plane(ref_d, st)], jd, ne))
    report(cpu, "pelFilterLumaStrong (4 lines)", ne, t, 4 * 8 + 4 * 6, "pelFilterLumaStrong[0]", SIG_DEBLOCK_LUMA, [(ref_h, st)], jb)

    # ---- a16 / a9 (round 6): frame-level helpers on 4K planes, RDOQ helpers over 2^17 coefficient groups.  CPU columns: the reference build's own slot (ref-C) and
    # the oracle's AVX2 restatement (port) - whole planes are ONE call of the slot (single thread, as the reference calls them: picyuv.cpp, lowres.cpp, slicetype.cpp);
    # the RDOQ helpers run over the same job list on all host threads (oracle/x265_oracle_bench.c: x265oracle_time_coeff_jobs)
    if on("frame") or on("coeff"):
        import time
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import harness_host as HH        # noqa: E402  (TEST INFRASTRUCTURE: scan tables / coefficient recipes of the reference's harnesses)
        ref_h8 = H.load_reference(8, ROOT)
        orc8 = H.load_oracle(8, ROOT, avx2=True, host=True)
        tabs = [t for t in (ref_h8, orc8) if t is not None]

        def cpu_cols(path, call, nbytes, reps=5, tables=None):
            cols, best = [], None
            for tab in (tables or (ref_h8, orc8)):
                if tab is None or not tab.ptr(path):
                    cols.append(f"{'-':>10s}"); continue
                f_ = tab.fn(path)
                call(f_)
                t0 = time.perf_counter()
                for _ in range(reps):
                    call(f_)
                tc = (time.perf_counter() - t0) / reps
                cols.append(f"{nbytes / tc / 1e9:10.2f}")
                best = tc if best is None or tc < best else best
            return cols, best
    if on("frame"):
        pw, ph, pst, NP = 3840, 2160, 3840 + 192, 24          # 24 planes per launch: 0.4 - 0.8 GB of traffic, past the 256 MiB Infinity Cache
        src8 = rng.integers(0, 256, size=pst * ph, dtype=np.uint8)
        src16 = rng.integers(0, 1024, size=pst * ph).astype(np.uint16)
        d8, d16 = to_dev(np.tile(src8, NP), dev), to_dev(np.tile(src16, NP), dev)
        o8, o16 = torch.zeros(NP * pst * ph, dtype=torch.uint8, device=dev), torch.zeros(NP * pst * ph, dtype=torch.int16, device=dev)
        out8, out16 = np.zeros(pst * ph, np.uint8), np.zeros(pst * ph, np.uint16)
        P2 = lambda a, b: [A.Plane(a.data_ptr(), pst), A.Plane(b.data_ptr(), pst)]
        orc10 = H.load_oracle(10, ROOT, avx2=True, host=True)
        ref10 = H.load_reference(10, ROOT)
        for name, kind, depth, src_d, dst_d, path, call, bpp, args_, tabs10 in (
                ("planecopy_cp 4K planes (u8 -> u8)", A.FR_PLANECOPY_CP, 8, d8, o8, "planecopy_cp", lambda f_: f_(H.ptr(src8), pst, H.ptr(out8), pst, pw, ph, 0), 2, (0, 0), False),
                ("planecopy_sp 4K planes (u16 -> u8)", A.FR_PLANECOPY_SP, 8, d16, o8, "planecopy_sp", lambda f_: f_(H.ptr(src16), pst, H.ptr(out8), pst, pw, ph, 2, 255), 3, (2, 255), False),
                ("planecopy_sp 4K planes (u16 -> 10 bit)", A.FR_PLANECOPY_SP, 10, d16, o16, "planecopy_sp", lambda f_: f_(H.ptr(src16), pst, H.ptr(out16), pst, pw, ph, 0, 1023), 4, (0, 1023), True),
                ("planecopy_pp_shr 4K planes", A.FR_PLANECOPY_PP_SHR, 8, d8, o8, "planecopy_pp_shr", lambda f_: f_(H.ptr(src8), pst, H.ptr(out8), pst, pw, ph, 2), 2, (2, 0), False)):
            jd = jobs_dev(jobs_np(NP, (np.arange(NP, dtype=np.int64) * pst * ph,) * 2, args_), dev)
            t = timeit(lambda: A.frame_batch(kind, depth, pw, ph, P2(src_d, dst_d), jd, NP))
            nbytes = bpp * pw * ph
            cols, best = cpu_cols(path, call, nbytes, tables=(ref10, orc10) if tabs10 else None)
            gbs = NP * nbytes / t / 1e9
            print(f"{name:40s} {NP:8d} {t * 1e6:9.1f} {nbytes:7d} {gbs:9.1f} {gbs / HBM:6.3f} {cols[0]} {cols[1]} {NP * best / t if best else 0:8.1f}  planes; bytes = one read + one write; CPU = one call per plane, one thread", flush=True)
        if orc10 is not None:          # planeClipAndMax exists in the high-bit-depth builds only
            clip = to_dev(np.tile(src16, NP), dev)
            outc = torch.zeros(2 * NP, dtype=torch.int64, device=dev)
            jd = jobs_dev(jobs_np(NP, (np.arange(NP, dtype=np.int64) * pst * ph,), (64, 940)), dev)
            t = timeit(lambda: A.frame_batch(A.FR_PLANE_CLIP_MAX, 10, pw, ph, P2(clip, clip), jd, NP, outc))
            tot = np.zeros(1, np.uint64); work = src16.copy()
            cols, best = cpu_cols("planeClipAndMax", lambda f_: f_(H.ptr(work), pst, pw, ph, H.ptr(tot), 64, 940), 4 * pw * ph, tables=(ref10, orc10))
            gbs = NP * 4 * pw * ph / t / 1e9
            print(f"{'planeClipAndMax 4K planes (10 bit)':40s} {NP:8d} {t * 1e6:9.1f} {4 * pw * ph:7d} {gbs:9.1f} {gbs / HBM:6.3f} {cols[0]} {cols[1]} {NP * best / t if best else 0:8.1f}  planes; clamp in place + maximum + sum", flush=True)
        # frameInitLowres 4K -> four 1920 x 1080 planes
        lw, lh, lst = 1920, 1080, 1920 + 64
        lo = [torch.zeros(lst * lh, dtype=torch.uint8, device=dev) for _ in range(4)]
        lo_h = [np.zeros(lst * lh, np.uint8) for _ in range(4)]
        srcL = rng.integers(0, 256, size=pst * (ph + 2), dtype=np.uint8)
        dL = to_dev(srcL, dev)
        t = timeit(lambda: A.frame_init_lowres(8, dL, 0, pst, lo, lst, lw, lh))
        nbytes = pw * ph + 4 * lw * lh
        cols, best = cpu_cols("frameInitLowres", lambda f_: f_(H.ptr(srcL), *[H.ptr(x) for x in lo_h], pst, lst, lw, lh), nbytes)
        gbs = nbytes / t / 1e9
        print(f"{'frameInitLowres 4K -> 4 x 1080p':40s} {1:8d} {t * 1e6:9.1f} {nbytes:7d} {gbs:9.1f} {gbs / HBM:6.3f} {cols[0]} {cols[1]} {best / t if best else 0:8.1f}  plane; bytes = source once + four planes", flush=True)
        # propagateCost over a 4K picture's 240 x 135 lowres blocks, ssim moments of a 4K plane
        nblk = 240 * 135
        pin, inter = rng.integers(0, 65536, size=nblk).astype(np.uint16), rng.integers(0, 65536, size=nblk).astype(np.uint16)
        intra, invq = rng.integers(1, 1 << 15, size=nblk).astype(np.int32), rng.integers(1, 1 << 15, size=nblk).astype(np.int32)
        fps = np.array([64.0]); outp = np.zeros(nblk, np.int32)
        dp = [to_dev(x, dev) for x in (pin, intra, inter, invq)]
        dd = torch.zeros(nblk, dtype=torch.int32, device=dev)
        t = timeit(lambda: A.propagate_cost(dd, dp[0], dp[1], dp[2], dp[3], 64.0, nblk))
        nbytes = nblk * 16
        cols, best = cpu_cols("propagateCost", lambda f_: f_(H.ptr(outp), H.ptr(pin), H.ptr(intra), H.ptr(inter), H.ptr(invq), H.ptr(fps), nblk), nbytes, reps=20)
        print(f"{'propagateCost 240 x 135 blocks':40s} {nblk:8d} {t * 1e6:9.1f} {16:7d} {nbytes / t / 1e9:9.1f} {nbytes / t / 1e9 / HBM:6.3f} {cols[0]} {cols[1]} {best / t if best else 0:8.1f}  launch-latency bound at this size", flush=True)
        pairs = [(y, x) for y in range(ph // 4) for x in range(0, pw // 4 - 1, 2)]
        jb = jobs_np(len(pairs), (np.array([4 * y * pst + 4 * x for y, x in pairs], np.int64),) * 2)
        jd = jobs_dev(jb, dev)
        osum = torch.zeros(len(pairs) * 8, dtype=torch.int32, device=dev)
        d8b = to_dev(np.clip(src8.astype(np.int32) + 3, 0, 255).astype(np.uint8), dev)
        t = timeit(lambda: A.frame_batch(A.FR_SSIM_CORE, 8, 8, 4, [A.Plane(d8.data_ptr(), pst), A.Plane(d8b.data_ptr(), pst)], jd, len(pairs), osum))
        nbytes = len(pairs) * 64
        print(f"{'ssim_4x4x2_core over a 4K plane':40s} {len(pairs):8d} {t * 1e6:9.1f} {64:7d} {nbytes / t / 1e9:9.1f} {nbytes / t / 1e9 / HBM:6.3f} {'-':>10s} {'-':>10s} {'-':>8s}  8x4 samples of two planes per job", flush=True)
    if on("coeff"):
        A.set_entropy_bits(H.host_tables(ROOT)["entropy_bits"])
        blib = cpu.lib
        blib.x265oracle_time_coeff_jobs.restype = ctypes.c_double
        blib.x265oracle_time_coeff_jobs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        nj = 1 << 17
        cdt = np.dtype([("off", "<i8", 5), ("arg", "<i4", 5), ("reserved", "<i4")])
        scan16 = np.tile(HH.diag_scan(4), (nj, 1))
        coeff = rng.integers(-300, 301, size=(nj, 16)).astype(np.int16)
        coeff[rng.random((nj, 16)) < 0.6] = 0
        coeff[:, 0] |= 1
        tabc = rng.integers(0, 9, size=(nj, 16)).astype(np.uint8)
        ctx = rng.integers(2, 125, size=(nj, 64)).astype(np.uint8)
        absb = rng.integers(1, 40, size=(nj, 24)).astype(np.uint16)
        masks = np.array([int("".join("1" if coeff[j, int(q)] else "0" for q in HH.diag_scan(4)), 2) for j in range(nj)], np.int64)
        idx = np.arange(nj, dtype=np.int64)

        def cj(offs, args):
            a = np.zeros(nj, cdt)
            for k, o in enumerate(offs): a["off"][:, k] = o
            for k, v in enumerate(args): a["arg"][:, k] = v
            return a
        cases = [
            ("costCoeffNxN (16 positions)", A.CF_COST_COEFF_NXN, "costCoeffNxN", cj((16 * idx, 16 * idx, 24 * idx, 16 * idx, 64 * idx), (4, masks, 12, 15, 16)), 128),
            ("costC1C2Flag (8 flags)", A.CF_COST_C1C2, "costC1C2Flag", cj((0, 0, 24 * idx, 0, 64 * idx), (8, 24)), 32),
            ("costCoeffRemain (16 levels)", A.CF_COST_COEFF_REMAIN, "costCoeffRemain", cj((0, 0, 24 * idx), (16, 0)), 32),
            ("findPosFirstLast", A.CF_FIND_POS_FIRST_LAST, "findPosFirstLast", cj((16 * idx, 16 * idx), (4,)), 64),
        ]
        bufs_h = [scan16.reshape(-1), coeff.reshape(-1), absb.reshape(-1), tabc.reshape(-1), ctx.reshape(-1)]
        bufs_d = [to_dev(b.copy(), dev) for b in bufs_h]
        res = torch.zeros(nj, dtype=torch.int32, device=dev)
        for name, kind, path, jobs, bpj in cases:
            jd = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
            t = timeit(lambda: A.coeff_batch(kind, 8, bufs_d, jd, nj, res))
            cols, best = [], None
            for tab in (ref_h8, orc8):
                if tab is None or not tab.ptr(path):
                    cols.append(f"{'-':>10s}"); continue
                hb = [b.copy() for b in bufs_h]
                ptrs = (ctypes.c_void_p * 5)(*[b.ctypes.data for b in hb])
                jj = np.ascontiguousarray(jobs)
                tc = blib.x265oracle_time_coeff_jobs(tab.ptr(path), kind, ptrs, jj.ctypes.data, nj, 5, cpu.threads, None)
                cols.append(f"{nj * bpj / tc / 1e9:10.2f}")
                best = tc if best is None or tc < best else best
            gbs = nj * bpj / t / 1e9
            print(f"{name:40s} {nj:8d} {t * 1e6:9.1f} {bpj:7d} {gbs:9.1f} {gbs / HBM:6.3f} {cols[0]} {cols[1]} {best / t if best else 0:8.1f}  calls; {nj / t / 1e6:.0f} M calls/s on the GPU, "
                  f"{nj / best / 1e6 if best else 0:.0f} M calls/s on {cpu.threads} host threads; serial in the CABAC state per call", flush=True)
        # scanPosLast: 2^13 32x32 TUs
        nt = 1 << 13
        scan32, inner = HH.block_scan(rng, 32, 0)
        cf = rng.integers(-300, 301, size=(nt, 1024)).astype(np.int16)
        cf[rng.random((nt, 1024)) < 0.9] = 0
        cf[:, 5] = 7
        nsig = np.count_nonzero(cf, axis=1)
        a = np.zeros(nt, cdt)
        ti = np.arange(nt, dtype=np.int64)
        a["off"][:, 1], a["off"][:, 2], a["off"][:, 3], a["off"][:, 4] = 1024 * ti, 64 * ti, 64 * ti, 64 * ti
        a["arg"][:, 0], a["arg"][:, 1] = nsig, 32
        hb = [scan32.copy(), cf.reshape(-1), np.zeros(nt * 64, np.uint16), np.zeros(nt * 64, np.uint16), np.zeros(nt * 64, np.uint8)]
        db = [to_dev(b.copy(), dev) for b in hb]
        jd = torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
        res = torch.zeros(nt, dtype=torch.int32, device=dev)
        t = timeit(lambda: A.coeff_batch(A.CF_SCAN_POS_LAST, 8, db, jd, nt, res))
        cols, best = [], None
        for tab in (ref_h8, orc8):
            if tab is None:
                cols.append(f"{'-':>10s}"); continue
            ptrs = (ctypes.c_void_p * 5)(*[b.ctypes.data for b in hb])
            tc = blib.x265oracle_time_coeff_jobs(tab.ptr("scanPosLast"), 0, ptrs, a.ctypes.data, nt, 5, cpu.threads, None)
            cols.append(f"{nt * 2368 / tc / 1e9:10.2f}")
            best = tc if best is None or tc < best else best
        gbs = nt * 2368 / t / 1e9
        print(f"{'scanPosLast 32x32 TU (10 % non-zero)':40s} {nt:8d} {t * 1e6:9.1f} {2368:7d} {gbs:9.1f} {gbs / HBM:6.3f} {cols[0]} {cols[1]} {best / t if best else 0:8.1f}  TUs; bytes = coefficients + the three 64-entry outputs; a wavefront per TU", flush=True)

    # ---- (f)-4: frame-level SAO passes on a 4K picture (statistics for every CTU / type / class; offsets applied out of place) ----
    if on("loopfilter") or on("sao"):
        import time
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_api as O  # noqa: E402  (bench.py's cpu_baseline leg)
        F = importlib.import_module("x265-yuuki-asuna_amd.frames")
        sw, sh = 3840, 2160
        yv = F.synth_clip(sw, sh, 1, depth=8, seed=3)[0][0]
        rc = np.clip(yv.astype(np.int32) + rng.integers(-3, 4, size=yv.shape), 0, 255).astype(np.uint8)
        fp, fst, forg, _, _ = F.pad_plane(yv)
        rp = F.pad_plane(rc)[0]
        nctu = ((sw + 63) // 64) * ((sh + 63) // 64)
        par = np.zeros((nctu, 7), np.int32)
        par[:, 0] = rng.integers(0, 5, size=nctu); par[:, 1] = rng.integers(0, 32, size=nctu); par[:, 2:6] = rng.integers(-7, 8, size=(nctu, 4))
        d_f, d_r = to_dev(fp.reshape(-1), dev), to_dev(rp.reshape(-1), dev)
        d_o = d_r.clone()
        d_c = torch.zeros(nctu * 160, dtype=torch.int32, device=dev)
        d_s = torch.zeros(nctu * 160, dtype=torch.int32, device=dev)
        d_p = torch.from_numpy(par.reshape(-1)).to(dev)
        for name, fn, bpp, cpu_fn in (
                ("sao stats 4K frame (5 types/CTU)", lambda: A.sao_stats(8, d_f, fst, forg, d_r, fst, forg, sw, sh, d_c, d_s), 2,
                 lambda: O.sao_stats(8, fp, rp, fst, forg, sw, sh, nthreads=cpu.threads, avx2=True)),
                ("sao apply 4K frame", lambda: A.sao_apply(8, d_r, fst, forg, d_o, fst, forg, sw, sh, d_p), 2,
                 lambda: O.sao_apply(8, rp, fst, forg, sw, sh, par, nthreads=cpu.threads, avx2=True))):
            t = timeit(fn)
            col, ratio = f"{'-':>10s}", f"{'-':>8s}"
            if cpu.port is not None:
                cpu_fn()
                t0 = time.perf_counter(); cpu_fn(); tc = time.perf_counter() - t0
                col, ratio = f"{sw * sh * bpp / tc / 1e9:10.1f}", f"{tc / t:8.1f}"
            gbs = sw * sh * bpp / t / 1e9
            print(f"{name:40s} {nctu:8d} {t * 1e6:9.1f} {4096 * bpp:7d} {gbs:9.1f} {gbs / HBM:6.3f} {'-':>10s} {col} {ratio}  "
                  f"CTUs; bytes = the two planes touched once; CPU column = oracle restatement on {cpu.threads} threads", flush=True)


if __name__ == "__main__":
    main()
