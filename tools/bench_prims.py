#!/usr/bin/env python3
"""Per-family throughput of the batch-layer kernels on one MI355X (inputs resident in HBM).

For every family: N blocks per launch, HIP-event time per launch, algorithmic bytes per block as defined in
SURVEY.md section 8(d), achieved GB/s and the fraction of the 8 TB/s HBM peak; for the MFMA transforms also
int8 TOPS (4*N^3 MACs per block incl. the two 8-bit limbs) against the 5 POPS dense int8 peak.
Prints a table; `> profiles/rNN_prims.txt` keeps it."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")

HBM = 8000.0
I8_PEAK_TOPS = 5000.0


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


def row(name, njobs, t, bytes_per_job, extra=""):
    gbs = njobs * bytes_per_job / t / 1e9
    print(f"{name:44s} {njobs:9d} {t * 1e6:10.1f} {bytes_per_job:9d} {gbs:10.1f} {gbs / HBM:7.3f} {extra}")


def main():
    import torch
    dev = "cuda:0"
    rng = np.random.default_rng(5)
    print(f"{'kernel (8-bit unless noted)':44s} {'blocks':>9s} {'us/launch':>10s} {'B/block':>9s} {'GB/s':>10s} {'frac':>7s}")

    # ---- pixel compare on a 4K frame: random candidate positions (fenc blocks at stride 64 like motion.cpp) ----
    W, Hh, st = 3840, 2176, 4032
    ref = torch.from_numpy(rng.integers(0, 256, size=st * (Hh + 160), dtype=np.uint8)).to(dev)
    for kind, name in ((A.CMP_SAD, "sad"), (A.CMP_SATD, "satd"), (A.CMP_SA8D, "sa8d")):
        for w in (8, 16, 32, 64):
            if kind == A.CMP_SA8D and w < 8:
                continue
            n = 1 << 20 if w <= 16 else 1 << 18
            fenc = torch.from_numpy(rng.integers(0, 256, size=n * 64 + 64 * 64, dtype=np.uint8)).to(dev)
            aoff = torch.arange(n, dtype=torch.int64, device=dev) * 64 % (n * 64 - 64 * 64)
            boff = torch.from_numpy((rng.integers(80, Hh - 80, size=n) * st + rng.integers(96, W - 96, size=n)).astype(np.int64)).to(dev)
            out = torch.zeros(n, dtype=torch.int64, device=dev)
            t = timeit(lambda: A.pixelcmp_batch(kind, 8, w, w, fenc, 64, ref, st, n, out, a_off=aoff, b_off=boff))
            row(f"pixelcmp {name} {w}x{w}", n, t, 2 * w * w + 8)

    # ---- interpolation: hvpp 16x16 / 64x64 luma ----
    for w in (16, 64):
        n = 1 << 18 if w == 16 else 1 << 15
        jobs = A.make_jobs([([int(y) * st + int(x), j * w * w], [int(ix), int(iy)]) for j, (y, x, ix, iy) in
                            enumerate(zip(rng.integers(80, Hh - 80, size=n), rng.integers(96, W - 96, size=n),
                                          rng.integers(1, 4, size=n), rng.integers(1, 4, size=n)))], dev)
        dst = torch.zeros(n * w * w, dtype=torch.uint8, device=dev)
        t = timeit(lambda: A.interp_batch(A.IP_HVPP, 8, 8, w, w, A.plane(ref, st), A.plane(dst, w), jobs, n))
        row(f"interp luma_hvpp {w}x{w}", n, t, (w + 7) * (w + 7) + w * w)

    # ---- transforms: VALU vs MFMA ----
    for n_ in (16, 32):
        nb = 1 << 17
        src = torch.from_numpy(rng.integers(-255, 256, size=nb * n_ * n_, dtype=np.int16)).to(dev)
        dst = torch.zeros(nb * n_ * n_, dtype=torch.int16, device=dev)
        jobs = A.make_jobs([([j * n_ * n_, j * n_ * n_], []) for j in range(nb)], dev)
        for kind, kname in ((A.TR_DCT, "dct"), (A.TR_IDCT, "idct")):
            for mf in (0, 1):
                t = timeit(lambda: A.transform_batch(kind, 8, n_, A.plane(src, n_), A.plane(dst, n_), jobs, nb, mf))
                tops = nb * 4 * n_ ** 3 * 2 / t / 1e12
                extra = f"int8 {tops:7.1f} TOPS = {tops / I8_PEAK_TOPS:.3f} of dense peak" if mf else "VALU"
                row(f"transform {kname}{n_} {'mfma' if mf else 'valu'}", nb, t, 2 * n_ * n_ * 2, extra)

    # ---- quant 32x32 ----
    nb, n2 = 1 << 16, 1024
    coef = torch.from_numpy(rng.integers(-255, 256, size=nb * n2, dtype=np.int16)).to(dev)
    qc = torch.from_numpy(rng.integers(1, 256, size=nb * n2).astype(np.int32)).to(dev)
    du = torch.zeros(nb * n2, dtype=torch.int32, device=dev)
    qo = torch.zeros(nb * n2, dtype=torch.int16, device=dev)
    res = torch.zeros(nb, dtype=torch.int32, device=dev)
    jobs = A.make_jobs([([j * n2] * 4, [17, 85 << 8, n2]) for j in range(nb)], dev)
    t = timeit(lambda: A.quant_batch(A.Q_QUANT, [A.plane(coef), A.plane(qc), A.plane(du), A.plane(qo)], jobs, nb, res))
    row("quant 32x32", nb, t, n2 * (2 + 4) * 2)

    # ---- intra 16x16, all 35 modes ----
    ntu = 1 << 13
    nbuf = torch.from_numpy(rng.integers(0, 256, size=ntu * 80, dtype=np.uint8)).to(dev)
    dsti = torch.zeros(ntu * 35 * 256, dtype=torch.uint8, device=dev)
    jobs = A.make_jobs([([t_ * 80, (t_ * 35 + m) * 256], [m, 1]) for t_ in range(ntu) for m in range(35)], dev)
    t = timeit(lambda: A.intra_batch(A.INTRA_PRED, 8, 16, A.plane(nbuf), A.plane(dsti, 16), jobs, ntu * 35))
    row("intra_pred 16x16 (35 modes per TU)", ntu * 35, t, 65 + 256)


if __name__ == "__main__":
    main()
