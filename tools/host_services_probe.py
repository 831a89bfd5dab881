#!/usr/bin/env python3
"""Wall time of the two host-pointer stage services at 4K against the CPU restatements of the loops they replace (the oracle's C, same primitives
as the reference's C build) - what a frame encoder / lookahead thread waits for per call.

  python tools/host_services_probe.py [--depth 8] [--reps 5]            (under rocprofv3 --kernel-trace --stats for the kernel table)
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import oracle_api as O
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    import test_gpu_weight_analyse as TW
    depth, W, H = a.depth, 3840, 2176
    # ---- weightAnalyse: a P slice (one list) and a B slice (two lists) on a fade, lowres vectors present
    for nlists in (1, 2):
        cur, refs, intra, lm, cm = TW._case(depth, W, H, nlists, 0.85, 5, True, seed=11)
        t0 = time.perf_counter(); want = O.weight_analyse(depth, cur, refs, W, H, intra); cpu = time.perf_counter() - t0
        A.weight_analyse_host(depth, cur, refs, W, H, intra, lm, cm)          # first call: scratch allocation
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); got = A.weight_analyse_host(depth, cur, refs, W, H, intra, lm, cm); ts.append(time.perf_counter() - t0)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        keys = [(99 << 32) | 1, (99 << 32) | 2, (99 << 32) | 3]
        A.weight_analyse_host(depth, cur, refs, W, H, intra, lm, cm, plane_keys=keys)
        tk = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); A.weight_analyse_host(depth, cur, refs, W, H, intra, lm, cm, plane_keys=keys); tk.append(time.perf_counter() - t0)
        A.lib().x265hip_lowres_planes_forget()
        print(f"weightAnalyse {W}x{H} {depth}-bit, {nlists} list(s), all three planes analysed: CPU restatement {cpu * 1e3:.0f} ms   x265hip_weight_analyse_host "
              f"{min(ts) * 1e3:.1f} ms (median {sorted(ts)[len(ts) // 2] * 1e3:.1f}; planes uploaded on every call)   {min(tk) * 1e3:.1f} ms with the lowres planes keyed "
              f"(resident)   weights {got[0][:nlists, :, :].tolist()}")
    # ---- calcAdaptiveQuantFrame
    yimg, cbimg, crimg = F.synth_clip(W, 2160, 1, depth=depth, seed=5)[0]
    pad = lambda img, m: (lambda b: (b.reshape(-1), b.shape[1], m * b.shape[1] + m))(np.ascontiguousarray(np.pad(img, m, mode="edge")))
    yp, cbp, crp = pad(np.ascontiguousarray(yimg), 32), pad(np.ascontiguousarray(cbimg), 16), pad(np.ascontiguousarray(crimg), 16)
    for qg in (16, 8):
        kw = dict(cb=cbp[0], cr=crp[0], stride_c=cbp[1], org_c=cbp[2])
        t0 = time.perf_counter(); want = O.aq_frame(depth, yp[0], yp[1], yp[2], W, 2160, qg_size=qg, aq_mode=2, aq_strength=1.0, weightp=True, **kw); cpu = time.perf_counter() - t0
        A.aq_frame_host(depth, yp[0], yp[1], yp[2], W, 2160, qg, 2, 1.0, **kw)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); got = A.aq_frame_host(depth, yp[0], yp[1], yp[2], W, 2160, qg, 2, 1.0, **kw); ts.append(time.perf_counter() - t0)
        assert np.array_equal(got["qp_aq_offset"], want[1]) and np.array_equal(got["inv_qscale"], want[2])
        print(f"calcAdaptiveQuantFrame {W}x2160 {depth}-bit qg {qg}, aq-mode 2: CPU restatement {cpu * 1e3:.1f} ms   x265hip_aq_frame_host {min(ts) * 1e3:.1f} ms "
              f"(median {sorted(ts)[len(ts) // 2] * 1e3:.1f}; 12.4 MB x {1 if depth == 8 else 2} of planes uploaded from pageable memory per call, the double-precision pass on the caller)")


if __name__ == "__main__":
    main()
