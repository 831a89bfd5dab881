#!/usr/bin/env python3
"""Where a banded 4K frame's time goes: host enqueue time against device time of stages.BandedFramePipeline, with the SAO passes fused
(x265hip_sao_planes) or per plane, and the time the host spends inside every hipabi entry.  python tools/band_probe.py [band_rows]"""
import collections
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def main():
    band_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    W, Hh = 3840, 2160
    clip = F.synth_clip(W, Hh, 4, depth=8, seed=265)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    bp = S.BandedFramePipeline(pics[0].w64, pics[0].h64, 8, dev, band_rows=band_rows, rng=57, subme=3, level=2, qp=30, want_surf=True, packed=True,
                               lookahead=(W, Hh), deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True)
    ref = pics[0].like([p.clone() for p in pics[0].planes()])
    spent = collections.Counter()
    calls = collections.Counter()
    for name in dir(A):
        fn = getattr(A, name)
        if callable(fn) and not isinstance(fn, type) and getattr(fn, "__module__", "") == A.__name__ and name not in ("lib", "check", "current_stream", "_p"):
            def wrap(fn=fn, name=name):
                def inner(*a, **k):
                    t = time.perf_counter()
                    try:
                        return fn(*a, **k)
                    finally:
                        spent[name] += time.perf_counter() - t
                        calls[name] += 1
                return inner
            setattr(A, name, wrap())

    def frame(i):
        cur = pics[1 + i % 3]
        bp.run(cur, ref)
        for d, s in zip(ref.planes(), bp.final_planes()):
            d.copy_(s)

    for fuse in (False, True, False, True):
        for pipe in bp.pipes.values():
            pipe.fuse_sao = fuse
        for i in range(3):
            frame(i)
        torch.cuda.synchronize()
        spent.clear(); calls.clear()
        n = 30
        t0 = time.perf_counter()
        for i in range(n):
            frame(3 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"bands of {band_rows} rows, fused SAO {fuse}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms / frame, with the device {1e3 * (t2 - t0) / n:.3f} ms / frame; "
              f"inside hipabi {1e3 * sum(spent.values()) / n:.3f} ms in {sum(calls.values()) // n} calls", flush=True)
        top = sorted(spent.items(), key=lambda kv: -kv[1])[:8]
        print("   " + ", ".join(f"{k} {1e6 * v / calls[k]:.1f} us x {calls[k] // n}" for k, v in top), flush=True)


if __name__ == "__main__":
    main()
