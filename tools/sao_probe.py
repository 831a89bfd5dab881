#!/usr/bin/env python3
"""Host-enqueue and device time of the SAO passes of one band / one picture: x265hip_sao_planes (three planes per launch) against the
per-plane x265hip_sao_stats / _decide / _apply.  python tools/sao_probe.py [rows]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")


def main():
    dev = torch.device("cuda:0")
    for rows in [int(a) for a in sys.argv[1:]] or [256, 2176]:
        W, stride = 3840, 4096
        geo = [(W, rows, (64, 64), 0, stride), (W // 2, rows // 2, (32, 32), 2, stride // 2), (W // 2, rows // 2, (32, 32), 2, stride // 2)]
        sa, src, rec, out = [], [], [], []
        for w, h, ctu, po, st in geo:
            sa.append(S.Sao(w, h, 8, dev, ctu=ctu, plane_offset=po))
            src.append(torch.randint(0, 255, (st * (h + 160),), dtype=torch.uint8, device=dev))
            rec.append((src[-1].to(torch.int16) + torch.randint(-3, 4, src[-1].shape, device=dev, dtype=torch.int16)).clamp(0, 255).to(torch.uint8))
            out.append(torch.zeros_like(rec[-1]))
        org = [g[4] * 80 + 80 for g in geo]

        def fused():
            H.sao_planes(8, [sa[i].plane(src[i], geo[i][4], org[i], rec[i], geo[i][4], org[i], out[i]) for i in range(3)])

        def single():
            for i in range(3):
                sa[i].stats(None, rec[i], geo[i][4], org[i], src_plane=src[i])
            for i in range(3):
                sa[i].decide()
                sa[i].apply(rec[i], geo[i][4], org[i], out[i])

        def fused_one_by_one():
            for i in range(3):
                H.sao_planes(8, [sa[i].plane(src[i], geo[i][4], org[i], rec[i], geo[i][4], org[i], out[i])])

        import numpy as np
        seq = [("fused", fused), ("fused", fused), ("single", single), ("fused", fused), ("fused", fused), ("planes x1", fused_one_by_one), ("fused", fused),
               ("single", single), ("fused", fused)]
        for name, fn in seq:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n = 300
            ts = np.zeros(n + 1)
            ts[0] = time.perf_counter()
            for k in range(n):
                fn()
                ts[k + 1] = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            d = np.diff(ts) * 1e6
            print(f"rows {rows} {name}: host enqueue mean {d.mean():.1f} median {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} max {d.max():.1f} us / call, "
                  f"with the device {1e6 * (t2 - ts[0]) / n:.1f} us / call", flush=True)


if __name__ == "__main__":
    main()
