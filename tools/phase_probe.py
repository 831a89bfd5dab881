"""Times x265hip_phase_planes alone (HIP events) at a BASELINE size: luma (15 planes) and one chroma plane (63 planes).  Measurement aid."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def main():
    import torch
    dev = torch.device("cuda:0")
    for depth in (8, 10):
        es = 1 if depth == 8 else 2
        for name, st, rw, chroma in (("luma 4032x2336", 4032, 2336, False), ("chroma 2112x1168", 2112, 1168, True)):
            nb = st * rw * es
            glo, ghi = 4 * st * es + 64, 8 * st * es
            src = torch.randint(0, 255, (glo + nb + ghi,), dtype=torch.uint8, device=dev)
            nph = 63 if chroma else 15
            dst = torch.zeros(nph * nb, dtype=torch.uint8, device=dev)
            for _ in range(2):
                A.phase_planes(depth, src, glo, dst, st, rw, chroma=chroma)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                A.phase_planes(depth, src, glo, dst, st, rw, chroma=chroma)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"phase_planes {depth}-bit {name}: {ms:.3f} ms, {nph * nb / 1e6:.0f} MB written -> {nph * nb / ms / 1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
