"""A/B timing of the two 8-bit exhaustive-search kernels (X265HIP_ME_KERNEL=rows: me_ctu_q_kernel, default: me_ctu_c_kernel) for
the three output modes.  Prints ms per launch (HIP events, 20 launches after 3 warm-ups).  Measurement aid."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")


def main():
    import torch
    dev = torch.device("cuda:0")
    w, h, rng = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160)), int(os.environ.get("R", 57))
    clip = F.synth_clip(w, h, 2, depth=8, seed=265)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    modes = [("surf(packed)+best", dict(packed=True)), ("surf(packed)", dict(packed=True, want_best=False)), ("best", dict(want_surf=False)),
             ("surf(i32)+best", dict())]
    if os.environ.get("ME_AB_MODES") == "t" or os.environ.get("ME_AB_BOTH"):
        modes = [("surf(packed_t)+best", dict(packed="t")), ("surf(packed_t)", dict(packed="t", want_best=False)), ("best", dict(want_surf=False))]
    if os.environ.get("ME_AB_BOTH"):          # the row-walking kernel's packed launch next to the record-per-lane kernel's chunk-major one
        modes += [("surf(packed)+best", dict(packed=True)), ("surf(packed)", dict(packed=True, want_best=False))]
    for name, kw in modes:
        ms = P.MotionSearch(cur.w64, cur.h64, rng, 8, dev, **kw)
        for _ in range(3):
            ms.run(cur, ref)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = 0.0
        for _ in range(20):
            ms.reset()
            e0.record()
            ms.search(cur, ref)
            e1.record()
            torch.cuda.synchronize()
            t += e0.elapsed_time(e1)
        print(f"{os.environ.get('X265HIP_ME_KERNEL', 'auto'):5s} var={os.environ.get('X265HIP_ME_CAND_VARIANT', '0')} {w}x{h} R={rng} {name:18s}: {t / 20:.3f} ms", flush=True)
        del ms
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
