#!/usr/bin/env python3
"""Randomised soak of the banded pipeline's launch structures: for random picture sizes / depths / band sizes / search formats, a closed loop
of frames through stages.BandedFramePipeline with band streams + fused SAO launches must leave exactly the planes of the same pipeline with
every launch on one stream and the SAO passes per plane.  Also FramePipeline(split=k) against the picture in one piece.
  python tools/band_soak.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0, cases, frames, moved, offs = time.time(), 0, 0, 0, 0
    while time.time() - t0 < budget:
        depth = int(rng.choice([8, 8, 10]))
        W, rows = 64 * int(rng.integers(2, 9)), int(rng.integers(2, 10))
        Hh = 64 * rows - int(rng.choice([0, 0, 8, 24]))
        band_rows, streams = int(rng.integers(1, 5)), int(rng.choice([2, 3]))
        packed = [True, "t", False][int(rng.integers(0, 3))] if depth == 8 else False
        R, subme, level = int(rng.choice([8, 12, 16, 24])), int(rng.integers(1, 4)), int(rng.integers(1, 3))
        qp = int(rng.integers(22, 38)) + 12 * (depth == 10)
        nfr = 3
        clip = F.synth_clip(W, Hh, nfr + 1, depth=depth, seed=int(rng.integers(1, 1 << 30)))
        pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
        kw = dict(rng=R, subme=subme, level=level, qp=qp, want_surf=bool(packed) or bool(rng.integers(0, 2)), packed=packed, deblock=True, sao=True, chroma=True,
                  sao_apply=True, sign_hide=bool(rng.integers(0, 2)), lookahead=(W, Hh))
        desc = f"{W}x{Hh} d{depth} bands {band_rows} streams {streams} packed {packed} R {R} subme {subme} level {level} qp {qp}"
        a = S.BandedFramePipeline(pics[0].w64, pics[0].h64, depth, dev, band_rows=band_rows, **kw)
        for pipe in a.pipes.values():
            pipe.fuse_sao = False
        b = S.BandedFramePipeline(pics[0].w64, pics[0].h64, depth, dev, band_rows=band_rows, streams=streams, **kw)
        # the picture in one piece: sequential launches against search -> reconstruction in parts on side streams
        kw2 = dict(kw, subpel_planes=bool(rng.integers(0, 2)), parallel_planes=True)
        c = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, **kw2)
        d = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, split=int(rng.integers(2, 5)), **kw2)
        refs = [pics[0].like([p.clone() for p in pics[0].planes()]) for _ in range(4)]
        for k in range(1, nfr + 1):
            for fp, ref in zip((a, b, c, d), refs):
                fp.run(pics[k], ref)
            torch.cuda.synchronize()
            for (x, y, what) in ((a, b, "band streams + fused SAO"), (c, d, "split")):
                for i, (p, q) in enumerate(zip(x.final_planes(), y.final_planes())):
                    if not torch.equal(p, q):
                        print(f"MISMATCH ({what}) frame {k} plane {i}: {desc}", flush=True)
                        sys.exit(1)
            moved += int(not torch.equal(a.final_planes()[0], pics[k].planes()[0])) and int(not torch.equal(a.final_planes()[0], refs[0].planes()[0]))
            offs += int((b.pipe_sets[0][b.bands[0][1]].sao.params.view(-1, 7)[:, 0] >= 0).sum())
            for fp, ref in zip((a, b, c, d), refs):
                for dst, src in zip(ref.planes(), fp.final_planes()):
                    dst.copy_(src)
            frames += 1
        cases += 1
    print(f"band soak: {cases} random configurations, {frames} frames each way, all planes equal ({time.time() - t0:.0f} s, seed {seed}); "
          f"{moved} of the {frames} filtered pictures differ from both their source and their reference, {offs} CTUs of first bands got SAO offsets")


if __name__ == "__main__":
    main()
