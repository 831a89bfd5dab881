#!/usr/bin/env python3
"""Latency of ONE frame cost estimate (x265hip_lowres_cost, P picture): one workgroup against the split form (measurement aid).
python tools/lowres_cost_split_probe.py [bands ...]     (X265HIP_LOWRES_COST_SPLIT values; "auto" = the library's own choice)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
dev = torch.device("cuda:0")
choices = sys.argv[1:] or ["1", "auto", "4", "8", "16"]
for W, H in ((1920, 1080), (3840, 2160)):
    clip = F.synth_clip(W, H, 2, depth=8, seed=5)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    lc, lr = S.Lookahead(W, H, 8, dev), S.Lookahead(W, H, 8, dev)
    lc.run(cur); lr.run(ref)
    st = [S.LookaheadCost(lc, dev)]
    for c in choices:
        if c == "auto":
            os.environ.pop("X265HIP_LOWRES_COST_SPLIT", None)
        else:
            os.environ["X265HIP_LOWRES_COST_SPLIT"] = c
        S.LookaheadCost.run_batch(st, [lc], [lr]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            S.LookaheadCost.run_batch(st, [lc], [lr])
        torch.cuda.synchronize()
        print(f"{W}x{H} ({lc.wcu} x {lc.hcu} blocks) bands={c:>4s}: {(time.perf_counter() - t0) / 5 * 1e3:7.3f} ms per estimate", flush=True)
