#!/usr/bin/env python3
"""Latency of ONE frame cost estimate (x265hip_lowres_cost) alone on the device: P picture (list 0 searched) and B picture (both lists searched), one workgroup
against the split form, and - B pictures, round 6 - the two lists walked side by side + the flat launch against both lists in one walk (measurement aid).
python tools/lowres_cost_split_probe.py [bands ...]     (X265HIP_LOWRES_COST_SPLIT values; "auto" = the library's own choice)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
dev = torch.device("cuda:0")
choices = sys.argv[1:] or ["1", "auto", "8", "16"]
refresh = A.lib().x265hip_lowres_cost_env_refresh


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for W, H in ((1920, 1080), (3840, 2160)):
    clip = F.synth_clip(W, H, 3, depth=8, seed=5)
    pics = [P.DevicePicture(c[0], dev) for c in clip]
    las = [S.Lookahead(W, H, 8, dev) for _ in range(3)]
    for la, pic in zip(las, pics):
        la.run(pic)
    l0, lc, l1 = las
    stP, stB = S.LookaheadCost(lc, dev), S.LookaheadCost(lc, dev, bidir=True)
    for c in choices:
        if c == "auto":
            os.environ.pop("X265HIP_LOWRES_COST_SPLIT", None)
        else:
            os.environ["X265HIP_LOWRES_COST_SPLIT"] = c
        os.environ.pop("X265HIP_LOWRES_COST_SO_OFF", None)
        refresh()
        tp = timed(lambda: stP.run(lc, l0))
        tb = timed(lambda: stB.run(lc, l0, l1))
        os.environ["X265HIP_LOWRES_COST_SO_OFF"] = "1"
        refresh()
        tb1 = timed(lambda: stB.run(lc, l0, l1))
        tf = timed(lambda: stB.run(lc, l0, l1, do_search=(0, 0)))
        print(f"{W}x{H} ({lc.wcu} x {lc.hcu} blocks) bands={c:>4s}: P {tp:7.3f} ms   B, both lists searched: side by side {tb:7.3f} ms, one walk {tb1:7.3f} ms   B, nothing searched (flat) {tf:6.3f} ms", flush=True)
os.environ.pop("X265HIP_LOWRES_COST_SO_OFF", None)
os.environ.pop("X265HIP_LOWRES_COST_SPLIT", None)
refresh()
