import importlib, sys, time
sys.path.insert(0, '.')
import torch
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
dev = torch.device("cuda:0")
for depth in (8, 10):
    clip = F.synth_clip(1920, 1080, 2, depth=depth, seed=5)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    for want_surf in (True, False):
        ms = P.MotionSearch(cur.w64, cur.h64, 57, depth, dev, want_surf=want_surf)
        ms.run(cur, ref); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ms.run(cur, ref)
        e1.record(); torch.cuda.synchronize()
        print(depth, "surf+best" if want_surf else "best", e0.elapsed_time(e1) / 5, "ms")
