#!/usr/bin/env python3
"""What does the FIRST call into each kernel family cost in a fresh process (code-object loading is per .hip file and lazy)?  First against second call, ms."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
t00 = time.perf_counter()
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
print(f"torch device start-up {1e3 * (time.perf_counter() - t00):8.1f} ms")
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
dev = torch.device("cuda:0")
W, H = 1920, 1088
clip = F.synth_clip(W, H, 2, depth=8, seed=3, chroma=True) if "chroma" in F.synth_clip.__code__.co_varnames else F.synth_clip(W, H, 2, depth=8, seed=3)
pics = [P.DevicePicture(c[0], dev, *(c[1:3] if len(c) >= 3 else ())) for c in clip]
torch.cuda.synchronize()


def timed(name, fn):
    out = []
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:34s} first {out[0]:8.2f} ms   second {out[1]:8.2f} ms", flush=True)


la = [S.Lookahead(W, H, 8, dev) for _ in range(2)]
timed("lookahead_kernels (init + intra)", lambda: [l.run(p) for l, p in zip(la, pics)])
lc = S.LookaheadCost(la[1], dev)
timed("lowres_cost_kernels", lambda: lc.run(la[1], la[0]))
ms = P.MotionSearch(pics[0].w64, pics[0].h64, 57, 8, dev, want_surf=False, want_best=True)
timed("me_kernels (minima)", lambda: (ms.reset(), ms.search(pics[1], pics[0])))
sp = P.SubpelRefine(ms, 3, dev, phase_planes=True)
timed("phase_kernels + subpel_kernels", lambda: sp.run(pics[1], pics[0]))
pipe = S.FramePipeline(pics[0].w64, pics[0].h64, 8, dev, rng=57, subme=3, level=2, qp=27, want_surf=False, lookahead=(W, H), deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True, subpel_planes=True)
timed("whole step (tu, deblock, sao, border)", lambda: pipe.run(pics[1], pics[0]))
