#!/usr/bin/env python3
"""[bench.py --search-probe: the primitive-level part of bench.py's cpu_baseline leg - the only place besides tests/ and smoke() that
loads oracle/, and only to time the CPU path beside the kernels]

Throughput probe of x265hip_me_search: every 8x8..64x64 PU of every CTU of a frame, predictor (0,0)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
dev = torch.device("cuda:0")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
clip = F.synth_clip(W, H, 2, depth=8, seed=5)
cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
jobs = []
for cy in range(0, cur.h64, 64):
    for cx in range(0, cur.w64, 64):
        for n in (8, 16, 32, 64):
            for y in range(0, 64, n):
                for x in range(0, 64, n):
                    jobs.append((cx + x, cy + y, n, n, 0, 0, 0, 0, 0))
jn = np.array(jobs, dtype=A.me_search_job_dtype())
jd = torch.from_numpy(jn.view(np.uint8).reshape(-1).copy()).to(dev)
cq, qoff = F.qpel_cost_table(57, qmax=8 * 64 + 300)
cq_d = torch.from_numpy(cq.view(np.int16)).to(dev)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_api as O            # CPU column only (the oracle restatement of motionEstimate, OpenMP over PUs)
import time
from bench import effective_cpus
threads = effective_cpus()
print(f"# x265hip_me_search: every 8x8..64x64 PU of every CTU, predictor (0,0), merange 57; CPU column = oracle/x265_oracle_search.c "
      f"(-march=x86-64-v3) on {threads} threads (container CPU quota) over a sample of the same jobs")
integral = A.sea_integral(8, ref.t, ref.stride, ref.org, ref.w64, ref.h64, F.MARGIN_X, F.MARGIN_Y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): A.sea_integral(8, ref.t, ref.stride, ref.org, ref.w64, ref.h64, F.MARGIN_X, F.MARGIN_Y, planes=integral[0])
e1.record(); torch.cuda.synchronize()
print(f"{W}x{H} x265hip_sea_integral (twelve block-sum planes of one reference): {e0.elapsed_time(e1) / 3:.3f} ms")
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for name, m in (("dia", A.ME_DIA), ("hex", A.ME_HEX), ("umh", A.ME_UMH), ("star", A.ME_STAR), ("sea", A.ME_SEA)):
    if only and name not in only:
        continue
    for subme in (2, 3):
        f = lambda: A.me_search(8, cur.t, cur.stride, cur.org, ref.t, ref.stride, ref.org, m, subme, 57, cq_d, qoff, (-57, -57), (57, 57), jd, len(jn),
                                integral=integral if m == A.ME_SEA else None)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): f()
        e1.record(); torch.cuda.synchronize()
        t_gpu = e0.elapsed_time(e1) / 3
        ns = min(len(jn), 85 * 64)                                   # 64 CTUs' worth of PUs on the CPU
        t0 = time.perf_counter()
        O.motion_estimate(8, cur.host, ref.host, cur.stride, cur.org, m, subme, 57, cq, qoff, (-57, -57), (57, 57), jn[:ns], nthreads=threads, avx2=O.host_has_avx2())
        t_cpu = (time.perf_counter() - t0) * len(jn) / ns
        print(f"{W}x{H} {name:4s} subme {subme}: {len(jn)} PUs, GPU {t_gpu:.3f} ms ({len(jn) / t_gpu / 1e3:.1f} M PU/s), "
              f"CPU {t_cpu * 1e3:.1f} ms extrapolated from {ns} PUs ({threads} threads) -> GPU/CPU {t_cpu * 1e3 / t_gpu:.0f}x")
