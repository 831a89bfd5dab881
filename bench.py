#!/usr/bin/env python3
"""bench.py - frame-pipeline throughput of the MI355X block-primitive path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one 1080p frame (BASELINE.json configs[1]: 1920x1080 8-bit, allocated as 1920x1088 whole
CTUs like the reference) through the batched stages of x265-yuuki-asuna_amd/pipeline.py, all inputs
resident in HBM.  With N GPUs the job is frame-parallel (one frame per GPU per step, weak scaling);
the only data-path exchange is the one-to-many broadcast of the newest reference picture (RCCL), the
seam where the reference raises m_reconRowFlag (framefilter.cpp:664).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     - dominant kernel (CTU motion search): algorithmic bytes per launch
                 / HIP-event-measured launch time, vs 8 TB/s HBM
  cpu_baseline - the oracle's restatement of the same stage on the host cores (bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(F, P, clip, rng_r, target_s=15.0):
    """Time the oracle (CPU restatement, AVX2 build when the host has AVX2) on a bounded number of
    CTUs of the same workload, all host cores via OpenMP."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O          # cpu_baseline leg only
    avx2 = O.host_has_avx2()
    cur_buf, stride, org, w64, h64 = F.pad_plane(clip[1][0])
    ref_buf = F.pad_plane(clip[0][0])[0]
    cost = F.mv_cost_table(rng_r)
    nctu = (w64 // 64) * (h64 // 64)
    cores = os.cpu_count() or 1

    def run(n):
        t = time.perf_counter()
        O.me_fullsearch(8, cur_buf, stride, org, ref_buf, stride, org, w64, h64, rng_r, 0, n,
                        cost, cost, want_surf=False, want_best=True, nthreads=cores, avx2=avx2)
        return time.perf_counter() - t

    probe = min(nctu, max(cores, 8))
    t_probe = run(probe)
    n = int(min(nctu, max(probe, probe * target_s / max(t_probe, 1e-3))))
    t = run(n)
    fps = (n / nctu) / t
    return {"value": round(fps, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} of {nctu} CTUs of one 1080p frame, exhaustive +-{rng_r} search, best-mv only, "
                      f"oracle C ({'-march=x86-64-v3' if avx2 else 'generic x86-64'}) with OpenMP over CTUs, {t:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--range", type=int, default=57)          # reference default merange (param.cpp:198)
    ap.add_argument("--mode", default="surface", choices=["surface+best", "surface", "best"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")

    nclip = 4
    clip = F.synth_clip(args.width, args.height, nclip, depth=8, seed=265 + rank)
    pics = [P.DevicePicture(y, dev) for (y, _, _) in clip]
    ms = P.MotionSearch(pics[0].w64, pics[0].h64, args.range, 8, dev,
                        want_surf="surface" in args.mode, want_best="best" in args.mode)
    ref = pics[0]
    ref_plane = ref.t.clone()                        # the reference picture every rank searches in
    ref_pic = P.DevicePicture.__new__(P.DevicePicture)
    ref_pic.__dict__.update(ref.__dict__)
    ref_pic.t = ref_plane

    fp = P.FrameParallel(rank, world)
    ev = []

    def step(i, timed):
        cur = pics[1 + i % (nclip - 1)]
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ms.run(cur, ref_pic)
        if timed:
            e1.record()
            ev.append((e0, e1))
        # frame-parallel hand-off: the last rank's newest picture becomes everyone's next reference
        fp.exchange(ref_plane, cur.t)

    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
    alg_bytes = ms.algorithmic_bytes(bpp=1)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    csum = ms.checksum()

    if rank == 0:
        fps = world * args.steps / dt
        out = {
            "metric": "encoded fps + bit-exact check, 4K preset=slow, 1/2/4/8 MI355X vs host AVX2",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.width}x{args.height} 8-bit (BASELINE configs[1] picture size), frame pipeline stages: "
                                   f"ME exhaustive +-{args.range} for every 8x8/16x16/32x32/64x64 PU ({args.mode})",
                       "frames_per_step_per_gpu": 1, "parallelism": f"frame-parallel x{world}",
                       "ctus_per_frame": ms.nctu, "checksum": csum},
            "roofline": {"bound": "hbm", "kernel": "me_ctu_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": round(kern_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F, P, clip, args.range)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
