"""Where does the real x265 (C primitives) spend its worker time, per family of EncoderPrimitives slots?  Runs one encode of a
tools/encoder_bench.py configuration with oracle/ref_profile.cpp's cycle-counting thunks in the table and prints each family's share
of the process CPU time.  Test infrastructure (needs oracle/_ref); used to rank what to take off the CPU next."""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tools import encoder_bench as EB          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the picture (the 8-vCPU container cannot do 4K in reasonable time)")
    ap.add_argument("--frame-threads", type=int, default=1)
    ap.add_argument("--seams", action="store_true", help="profile the encode WITH the row-granular seams in place (libx265hip.so providers: needs the GPU): "
                    "the lookup stubs sit over the counting thunks, so what the families show is the host work that REMAINS")
    ap.add_argument("--seam-range", type=int, default=24)
    ap.add_argument("--seam-min-level", type=int, default=1)
    ap.add_argument("--seam-min-pu", type=int, default=16)
    ap.add_argument("--no-lookahead-seam", action="store_true")
    ap.add_argument("--seam-layout", default="records", choices=["records", "planes"])
    ap.add_argument("--seam-centre-range", type=int, default=0)
    ap.add_argument("--control", action="store_true", help="profile the HOST-ONLY control instead of the C table: sad_x3 / sad_x4 split into single SADs (x265ref_split_fill_table), no GPU")
    ap.add_argument("--seam-split-rest", action="store_true", help="with --seams: whatever the services do not answer takes the control's split SADs (bench.py's seam legs)")
    ap.add_argument("--seam-no-sad", action="store_true", help="with --seams: no SAD lookup stubs (sub-sample / lookahead / AQ / weightAnalyse services only)")
    ap.add_argument("--seam-aq", action="store_true")
    ap.add_argument("--seam-weight-analyse", action="store_true")
    ap.add_argument("--seam-cost", action="store_true", help="with --seams: the sub-sample cost tables (x265hip_cost_stream) behind subpelCompare, bench.py's configuration (1 candidate, the 85-position set)")
    ap.add_argument("--no-subpel-seam", action="store_true", help="with --seams: no phase-plane service (bench.py's cfg3 leg: cost tables alone)")
    ap.add_argument("--provider", default="gpu", choices=["gpu", "oracle"], help="oracle = the CPU checker providers (plumbing test of this tool without a GPU)")
    a = ap.parse_args()
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    cfg = EB.CONFIGS[a.config]
    w, h, depth = int(cfg["width"] * a.scale) // 16 * 16, int(cfg["height"] * a.scale) // 16 * 16, cfg["depth"]
    clip = F.synth_clip(w, h, a.frames, depth=depth, seed=265, fade=cfg.get("fade"))
    yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
    lib = EB.ref_lib(depth)
    cores = EB.effective_cpus()
    opts = [("pools", str(cores)), ("frame-threads", str(a.frame_threads)), ("crf", "28")] + cfg["opts"]
    filler = ctypes.cast(lib.x265ref_profile_fill_table, ctypes.c_void_p)
    note = closer = None
    if a.control and not a.seams:
        from tools import seam_driver as SD
        lib = SD.seam_lib(depth)
        lib.x265ref_seam_disable()
        filler = ctypes.cast(lib.x265ref_split_fill_table_profiled, ctypes.c_void_p)
    if a.seams:
        from tools import seam_driver as SD
        if not a.no_lookahead_seam:
            opts.append(("lookahead-slices", "1"))
        lib, _, note, closer, _ = SD.install(depth, w, h, provider=a.provider, rng=a.seam_range, slots=24 if depth == 8 else 40, min_pu=128 if a.seam_no_sad else a.seam_min_pu, verify=False,
                                             lookahead=None if a.no_lookahead_seam else a.provider, subpel=None if a.no_subpel_seam else a.provider, subpel_slots=12, streamed=True,
                                             cost=a.provider if a.seam_cost else None, cost_cfg=SD.cost_config(cfg["preset"], opts, set_subme=4) if a.seam_cost else None,
                                             min_level=a.seam_min_level, pictures=24, layout=1 if a.seam_layout == "planes" else 0, centre_range=a.seam_centre_range,
                                             lookahead_min_blocks=None, min_ctus=None, split_rest=a.seam_split_rest, aq=a.provider if a.seam_aq else None, aq_min_blocks=None,
                                             weight_analyse=a.provider if a.seam_weight_analyse else None, weight_min_blocks=None)
        filler = ctypes.cast(lib.x265ref_seam_fill_table_profiled, ctypes.c_void_p)
    lib.x265ref_profile_tsc.restype = ctypes.c_uint64
    t0, c0, tsc0 = time.perf_counter(), time.process_time(), lib.x265ref_profile_tsc()
    md5, nbytes, sec, filled = EB.encode(lib, yuv, w, h, a.frames, cfg["preset"], opts, filler)
    wall, cpu, tsc = time.perf_counter() - t0, time.process_time() - c0, lib.x265ref_profile_tsc() - tsc0
    hz = tsc / wall
    cyc, cnt, names = (ctypes.c_uint64 * 32)(), (ctypes.c_uint64 * 32)(), (ctypes.c_char_p * 32)()
    lib.x265ref_profile_report.argtypes = [ctypes.c_void_p] * 3
    n = lib.x265ref_profile_report(cyc, cnt, names)
    rows = sorted(((names[i].decode(), cyc[i] / hz, int(cnt[i])) for i in range(n)), key=lambda r: -r[1])
    inside = sum(r[1] for r in rows)
    print(f"# {a.config} {w}x{h} {depth}-bit preset {cfg['preset']} {dict(cfg['opts'])}, {a.frames} frames, {cores} pool threads: {a.frames / sec:.3f} fps, "
          f"process CPU {cpu:.2f} s, wall {wall:.2f} s, {filled} slots wrapped (C primitives, -O3, no asm)")
    print(f"# {'family':44s} {'CPU s':>8s} {'% of CPU':>9s} {'calls':>12s} {'ns/call':>9s}")
    for name, s, c in rows:
        if c:
            print(f"  {name:44s} {s:8.3f} {100 * s / cpu:8.1f}% {c:12d} {1e9 * s / c:9.0f}")
    print(f"  {'(all wrapped primitives)':44s} {inside:8.3f} {100 * inside / cpu:8.1f}%")
    print(f"  {'(encoder code outside the table)':44s} {cpu - inside:8.3f} {100 * (cpu - inside) / cpu:8.1f}%")
    if a.seams:
        st = (ctypes.c_uint64 * 12)()
        lib.x265ref_seam_profile_report.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
        lib.x265ref_seam_profile_report(st)
        print("# stages of binding/x265hip_x265_binding.cpp, whole (the primitives they call are ALSO in the families above; process CPU includes the providers' worker threads)")
        for i, name in enumerate(("MotionEstimate::motionEstimate (integer search on lookups + sub-sample refinement)", "MotionEstimate::subpelCompare (inside motionEstimate)",
                                  "CostEstimateGroup::estimateFrameCost (waits for x265hip_lowres_cost_host)", "row hand-over in FrameFilter::processPostRow (memcpy into pinned staging)",
                                  "SAD lookups of the integer search (served or passed on; inside motionEstimate)", "context set-up of the motionEstimate wrapper (pair / view look-up)")):
            s_, c_ = st[2 * i] / hz, int(st[2 * i + 1])
            if c_:
                print(f"  {name:100s} {s_:8.3f} s {100 * s_ / cpu:6.1f}% of CPU {c_:10d} calls {1e9 * s_ / c_:9.0f} ns/call")
        print("# seam report: " + json.dumps(note()))
        closer()


if __name__ == "__main__":
    main()
