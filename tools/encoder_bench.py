#!/usr/bin/env python3
"""Encoder-level measurement, SURVEY.md section 8(d)(iii) / tier T3: the REAL reference encoder (oracle/_ref/libx265ref<depth>.so =
x265 3.5 compiled from /root/reference by oracle/Makefile, C primitives, no asm - nasm is not in the image) on the synthetic clip of
each BASELINE.json configuration, once per primitive-table flavour:

    c      the reference's own C table (setupCPrimitives + aliases)            -> the host-CPU baseline, kind "reference"
    csse   (v3 build only) c + the reference's SSE intrinsic DCT / iDCT / dequant_scaling (common/vec/*.cpp) -> the strongest host table this image can build
    hip    1816 slots served by libx265hip.so's per-call stubs (table layer)   -> drop-in, byte-identical, launch-bound
    seam   C table + the stage-level seam (binding/x265hip_x265_binding.cpp): MotionEstimate::motionEstimate replays its integer search on the
           SAD surfaces one x265hip_me_fullsearch launch per (picture, reference) produced (batch layer)

fps = frames / seconds of the encode loop (what encoder.cpp:2708 prints), bitstreams compared by md5 (--no-info, CRF, fixed
frame threads: SURVEY section 4 determinism rules).  Used by `bench.py --encoder`; test infrastructure, not product code.

  python tools/encoder_bench.py [--configs cfg1,cfg2,cfg3] [--frames N] [--tables c,hip,seam] [--budget-s S]
"""
import argparse
import ctypes
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# BASELINE.json configs -> (width, height, depth, preset, extra options, default frame count for the C table)
CONFIGS = {
    "cfg1": dict(width=1280, height=720, depth=8, preset="ultrafast", opts=[("keyint", "1")], frames=8,
                 name="configs[0]: 720p 8-bit --preset ultrafast --keyint 1"),
    "cfg2": dict(width=1920, height=1080, depth=8, preset="medium", opts=[], frames=6,
                 name="configs[1]: 1080p 8-bit --preset medium"),
    "cfg3": dict(width=3840, height=2160, depth=8, preset="slow", opts=[("me", "star")], frames=4,
                 name="configs[2]: 2160p 8-bit --preset slow --me star"),
    # cfg3 on a luma FADE: the content x265's default --weightp / --weightb exist for - every P slice searches weighted reference planes
    # (MotionReference::applyWeight, reference.cpp:119-178); round-3 verdict, next 1
    "cfg3f": dict(width=3840, height=2160, depth=8, preset="slow", opts=[("me", "star")], frames=4, fade=(1.0, 0.3),
                  name="configs[2] on a luma fade (gain 1.0 -> 0.3): 2160p 8-bit --preset slow --me star, weighted references"),
    "cfg4": dict(width=3840, height=2160, depth=10, preset="slower", opts=[], frames=3,
                 name="configs[3]: 2160p 10-bit --preset slower (one GPU's share of the frame-parallel job)"),
    "cfg5": dict(width=7680, height=4320, depth=10, preset="veryslow", opts=[("ctu", "64"), ("rd", "6")], frames=3,
                 name="configs[4]: 4320p 10-bit --preset veryslow --ctu 64 --rd 6 (one GPU's share of the frame-parallel job)"),
}
FILL = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)


def ref_lib(depth, build=""):
    """build: "" = g++ -O3 (x86-64 baseline, what every parity test uses); "v3" = the same sources with -march=x86-64-v3 (AVX2 auto-vectorised
    C primitives, 8-bit only: oracle/Makefile `refv3`)"""
    path = os.path.join(ROOT, "oracle", "_ref", f"libx265ref{depth}{build}.so")
    if not os.path.exists(path):
        raise SystemExit(f"{path} missing: the real-reference build (make -C oracle ref) only exists where /root/reference does; "
                         "the built .so travels to the GPU box with the snapshot")
    lib = ctypes.CDLL(path)
    lib.x265ref_encode.restype = ctypes.c_long
    lib.x265ref_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    return lib


def encode(lib, yuv, w, h, nframes, preset, opts, filler=None):
    out = np.zeros(64 << 20, np.uint8)
    arr = (ctypes.c_char_p * (2 * len(opts)))()
    for i, (k, v) in enumerate(opts):
        arr[2 * i] = k.encode()
        arr[2 * i + 1] = v.encode() if v is not None else None
    sec, filled = ctypes.c_double(), ctypes.c_int()
    n = lib.x265ref_encode(yuv.ctypes.data, w, h, nframes, preset.encode(), arr, len(opts), filler,
                           out.ctypes.data, out.size, ctypes.byref(sec), ctypes.byref(filled))
    if n <= 0:
        raise RuntimeError(f"reference encode failed ({n})")
    return hashlib.md5(out[:n].tobytes()).hexdigest(), int(n), sec.value, filled.value


def effective_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def run_config(key, tables, frames=None, frame_threads=1, budget_s=240.0, log=sys.stderr, seam=None, build=""):
    seam = seam or {"range": 32, "slots": 8, "min_pu": 8, "verify": False, "lookahead": False}
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    cfg = CONFIGS[key]
    w, h, depth = cfg["width"], cfg["height"], cfg["depth"]
    n = frames or cfg["frames"]
    cores = effective_cpus()
    clip = F.synth_clip(w, h, n, depth=depth, seed=265, fade=cfg.get("fade"))
    yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
    lib = ref_lib(depth, build)
    opts = [("pools", str(cores)), ("frame-threads", str(frame_threads)), ("crf", "28")] + cfg["opts"]
    # diagnostic only (what bounds the encode?): ENCODER_BENCH_EXTRA_OPTS="b-adapt=0,rc-lookahead=10" appends x265 options to EVERY leg of the run
    opts += [tuple(kv.split("=", 1)) if "=" in kv else (kv, None) for kv in os.environ.get("ENCODER_BENCH_EXTRA_OPTS", "").split(",") if kv]
    if seam.get("lookahead"):        # the lookahead seam serves unsliced frame cost estimates: every leg of this run walks the lowres picture in one piece
        opts.append(("lookahead-slices", "1"))
    res = {"config": cfg["name"], "size": f"{w}x{h}", "depth": depth, "preset": cfg["preset"], "options": dict(opts), "pool_threads": cores,
           "reference_build": "x265 3.5 C primitives (no asm: nasm is not in the image), g++ -O3" +
                              (" -march=x86-64-v3 -ffp-contract=off (AVX2 auto-vectorised; the hand-written NASM AVX2 path cannot be assembled here)" if build == "v3" else "")}
    md5_c = {}
    for t in tables:
        nf = n
        filler, note, closer, enc_lib = None, None, None, lib
        if t == "csplit":          # host-only control: the C table with sad_x3 / sad_x4 answered by N calls of its own sad (what the seam stubs do on a miss)
            from tools import seam_driver as SD
            enc_lib = SD.seam_lib(depth, build)
            enc_lib.x265ref_seam_disable()
            filler = ctypes.cast(enc_lib.x265ref_split_fill_table, ctypes.c_void_p)
        if t == "csse":            # the C table + the reference's own SSE3 / SSSE3 / SSE4.1 intrinsic transforms (common/vec/*.cpp): only in the v3 flavour (oracle/Makefile)
            if not hasattr(lib, "x265ref_sse_fill_table") or build != "v3":
                continue
            filler = ctypes.cast(lib.x265ref_sse_fill_table, ctypes.c_void_p)
        if t == "hip":
            A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
            L = A.lib()
            A.set_entropy_bits(list((ctypes.c_uint32 * 128).in_dll(lib, "x265_entropyStateBits")))      # the host encoder's own CABAC bit costs
            filler = ctypes.cast(L.x265hip_setup_primitives, ctypes.c_void_p)
            calls0 = L.x265hip_table_calls()
            c = res.get("c")
            if c and budget_s:         # per-call stubs cost ~20 us per primitive call: shorten the clip so the leg fits the budget
                est_calls = c.get("est_primitive_calls_per_frame", 0)
                # measured on the first frames below; start from 2 frames, never less
                nf = max(2, min(n, int(budget_s / max(1e-9, est_calls * 20e-6 / max(1, min(cores, 8)))))) if est_calls else min(n, 2)
        elif t == "seam":
            if cfg["preset"] == "ultrafast":       # --ctu 32 and, in cfg1, no inter pictures at all: nothing for the seam to serve
                continue
            from tools import seam_driver as SD
            enc_lib, filler, note, closer, _ = SD.install(depth, w, h, provider="gpu", rng=seam["range"], slots=seam["slots"],
                                                          min_pu=128 if seam.get("no_sad") else seam["min_pu"],
                                                          verify=seam["verify"], lookahead="gpu" if seam.get("lookahead") else None,
                                                          subpel="gpu" if seam.get("subpel") else None, subpel_slots=seam.get("subpel_slots", 6),
                                                          streamed=bool(seam.get("streamed")), min_level=seam.get("min_level", 0),
                                                          pictures=seam.get("pictures", 24), band_rows=seam.get("band_rows", 0),
                                                          weighted=seam.get("weighted", True), layout=seam.get("layout", 0), centre_range=seam.get("centre_range", 0),
                                                          lookahead_min_blocks=seam.get("lookahead_min_blocks"),      # None: the binding's own size gates
                                                          min_ctus=seam.get("min_ctus"), build=build, aq="gpu" if seam.get("aq") else None,
                                                          aq_min_blocks=seam.get("aq_min_blocks"), weight_analyse="gpu" if seam.get("weight_analyse") else None,
                                                          weight_min_blocks=seam.get("weight_min_blocks"), split_rest=bool(seam.get("split_rest")),
                                                          cost="gpu" if seam.get("cost") else None,
                                                          cost_cfg=SD.cost_config(cfg["preset"], opts, centre_range=seam.get("cost_centre_range", 57), window=seam.get("cost_window", 8),
                                                                                  candidates=seam.get("cost_candidates", 1), slots=seam.get("cost_slots", 24),
                                                                                  pictures=seam.get("cost_pictures", 40), views=seam.get("cost_views", 12), set_subme=seam.get("cost_set_subme"), sad_costs=seam.get("cost_sad")) if seam.get("cost") else None)
        t0, c0 = time.perf_counter(), time.process_time()
        md5, nbytes, sec, filled = encode(enc_lib, yuv[: nf * (yuv.size // n)], w, h, nf, cfg["preset"], opts, filler)
        wall, cpu = time.perf_counter() - t0, time.process_time() - c0
        r = {"frames": nf, "seconds": round(sec, 3), "fps": round(nf / sec, 4), "bytes": nbytes, "md5": md5, "slots_replaced": filled,
             "wall_seconds_with_open_close": round(wall, 3), "process_cpu_seconds": round(cpu, 2)}
        if t == "c":
            md5_c[nf] = md5
            # SURVEY section 6 call-rate probe (counting thunks, 1080p): medium ~3.0 M, slow ~3.4 M primitive calls per frame; scale by area
            per_1080p = {"ultrafast": 1.2e6, "medium": 3.0e6, "slow": 3.4e6, "slower": 6e6}.get(cfg["preset"], 3e6)
            r["est_primitive_calls_per_frame"] = int(per_1080p * (w * h) / (1920 * 1080))
        else:
            if nf not in md5_c and not os.environ.get("ENCODER_BENCH_NO_MD5"):      # the C table on the same shortened clip, for the md5 comparison
                md5_c[nf] = encode(lib, yuv[: nf * (yuv.size // n)], w, h, nf, cfg["preset"], opts, None)[0]
            r["md5_equal_to_c_table"] = (md5 == md5_c[nf]) if nf in md5_c else None      # ENCODER_BENCH_NO_MD5 (A/B matrices: the bench legs hold the md5 check)
            if t == "hip":
                r["primitive_calls_through_gpu"] = int(L.x265hip_table_calls() - calls0)
                r["us_per_primitive_call"] = round(1e6 * sec / max(1, r["primitive_calls_through_gpu"]), 2)
            if note:
                r["seam"] = note()
            if closer:
                closer()
        res[t] = r
        print(f"[encoder] {key} {t}: {json.dumps(r)}", file=log, flush=True)
    return res


def main():
    if os.environ.get("X265HIP_BLOCKING_SYNC"):        # A/B: do the providers' waits burn host cores?  (hipDeviceScheduleBlockingSync = 4, before the first HIP call)
        print("hipSetDeviceFlags(blocking sync) ->", ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(4), file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg1,cfg2,cfg3")
    ap.add_argument("--tables", default="c,hip")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--frame-threads", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=240.0, help="target seconds for one per-call-stub (hip) leg; the clip is shortened to fit")
    ap.add_argument("--seam-range", type=int, default=32, help="displacements the SAD surfaces cover (lookups outside fall back to the C primitive)")
    ap.add_argument("--seam-slots", type=int, default=8, help="(picture, reference) pairs resident in pinned host memory")
    ap.add_argument("--seam-min-pu", type=int, default=8, help="serve partitions whose smaller side is at least this")
    ap.add_argument("--seam-verify", action="store_true", help="check every lookup against the C primitive in flight (slow)")
    ap.add_argument("--seam-lookahead", action="store_true",
                    help="also serve CostEstimateGroup::estimateFrameCost's block loop from x265hip_lowres_cost_host (adds --lookahead-slices 1 to every leg)")
    ap.add_argument("--seam-aq", action="store_true",
                    help="also serve LookaheadTLD::calcAdaptiveQuantFrame (every quantisation group's AC energy + the QP offsets) from x265hip_aq_frame_host")
    ap.add_argument("--seam-weight-analyse", action="store_true",
                    help="also serve the frame encoder's weightAnalyse (compensated planes, weightCost of every scale / offset pair) from x265hip_weight_analyse_host")
    ap.add_argument("--seam-split-rest", action="store_true",
                    help="whatever the services do not answer uses the host-only control's split sad_x3 / sad_x4 (compare with --tables csplit: the difference is the services alone)")
    ap.add_argument("--seam-subpel", action="store_true",
                    help="also serve MotionEstimate::subpelCompare from x265hip_phase_cache (every fractional phase of a reference picture interpolated once)")
    ap.add_argument("--seam-streamed", action="store_true",
                    help="row-granular providers (x265hip_me_stream / x265hip_phase_stream fed by the FrameFilter::processPostRow hook): serve under --frame-threads > 1")
    ap.add_argument("--seam-min-level", type=int, default=1, help="row-granular SAD provider: 1 = download only the 16x16-and-up tail of every record")
    ap.add_argument("--seam-pictures", type=int, default=24, help="row-granular SAD provider: pictures resident on the device")
    ap.add_argument("--seam-band-rows", type=int, default=0, help="row-granular SAD provider: most CTU rows per search launch (0 = 8)")
    ap.add_argument("--seam-no-sad", action="store_true", help="install no SAD lookup stubs (sub-sample / lookahead seams only)")
    ap.add_argument("--seam-layout", default="records", choices=["records", "planes"], help="row-granular SAD provider: what lands in host memory - records "
                    "(all PUs of a displacement together) or PU-major planes (X265HIP_STREAM_PLANES)")
    ap.add_argument("--seam-centre-range", type=int, default=0, help="row-granular SAD provider: centre every CTU's window on its own displacement, found within +-this (0 = off)")
    ap.add_argument("--ref-build", default="", choices=["", "v3"], help="reference build flavour of every leg: '' = g++ -O3, v3 = + -march=x86-64-v3 (AVX2 auto-vectorised C)")
    ap.add_argument("--seam-no-weighted", action="store_true", help="weighted references pass to the host (the round-3 behaviour), for A/B on a fade")
    ap.add_argument("--plan", default="", help="bench.py's child-process mode: a JSON list of legs {name, key, tables, frames, frame_threads, seam, build}; the other options are ignored")
    ap.add_argument("--deadline-s", type=float, default=0.0, help="--plan: legs are not started once this many seconds have passed (0 = no deadline)")
    ap.add_argument("--out", default="", help="--plan: write {\"encoder\": {name: result}} to this file after EVERY leg (what is there survives a crash or a timeout of a later leg)")
    ap.add_argument("--seam-cost", action="store_true",
                    help="also answer MotionEstimate::subpelCompare's SATD comparisons from x265hip_cost_stream's records (sub-sample costs around each PU's best integer vectors)")
    ap.add_argument("--seam-cost-candidates", type=int, default=1, help="cost tables: integer vectors per PU (1 or 2)")
    ap.add_argument("--seam-cost-set-subme", type=int, default=0, help="cost tables: hold the position set of this --subme row when it is larger than the encode's own (4: 85 positions)")
    ap.add_argument("--seam-cost-sad", action="store_true", help="cost tables: the records also carry the costs of the SAD-typed comparisons (the search's predictor candidates)")
    ap.add_argument("--seam-lookahead-min-blocks", type=int, default=-1, help="size gate of the lookahead seam in 8x8 lowres blocks (-1 = the binding's own 16384: serve from 4K up; 0 = any size)")
    ap.add_argument("--seam-min-ctus", type=int, default=-1, help="size gate of the search seams in CTUs (-1 = the binding's own 1000: serve from 4K up; 0 = serve any size)")
    ap.add_argument("--seam-cost-window", type=int, default=8, help="cost tables: the candidates are the smallest SADs within +-this of each CTU's own displacement")
    ap.add_argument("--seam-cost-slots", type=int, default=24, help="cost tables: (picture, reference) pairs resident in pinned host memory (37 MB each at 4K preset slow)")
    ap.add_argument("--seam-cost-views", type=int, default=12, help="cost tables: reference views (phase planes, 450 MB each at 4K 8-bit) resident on the device")
    ap.add_argument("--seam-subpel-slots", type=int, default=6, help="reference pictures whose phase planes stay in pinned host memory (450 MB each at 4K 8-bit)")
    args = ap.parse_args()
    if args.plan:
        out = {}
        t_start = time.perf_counter()
        for leg in json.loads(args.plan):
            # every leg has a time budget of its own kind: a leg is not STARTED once the plan's deadline has passed (round-5 advisor: all legs in one child with one
            # limit - a slow box lost every result that came after the limit, and the caller's line with them); what has been measured is already in --out
            if args.deadline_s and time.perf_counter() - t_start > args.deadline_s:
                out[leg["name"]] = {"skipped": f"not started: the plan's deadline of {args.deadline_s:.0f} s had passed"}
                if args.out:
                    with open(args.out + ".tmp", "w") as f:
                        json.dump({"encoder": out}, f)
                    os.replace(args.out + ".tmp", args.out)
                continue
            try:
                out[leg["name"]] = run_config(leg["key"], leg["tables"], leg.get("frames") or None, leg.get("frame_threads", 1), args.budget_s, seam=leg.get("seam"),
                                              build=leg.get("build", ""))
            except BaseException as e:       # noqa: BLE001 - incl. SystemExit from a missing oracle/_ref build: the other legs stand
                out[leg["name"]] = {"error": repr(e)}
            if args.out:
                with open(args.out + ".tmp", "w") as f:
                    json.dump({"encoder": out}, f)
                os.replace(args.out + ".tmp", args.out)
        if not args.out:
            print(json.dumps({"encoder": out}))
        return
    seam = {"range": args.seam_range, "slots": args.seam_slots, "min_pu": args.seam_min_pu, "verify": args.seam_verify, "lookahead": args.seam_lookahead,
            "subpel": args.seam_subpel, "subpel_slots": args.seam_subpel_slots, "streamed": args.seam_streamed, "min_level": args.seam_min_level,
            "pictures": args.seam_pictures, "band_rows": args.seam_band_rows, "no_sad": args.seam_no_sad, "weighted": not args.seam_no_weighted,
            "layout": 1 if args.seam_layout == "planes" else 0, "centre_range": args.seam_centre_range, "aq": args.seam_aq, "weight_analyse": args.seam_weight_analyse, "split_rest": args.seam_split_rest,
            "cost": args.seam_cost, "cost_candidates": args.seam_cost_candidates, "cost_window": args.seam_cost_window, "cost_slots": args.seam_cost_slots, "cost_views": args.seam_cost_views,
            "cost_centre_range": args.seam_centre_range or 57, "cost_set_subme": args.seam_cost_set_subme or None, "cost_sad": args.seam_cost_sad, "min_ctus": None if args.seam_min_ctus < 0 else args.seam_min_ctus,
            "lookahead_min_blocks": None if args.seam_lookahead_min_blocks < 0 else args.seam_lookahead_min_blocks}
    out = {k: run_config(k, args.tables.split(","), args.frames or None, args.frame_threads, args.budget_s, seam=seam, build=args.ref_build) for k in args.configs.split(",")}
    print(json.dumps({"encoder": out}))


if __name__ == "__main__":
    main()
