#!/usr/bin/env python3
"""bench.py - closed-loop frame-pipeline throughput of the MI355X block-primitive path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one 3840x2160 8-bit frame (BASELINE.json `metric`: "4K preset=slow" = configs[2]; allocated as
3840x2176 whole CTUs like the reference; --width/--height select the other picture sizes) through the batched stages of
the frame pipeline, everything resident in HBM:

    LA   lookahead preparation of the source picture: four half-resolution planes + border extension, intra cost
         estimate of every 8x8 lowres block (Lowres::init / lowresIntraEstimate); the P-frame cost estimate against the
         previous picture (estimateFrameCost) runs ahead on a side stream, --lookahead-batch pictures per launch
    ME   exhaustive +-57 search (the reference's default merange), all 85 PUs of every CTU: SAD surfaces
         (sad_x4 grouping) + best mv, one launch
    SUB  sub-pel refinement of every PU (subme 3 = preset slow)
    REC  32x32 prediction + residual DCT / quant / dequant / iDCT / reconstruction + SSE (MC + TU round trip)
    DBK  in-loop luma deblocking of the reconstruction (boundary strengths from the mvs / coded flags, edge filters)
    SAO  sample-adaptive-offset statistics of the deblocked reconstruction for every CTU, type and class (calcSaoStatsCTU;
         the parameter decision and therefore the offsets themselves are host work)
    EXT  border extension; the filtered reconstruction is the next frame's reference (closed loop)

With N GPUs the job is frame-parallel (one frame per GPU per step, weak scaling); the only data-path exchange is
the one-to-many broadcast of the newest reconstructed reference (RCCL), the seam where the reference raises
m_reconRowFlag (framefilter.cpp:664).  Every stage is parity-tested bit-exact against the oracle (tests/ -m gpu).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     - dominant kernel (ME surface kernel): algorithmic bytes per launch /
                 HIP-event launch time vs 8 TB/s; `traffic` = HBM bytes from the committed rocprofv3 PMC passes
  stages_ms    - HIP-event time of each stage (a separate, untimed pass)
  cpu_baseline - the oracle's restatement of the same pipeline on the host cores (bounded CTU sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

# The encoder-level leg calls x265hip_lowres_cost_host from several lookahead threads at once, each on its own stream; the HIP runtime
# maps streams onto 4 hardware queues by default, which serialises those latency-bound launches (5.35 instead of 5.8 fps at 4K).  Read
# when the runtime initialises, so it is set before torch is imported; a host encoder sets it in its environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG_DEPTH = {"cfg1": 8, "cfg2": 8, "cfg3": 8, "cfg4": 10, "cfg5": 10}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def oracle_chain(F, clip, rng_r, subme, level, qp, depth, n, cores, avx2, ref_planes=None, cur_index=1, tu_flags=2, band=None, sao_rdo=None):
    """The oracle's restatement (CPU; checker + cpu_baseline leg only) of one frame of the pipeline - clip[cur_index] searched in
    clip[cur_index - 1] (or in ref_planes = padded Y, Cb, Cr of a reconstruction) - on the first n CTUs.  n == all CTUs also runs the per-picture stages (lookahead, deblocking, SAO statistics)
    and returns every stage output for the bit-exact comparison with the device pipeline.  Returns (seconds, outputs)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O          # cpu_baseline / bit-exact checker leg only
    cur, stride, org, w64, h64 = F.pad_plane(clip[cur_index][0])
    row0, first_band, last_band, h64_full = 0, True, True, h64
    if band is not None:          # (first CTU row, CTU rows): the band is a slice of its own (stages.BandedFramePipeline); n counts the band's CTUs
        row0, nrows = band
        first_band, last_band = row0 == 0, (row0 + nrows) * 64 == h64
        org += row0 * 64 * stride
        h64 = nrows * 64
    src_prev = F.pad_plane(clip[cur_index - 1][0])[0]              # the lookahead scores SOURCE pictures
    # the picture searched / predicted from: the previous source frame, or (closed loop) the padded Y / Cb / Cr planes handed in
    ref = src_prev if ref_planes is None else ref_planes[0]
    cost = F.mv_cost_table(rng_r)
    cq, qoff = F.qpel_cost_table(rng_r)
    nctu = (w64 // 64) * (h64 // 64)
    lw, lh = ((clip[0][0].shape[1] // 2 + 7) >> 3) * 8, ((clip[0][0].shape[0] // 2 + 7) >> 3) * 8
    lstride = (lw + 2 * F.MARGIN_X + 31) & ~31
    out = {}
    t = time.perf_counter()
    if n == nctu and band is None:       # the lookahead stage is per picture: include it with whole-frame samples
        lp = O.lowres_init(depth, cur, stride, org, lstride, lstride * F.MARGIN_Y + F.MARGIN_X, lh + 2 * F.MARGIN_Y, lw, lh,
                           F.MARGIN_X, F.MARGIN_Y, avx2=avx2)
        ic, im, lc = O.lowres_intra(depth, lp[0], lstride, lstride * F.MARGIN_Y + F.MARGIN_X, lw // 8, lh // 8, 5, nthreads=cores, avx2=avx2)
        # the frame cost estimate against the previous picture is serial per picture (the reference spreads pictures over threads)
        lq, lqoff = F.qpel_cost_table(16, lam=1.0 if depth == 8 else 16.0, qmax=4 * (max(lw, lh) + 64))
        lpr = O.lowres_init(depth, src_prev, stride, org, lstride, lstride * F.MARGIN_Y + F.MARGIN_X, lh + 2 * F.MARGIN_Y, lw, lh,
                            F.MARGIN_X, F.MARGIN_Y, avx2=avx2)
        O.lowres_cost(depth, lp[0], lpr, lstride, lstride * F.MARGIN_Y + F.MARGIN_X, lw // 8, lh // 8, lq, lqoff, ic, avx2=avx2)
        out.update({"lowres_plane%d" % i: lp[i] for i in range(4)})
        out.update({"intra_cost": ic, "intra_mode": im, "lowres_costs": lc})
    _, best = O.me_fullsearch(depth, cur, stride, org, ref, stride, org, w64, h64, rng_r, 0, n, cost, cost,
                              want_surf=False, want_best=True, nthreads=cores, avx2=avx2)
    mv = O.subpel_refine(depth, cur, stride, org, ref, stride, org, w64, h64, rng_r, 0, n, best, cq, qoff, subme,
                         nthreads=cores, avx2=avx2)
    rec, lev, ns, dist = O.inter_recon(depth, cur, stride, org, ref, stride, org, w64, h64, level, mv, qp, ctu_begin=0, ctu_end=n,
                                       nthreads=cores, avx2=avx2, intra_slice=tu_flags)         # tu_flags 2 = sign-bit hiding (the x265 default)
    if n == nctu:       # per-picture stages, single-threaded in the restatement
        S = importlib.import_module("x265-yuuki-asuna_amd.stages")
        cuqp = max(qp - 6 * (depth - 8), 0)
        bv, bh = O.deblock_bs_inter(depth, w64, h64, level, mv, ns, avx2=avx2)
        dbk = O.deblock_luma(depth, rec.reshape(-1), stride, org, w64, h64, bv, bh, cuqp, avx2=avx2)
        cnt, off = O.sao_stats(depth, cur.reshape(-1), dbk.reshape(-1), stride, org, w64, h64, nthreads=cores, avx2=avx2)
        # chroma planes: prediction + residual round trip with the luma mvs, chroma edge filter (Bs 2 only), SAO on 32x32 footprints
        cpl = [(F.pad_chroma(clip[cur_index][c], w64, h64_full),
                F.pad_chroma(clip[cur_index - 1][c], w64, h64_full) if ref_planes is None else (ref_planes[c],)) for c in (1, 2)]
        sc, oc = cpl[0][0][1], cpl[0][0][2] + row0 * 32 * cpl[0][0][1]
        qpc = S.chroma_quant_qp(qp, depth)
        crec = [O.inter_recon_chroma(depth, cpl[i][0][0].reshape(-1), cpl[i][1][0].reshape(-1), sc, oc, w64, h64, level, mv, qpc, nthreads=cores, avx2=avx2,
                                      intra_slice=tu_flags)
                for i in range(2)]
        cdb = O.deblock_chroma(depth, crec[0][0], crec[1][0], sc, oc, w64, h64, bv, bh, cuqp, avx2=avx2)
        cstat = [O.sao_stats(depth, cpl[i][0][0].reshape(-1), cdb[i].reshape(-1), sc, oc, w64 // 2, h64 // 2, nthreads=cores, avx2=avx2, ctu=(32, 32), plane_offset=2)
                 for i in range(2)]
        if sao_rdo is not None:
            # the reference's own decision: SAO::rdoSaoUnitCu over the picture on all planes' statistics (oracle/x265_oracle_pipeline6.c)
            lam = np.tile(np.array(sao_rdo["lambdas"], np.int64), (nctu, 1))
            pars, _ = O.sao_rdo(depth, [cnt, cstat[0][0], cstat[1][0]], [off, cstat[0][1], cstat[1][1]], w64 // 64, h64 // 64, lam, sao_rdo["ctx_merge"], sao_rdo["ctx_type"],
                                sao_rdo["entropy_bits"], avx2=avx2)
            par, cpars = pars[0].reshape(-1), [pars[1].reshape(-1), pars[2].reshape(-1)]
        else:
            _, par = O.sao_decide(depth, cnt, off, avx2=avx2)
            cpars = [O.sao_decide(depth, cstat[i][0], cstat[i][1], avx2=avx2)[1] for i in range(2)]
        fin = O.sao_apply(depth, dbk.reshape(-1), stride, org, w64, h64, par, nthreads=cores, avx2=avx2).reshape(rec.shape)
        cfin = []
        for i in range(2):
            cf = O.sao_apply(depth, cdb[i].reshape(-1), sc, oc, w64 // 2, h64 // 2, cpars[i], nthreads=cores, avx2=avx2, ctu=(32, 32))
            out["sao_params_c%d" % i], out["levels_c%d" % i], out["sao_count_c%d" % i] = cpars[i], crec[i][1], cstat[i][0]
            cfin.append(cf)
        dt = time.perf_counter() - t
        # extendPicBorder; a band gets its side margins, the picture's top margin when it is the first band, the bottom margin when the last
        y0 = F.MARGIN_Y + row0 * 64
        inner = fin[y0:y0 + h64, F.MARGIN_X:F.MARGIN_X + w64]
        out.update({"me_best": best, "subpel_mv": mv, "levels": lev, "num_sig": ns, "dist": dist, "sao_count": cnt, "sao_offset_org": off,
                    "sao_params": par,
                    "recon": np.pad(inner, ((F.MARGIN_Y * first_band, F.MARGIN_Y * last_band), (F.MARGIN_X, F.MARGIN_X)), mode="edge")})
        for i in range(2):
            yc = F.CHROMA_MARGIN_Y + row0 * 32
            ci = cfin[i].reshape(-1, sc)[yc:yc + h64 // 2, F.CHROMA_MARGIN_X:F.CHROMA_MARGIN_X + w64 // 2]
            out["recon_c%d" % i] = np.pad(ci, ((F.CHROMA_MARGIN_Y * first_band, F.CHROMA_MARGIN_Y * last_band), (F.CHROMA_MARGIN_X, F.CHROMA_MARGIN_X)), mode="edge")
        return dt, out
    return time.perf_counter() - t, out


def device_outputs(pipe, cur, ref):
    """One frame of the device pipeline (cur searched in ref) with every stage output copied to the host."""
    import torch
    rec = pipe.run(cur, ref)
    torch.cuda.synchronize()
    dt = cur.host.dtype
    out = {"lowres_plane%d" % i: pipe.la.planes[i].cpu().numpy().view(dt) for i in range(4)}
    out.update({"intra_cost": pipe.la.intra_cost.cpu().numpy(), "intra_mode": pipe.la.intra_mode.cpu().numpy(),
                "lowres_costs": pipe.la.lowres_costs.cpu().numpy().view(np.uint16),
                "me_best": pipe.ms.best.cpu().numpy().view(np.uint64), "subpel_mv": pipe.sp.out.cpu().numpy().reshape(-1, 2),
                "levels": pipe.rc.levels.cpu().numpy(), "num_sig": pipe.rc.num_sig.cpu().numpy(), "dist": pipe.rc.dist.cpu().numpy(),
                "sao_count": pipe.sao.count.cpu().numpy(), "sao_offset_org": pipe.sao.offset_org.cpu().numpy(),
                "sao_params": pipe.sao.params.cpu().numpy(),
                "recon": rec.cpu().numpy().view(dt).reshape(cur.host.shape)})
    for i in range(2):
        out["sao_params_c%d" % i] = pipe.sao_c[i].params.cpu().numpy()
        out["sao_count_c%d" % i] = pipe.sao_c[i].count.cpu().numpy()
        out["levels_c%d" % i] = pipe.rc_c[i].levels.cpu().numpy()
        out["recon_c%d" % i] = pipe.final_c[i].cpu().numpy().view(dt)
    return out


def compare_outputs(dev_out, cpu_out):
    """Bit-exact comparison, stage by stage.  Returns {"ok": bool, "stages": {name: "equal" | "<k> of <n> values differ"}}."""
    stages, ok = {}, True
    for k, e in cpu_out.items():
        g = np.asarray(dev_out[k]).reshape(-1)
        e = np.asarray(e).reshape(-1)
        if g.dtype != e.dtype:
            g, e = g.astype(np.int64), e.astype(np.int64)
        if g.shape == e.shape and np.array_equal(g, e):
            stages[k] = "equal"
        else:
            ok = False
            stages[k] = "shape %s vs %s" % (g.shape, e.shape) if g.shape != e.shape else "%d of %d values differ" % (int(np.count_nonzero(g != e)), e.size)
    return {"ok": ok, "stages": stages, "values_compared": int(sum(np.asarray(v).size for v in cpu_out.values()))}


def cpu_baseline(F, clip, rng_r, subme, level, qp, depth=8, target_s=15.0, dev_out=None, sao_rdo=None):
    """The oracle chain (CPU restatement, AVX2 build when the host has AVX2) for one frame, all host cores via OpenMP:
    lookahead -> exhaustive search (best mv) -> sub-pel -> prediction/residual round trip -> deblocking -> SAO statistics.
    Timed on whole frames when one fits the budget, else on a bounded CTU sample; with dev_out (the device pipeline's outputs
    for the same frame pair) ONE whole-frame pass is also compared with them bit for bit.  Returns (cpu_baseline, bit_exact)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O          # cpu_baseline leg only
    avx2 = O.host_has_avx2()
    w64, h64 = F.pad_plane(clip[1][0])[3:5]
    nctu = (w64 // 64) * (h64 // 64)
    cores = effective_cpus()

    def run(n, c=cores):
        return oracle_chain(F, clip, rng_r, subme, level, qp, depth, n, c, avx2, sao_rdo=sao_rdo)

    bit_exact = None
    reps, t, n = 0, 0.0, nctu
    if dev_out is not None:                          # the whole frame once: timed AND compared
        t, cpu_out = run(nctu)
        reps = 1
        bit_exact = compare_outputs(dev_out, cpu_out)
        bit_exact["ctus"] = nctu
        bit_exact["what"] = ("device pipeline == oracle chain (C restatement, pinned against the real reference) for one whole frame "
                             "searched in the previous source frame: lookahead planes / intra costs, integer mvs, sub-pel mvs, luma + chroma levels, "
                             "numSig, SSE, SAO statistics and parameters, and the deblocked + SAO-filtered + border-extended Y / Cb / Cr reconstruction")
    else:
        probe = min(nctu, max(cores, 8))
        t_probe, _ = run(probe)
        n = int(min(nctu, max(probe, probe * target_s / max(t_probe, 1e-3))))
    while t < target_s * 0.66 and reps < 64:      # ~10-15 s of wall time: repeat the (sub-)frame if one pass is shorter
        t += run(n)[0]
        reps += 1
    # the same chain on ONE thread (SURVEY 8(d) asks for both figures): a short CTU sample, per-picture stages left out
    n1 = min(nctu - 1, 8)
    t1, _ = run(n1, 1)
    base = {"value": round(reps * (n / nctu) / t, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "one_thread_value": round((n1 / nctu) / t1, 5),
            "sample": f"{reps} x {n} of {nctu} CTUs of one {clip[0][0].shape[1]}x{clip[0][0].shape[0]} frame, same stages, oracle C + OpenMP on {cores} threads, {t:.1f} s",
            "sample_detail": f"{reps} x {n} of {nctu} CTUs of the same {clip[0][0].shape[1]}x{clip[0][0].shape[0]} frame through the same stages (search keeps only the best mv), "
                      f"oracle C ({'-march=x86-64-v3' if avx2 else 'generic x86-64'}) with OpenMP over CTUs on {cores} threads "
                      f"(the container's CPU quota; {os.cpu_count()} hardware threads visible), {t:.1f} s; an EXHAUSTIVE +-{rng_r} search on the CPU is "
                      f"not what x265 runs at preset slow - the real reference encoder is timed by `bench.py --encoder` (profiles/)"}
    return base, bit_exact


def effective_cpus():
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota (cpu.max) - more OpenMP
    threads than that only get throttled."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def load_traffic(width, height, rng_r, fmt):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = t.get(f"me_{fmt}_{width}x{height}_r{rng_r}")
        return (e["fetch_bytes"] + e["write_bytes"], e["source"]) if e else (None, None)
    except (OSError, ValueError, KeyError):
        return None, None


def load_stage_traffic(width, height, depth):
    """Per-kernel HBM traffic, duration and matrix-core occupancy of the default step's launches from the committed PMC summaries
    (profiles/stage_traffic.json, made by tools/pmc_to_traffic.py from profiles/rNN_bench_pmc*.txt + rNN_bench_kernel_stats*.txt): the
    physical roofline fraction of every stage kernel, weakest first.  One entry per (picture size, bit depth) a profile was taken on
    (round 4: 4K 8-bit, 4K 10-bit, 8K 10-bit - BASELINE configs[2], [3], [4]); None for any other configuration."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "stage_traffic.json")))
    except (OSError, ValueError):
        return None
    t = t.get("configs", {}).get(f"{width}x{height}_d{depth}") if "configs" in t else (t if (width, height, depth) == (3840, 2160, 8) else None)
    if not t:
        return None
    rows = [{"kernel": k, "hbm_bytes": v["fetch_bytes"] + v["write_bytes"], "avg_us": v["avg_us"], "gbytes_per_s": v["gbytes_per_s"],
             "frac_traffic": v["frac_of_8tb"], **({"mfma_busy_frac": v["mfma_busy_frac"]} if v.get("mfma_busy_frac") is not None else {})}
            for k, v in t["kernels"].items()]
    rows.sort(key=lambda r: r["frac_traffic"])
    return {"source": t["source"], "peak_gbytes_per_s": HBM_PEAK_GBS, "kernels": rows,
            "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs): the share of the matrix cores' issue capacity the kernel's int8-limb "
                              "DCT / iDCT products occupy (north_star: DCT MFMA utilisation)"}


def minima_kernel_name(A, depth, rng_r, stub=False):
    """The kernel the step's minima-only search launch runs (asked of the library: it honours the same A/B switches as the launch)."""
    if stub:
        return "stub"
    f = A.lib().x265hip_me_minima_kernel_name
    f.restype, f.argtypes = __import__("ctypes").c_char_p, [__import__("ctypes").c_int] * 2
    return f(depth, rng_r).decode()


def valu_bound(nctu, rng_r, depth, launch_ms):
    """The search kernels are bound by VALU issue, not by HBM (round-3 verdict, weak 2): the SAD instruction stream alone - one
    v_qsad_pk_u16_u8 per 16 pixel-candidates per lane at 8 bits, one v_sad_u16 per 2 at 16 bits - priced at the issue cost the
    microbenchmark measures for that instruction ALONE (tools/ubench/valu_rates.hip, OPs 17 - 19; profiles/valu_rates.json), on 1024 SIMDs."""
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "valu_rates.json")))
    except (OSError, ValueError):
        return None
    nc = 2 * rng_r + 1
    cand = nctu * 4096 * nc * (4 * ((nc + 3) // 4))            # pixel-candidates (motion-vector columns come in groups of four)
    per, ns, name = (16, v["v_qsad_pk_u16_u8_ns"], "v_qsad_pk_u16_u8") if depth == 8 else (2, v["v_sad_u16_ns"], "v_sad_u16")
    floor_ms = cand / per / 64 / 1024 * ns * 1e-6
    return {"bound": "valu", "instruction": name, "pixel_candidates": cand, "ns_per_wave_instruction_per_simd": ns, "floor_ms": round(floor_ms, 4),
            "frac": round(floor_ms / launch_ms, 4), "source": v.get("source")}


# Milliseconds per picture of the BANDED step against the CTU rows per band, one MI355X, MEASURED per (bit depth, picture size) - round-3
# verdict, next 8: rounds 2 - 3 scaled one 4K 8-bit table by the ratio of the whole-picture steps.  Round 5's values (the search kernels of round 5, column groups of a small
# launch dealt over several workgroups at 8 bits): profiles/r05_band_tables.txt (tools/r4_band_tables.sh; round 4's: profiles/r04_band_tables.txt).  Small bands cost launches whose grids no longer fill the
# chip; large bands make the next rank wait longer for its first reference rows.
BANDED_STEP_MS = {
    # (round 6 closing library, visit r9j: one border launch per band, XCD-ordered kernels; round 5's table: 4K 8-bit 4 rows 2.38, 1 row 5.54)
    (8, "4k"): {1: 4.81, 2: 2.68, 3: 2.36, 4: 2.29, 5: 2.40, 6: 2.23, 8: 2.17, 12: 2.09, 17: 2.00},
    (10, "4k"): {2: 4.37, 3: 4.00, 4: 3.74, 6: 3.61, 8: 3.50, 12: 3.26, 17: 3.19},
    (10, "8k"): {4: 13.02, 6: 12.60, 8: 12.25, 12: 12.27, 17: 12.46, 34: 12.19},
    (8, "1080p"): {1: 2.53, 2: 1.26, 3: 0.97, 5: 0.86, 9: 0.80},
}
# configurations without a table of their own borrow the nearest one, scaled by the whole-picture steps (ms, same round)
WHOLE_STEP_MS = {(8, "4k"): 1.62, (10, "4k"): 2.87, (12, "4k"): 2.87, (8, "8k"): 6.5, (10, "8k"): 11.30, (12, "8k"): 11.30, (8, "1080p"): 0.56, (10, "1080p"): 0.93,
                 (12, "1080p"): 0.93}


def pick_band_rows(world, ctu_rows=34, lag_rows_luma=73, depth=8, width=3840):
    """Band size for a ring of `world` ranks: rank r + 1 may start band b once rank r has finished every band its search window and
    interpolation taps reach (bands_needed: b plus the bands that begin inside the lag rows below it), so consecutive ranks run about
    `lag` apart and a rank comes round again after world x lag - throughput is world pictures per max(step, world x lag).  The step
    times are the measured 4K 8-bit ones above scaled to the configuration (WHOLE_STEP_MS), a hand-over is taken as 0.1 ms."""
    size = "8k" if width > 5000 else ("4k" if width > 2500 else "1080p")
    key = (10 if depth > 8 else 8, size)
    if key in BANDED_STEP_MS:
        table, scale = BANDED_STEP_MS[key], 1.0
    else:
        near = (10, "8k") if size == "8k" else ((8, "1080p") if size == "1080p" else (8, "4k"))
        table, scale = BANDED_STEP_MS[near], WHOLE_STEP_MS.get((depth, size), WHOLE_STEP_MS[near]) / WHOLE_STEP_MS[near]
    best = None
    for rows, step8 in table.items():
        if rows > ctu_rows:
            continue
        step = step8 * scale
        nb = -(-ctu_rows // rows)
        ahead = 1 + -(-lag_rows_luma // (rows * 64))            # band periods until the bands a start needs are final
        lag = ahead * step / nb + 0.1
        period = max(step, world * lag)
        if best is None or period < best[0]:
            best = (period, rows)
    return best[1]


MAX_LINE_BYTES = 4096      # the driver keeps a bounded tail of stdout: the one JSON line stays far below it (round-4 verdict: a 37 KB line was never parsed)

# encoder-level legs (tier T3): configuration -> (frames, --frame-threads).  48 frames at cfg3 = more than preset slow's 25-picture lookahead, so the
# lookahead runs ahead of the frame encoders the way it does in a real encode; cfg5 (8K veryslow) is 3 frames - about a minute per leg on 16 cores
ENC_DEFAULTS = {"cfg3": (48, 5), "cfg3f": (24, 5), "cfg4": (24, 5), "cfg2": (96, 3), "cfg1": (8, 1), "cfg5": (3, 5)}


def band_table(depth=8, width=3840):
    """(ms per banded picture by band height, whole-picture ms) for a configuration: its own measured table or the nearest one scaled."""
    size = "8k" if width >= 7000 else ("1080p" if width <= 2000 else "4k")
    key = (depth, size)
    if key in BANDED_STEP_MS:
        return dict(BANDED_STEP_MS[key]), WHOLE_STEP_MS[key]
    near = (10, "8k") if size == "8k" else ((8, "1080p") if size == "1080p" else (8, "4k"))
    scale = WHOLE_STEP_MS.get((depth, size), WHOLE_STEP_MS[near]) / WHOLE_STEP_MS[near]
    return {r: v * scale for r, v in BANDED_STEP_MS[near].items()}, WHOLE_STEP_MS.get((depth, size), WHOLE_STEP_MS[near])


def ring_model(world, rows, gop=0, ctu_rows=34, lag_rows_luma=73, depth=8, width=3840, frames=2000):
    """Pictures per ms the ring delivers by the band model - a discrete simulation instead of round 5's closed form, so that mini-GOPs fit in:
    frame f on rank f % world, ranks work in frame order; a picture may start `lag` after the picture it READS started and cannot end earlier than
    `lag` after that one ended; lag = the band periods until the rows a band's search window reaches are final + a hand-over.  gop = 0: every
    picture reads the one before it (round 5's chain: N pictures per max(step, N x lag)); gop = G: every picture reads the newest multiple of G
    before it (FrameParallelRing(gop=G))."""
    table, _ = band_table(depth, width)
    step = table[rows]
    nb = -(-ctu_rows // rows)
    lag = (1 + -(-lag_rows_luma // (rows * 64))) * step / nb + 0.1
    free, start, end = [0.0] * world, {}, {}
    for f in range(frames):
        r = f % world
        a = (f - 1 if not gop else ((f - 1) // gop) * gop) if f > 0 else None
        s_ = max(free[r], start[a] + lag if a is not None else 0.0)
        e_ = max(s_ + step, end[a] + lag if a is not None else 0.0)
        start[f], end[f], free[r] = s_, e_, e_
    return frames / end[frames - 1]


def pick_band_rows_gop(world, gop, ctu_rows=34, lag_rows_luma=73, depth=8, width=3840):
    """Band size for the mini-GOP ring: the measured size with the largest modelled throughput."""
    table, _ = band_table(depth, width)
    best = max((ring_model(world, rows, gop, ctu_rows, lag_rows_luma, depth, width), -rows) for rows in table if rows <= ctu_rows)
    return -best[1]


def seam_config(key):
    """How the encoder legs configure the consumer services for BASELINE configuration `key` - by what was MEASURED on one box, every leg against the
    host-only control (profiles/r05_seam_matrix.txt, tools/r5_seam_matrix.sh), not by how many lookups get served (round-4 verdict, next 4):
      * 8 bits: NO SAD lookups.  With the host path equal they add nothing at cfg3 (7.95 fps without, 7.89 - 7.98 with, 9.4 GB instead of 14.5 - 22.7 GB
        downloaded per 48 frames) and cost 6 % on the fade (4.46 against 4.10 - 4.20: the searches leave the windows, hit rate 0.11 - 0.17);
      * above 8 bits: the 32x32 / 64x64 rasters only (`min_level` 2): 1.88 fps like level 1, 8.9 instead of 13.2 GB; without the SAD seam 1.85;
      * always: sub-sample comparisons, lookahead frame costs, AQ and weightAnalyse from the device; the binding's hit-rate gate (binding/x265hip_x265_binding.cpp) stops opening
        pairs when fewer than half of the lookups of a window hit; everything the services do not answer takes the host-only control's split SADs."""
    depth = CFG_DEPTH.get(key, 8)
    # round 6: the SUB-SAMPLE COST TABLES (x265hip_cost_stream behind MotionEstimate::subpelCompare: records of the refinement's SATD costs around each PU's best integer
    # vector, one candidate, the 85-position set of --subme 4 also under --subme 3) - one box, interleaved, against round 5's legs (tools/r6_cost_ab.sh,
    # profiles/r06_cost_ab_*.txt): cfg3 7.87 -> 8.42 fps with tables + phase planes, 8.63 with the tables ALONE (11.1 instead of 9.4 + 11.1 GB downloaded: the phase planes'
    # transfers cost more than the comparisons they still serve); cfg4 1.89 -> 2.21 with both (2.14 tables alone: at 10 bits the host's own 16-bit interpolation of what
    # the tables miss is the dearer path).  So: 8 bits without a fade = tables alone; everything else = tables first, phase planes behind them.
    base = {"range": 12, "centre_range": 57, "layout": 1, "min_pu": 16, "verify": False, "lookahead": True, "subpel": key != "cfg3" or os.environ.get("X265HIP_SEAM_PHASES") == "1",
            "streamed": True, "aq": True,
            "weight_analyse": True, "split_rest": True, "no_sad": depth == 8 and os.environ.get("X265HIP_SEAM_SAD_8BIT") != "1",
            "min_level": int(os.environ.get("X265HIP_SEAM_MIN_LEVEL", "2")),
            "cost": os.environ.get("X265HIP_SEAM_COST", "1") != "0", "cost_candidates": 1, "cost_window": 8, "cost_set_subme": 4, "cost_centre_range": 57}
    if key == "cfg5":       # 8K: a reference picture's phase planes are 3.4 GB of pinned memory, three frames need few resident pairs / views
        return {**base, "slots": 12, "subpel_slots": 4, "pictures": 8, "min_level": 1, "cost_slots": 12, "cost_pictures": 12, "cost_views": 6}
    return {**base, "slots": 24 if depth == 8 else 40, "subpel_slots": 12, "pictures": 24, "cost_slots": 24 if depth == 8 else 40, "cost_pictures": 40, "cost_views": 12}


def encoder_plan(args):
    """The legs of the encoder-level measurement: every configuration of --encoder with the plain reference build, then the metric's
    configuration and the 10-bit one again on the AVX2 auto-vectorised build ("vs host AVX2": the hand-written NASM kernels cannot be
    assembled here - no nasm - so g++ -O3 -march=x86-64-v3 of the reference's C path is the closest thing that can be timed)."""
    keys = [k for k in args.encoder.split(",") if k]
    plan = []
    for key in keys:
        nf, ft = ENC_DEFAULTS.get(key, (24, 5))
        plan.append({"name": key, "key": key, "tables": args.encoder_tables.split(","), "frames": args.encoder_frames or nf,
                     "frame_threads": args.encoder_frame_threads or ft, "seam": seam_config(key), "build": ""})
    for key in [k for k in ("cfg3", "cfg4") if k in keys]:
        nf, ft = ENC_DEFAULTS[key]
        plan.append({"name": key + "_v3", "key": key, "tables": ["c", "csse"] + [t for t in args.encoder_tables.split(",") if t != "c"], "frames": args.encoder_frames or nf,
                     "frame_threads": args.encoder_frame_threads or ft, "seam": seam_config(key), "build": "v3"})
    return plan


def encoder_leg(args, timeout_s=None):
    """Tier T3: the real reference encoder on the host cores - its C table, the host-only control, the seams - in a
    CHILD process (tools/encoder_bench.py --plan): a crash or a hang of a leg costs that leg, never the headline of this line.  The child
    writes its result file after every leg; whatever is there when it ends (or is stopped) is reported.  Returns {"encoder": ..,
    "encoder_summary": ..}."""
    import subprocess
    import tempfile
    timeout_s = timeout_s or float(os.environ.get("X265HIP_ENCODER_TIMEOUT_S", "900"))
    fd, path = tempfile.mkstemp(prefix="x265hip_encoder_", suffix=".json")
    os.close(fd)
    enc, err = {}, None
    try:
        # legs are not STARTED after 60 % of the limit (the longest leg is a fifth of the default plan): the child then ends by itself and the line is printed
        cmd = [sys.executable, os.path.join(ROOT, "tools", "encoder_bench.py"), "--plan", json.dumps(encoder_plan(args)), "--out", path, "--deadline-s", str(0.6 * timeout_s)]
        try:
            rc = subprocess.run(cmd, stdout=sys.stderr, stderr=sys.stderr, timeout=timeout_s, cwd=ROOT).returncode
            if rc != 0:
                err = f"encoder legs exited with status {rc}"
        except subprocess.TimeoutExpired:
            err = f"encoder legs stopped after {timeout_s:.0f} s"
        try:
            enc = json.load(open(path)).get("encoder", {})
        except (OSError, ValueError):
            pass
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    if err:
        enc["error"] = err
    return {"encoder": enc, "encoder_summary": encoder_summary(enc)}


def encoder_leg_numbers(c):
    """One leg (tools/encoder_bench.run_config result) as numbers only."""
    if not isinstance(c, dict) or "c" not in c:
        return None
    sm, cs = c.get("seam", {}), c.get("csplit", {})
    rep = sm.get("seam", {})
    r = {"frames": c["c"]["frames"], "F": int(c["options"]["frame-threads"]), "cores": c["pool_threads"], "c_fps": c["c"]["fps"],
         "control_fps": cs.get("fps"), "seam_fps": sm.get("fps"), "md5_equal": sm.get("md5_equal_to_c_table"),
         "x_c": round(sm["fps"] / c["c"]["fps"], 3) if sm.get("fps") else None,
         "x_control": round(sm["fps"] / cs["fps"], 3) if sm.get("fps") and cs.get("fps") else None,
         "hit_rate": rep.get("lookup_hit_rate"),
         "gb_down": round(((rep.get("bytes_downloaded") or 0) + (rep.get("subpel_seam", {}).get("bytes_downloaded") or 0) +
                           (rep.get("cost_seam", {}).get("bytes_downloaded") or 0)) / 1e9, 2) if rep else None,
         "cost_share": rep.get("cost_seam", {}).get("served_share_of_satd_comparisons_with_context") if rep else None}
    sat = rep.get("subpel_seam", {}).get("satd_lookups_served")
    if sat is not None:
        r["satd_served"] = sat
    if "csse" in c:          # v3 legs: the C table + the reference's SSE intrinsic transforms (common/vec), the strongest host table that can be built here
        r["c_sse_fps"] = c["csse"].get("fps")
        r["sse_md5_equal"] = c["csse"].get("md5_equal_to_c_table")
    return r


def encoder_summary(enc):
    """Numbers only (the per-leg diagnostics stay in bench_detail.json): per configuration the reference's C table, the host-only control
    (C table with split sad_x3 / sad_x4, no GPU) and the seams, frames/s; md5 equality of the bitstreams; hit rate; GB downloaded."""
    legs = {k: encoder_leg_numbers(v) for k, v in enc.items() if k != "error"}
    s = {"kind": "reference x265 3.5, C primitives (no nasm in the image); *_v3 = g++ -march=x86-64-v3", **{k: v for k, v in legs.items() if v}}
    if "error" in enc:
        s["error"] = str(enc["error"])[:160]
    skipped = [k for k, v in enc.items() if isinstance(v, dict) and "skipped" in v]
    if skipped:
        s["legs_not_started_deadline"] = skipped
    return s


def write_detail(out, args):
    """Everything the one-line record leaves out - per-leg encoder diagnostics, the stage-by-stage comparison, per-kernel rooflines,
    checksums, prose - as a file (bench_detail.json beside bench.py and, on a GPU visit, under gpurun_out/) and on stderr."""
    text = json.dumps(out, indent=1, sort_keys=False)
    paths = [os.environ.get("X265HIP_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(text + "\n")
        except OSError:
            pass
    sys.stderr.write("bench.py detail (also in %s):\n%s\n" % (paths[0], json.dumps(out)))
    sys.stderr.flush()


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(out, minimal=False):
    """The ONE line rank 0 prints: the contract's fields plus `roofline`, `cpu_baseline`, `bit_exact` and the encoder summary, numbers
    and short labels only - below MAX_LINE_BYTES by construction (tests/test_bench_line.py).  `minimal` drops the optional objects."""
    cfg = out.get("config", {})
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    c = {"workload": str(cfg.get("workload", ""))[:200], "parallelism": str(cfg.get("parallelism", ""))[:120]}
    c.update(_pick(cfg, ("ctus_per_frame", "band_rows", "sharding")))
    if "ring" in cfg:
        c["ring"] = _pick(cfg["ring"], ("ranks_seen", "transport", "communicators", "bands_per_frame", "refs", "gop", "model_x_one_gpu", "band_wait_ms_per_frame_max_over_ranks", "comm_init_s"))
    line["config"] = c
    if "replicas" in out:
        line["replicas"] = _pick(out["replicas"], ("value", "ms_per_step", "unit"))
    r = out.get("roofline") or {}
    rl = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "launch_ms"))
    if isinstance(r.get("valu"), dict):
        rl["valu"] = _pick(r["valu"], ("floor_ms", "frac"))
    line["roofline"] = rl
    if "cpu_baseline" in out:
        cb = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "one_thread_value"))
        cb["sample"] = str(out["cpu_baseline"].get("sample", ""))[:160]
        line["cpu_baseline"] = cb
    if "bit_exact" in out:
        line["bit_exact"] = out["bit_exact"]
        if isinstance(out.get("bit_exact_detail"), dict):
            line["bit_exact_values"] = out["bit_exact_detail"].get("values_compared")
    if not minimal:
        if isinstance(out.get("stages_ms"), dict):
            line["stages_ms"] = {k[:24]: v for k, v in list(out["stages_ms"].items())[:16]}
        if "encoder_summary" in out:
            line["encoder_summary"] = out["encoder_summary"]
        line["detail"] = "bench_detail.json"
    return line


class _StubPipeline:
    """X265HIP_BENCH_STUB=1 (tests/test_dist_cpu.py, no GPU): a CPU stand-in with the interface main() uses of stages.FramePipeline /
    stages.BandedFramePipeline, so that the control flow of `bench.py --gpus N` - process group, ring set-up, band hand-offs, barriers, the
    replicas pass, max-over-ranks timing, the one-line record - runs end to end on gloo.  Its "encode" of a band is the average of the source and
    the reference rows (so a band that ran before its reference rows arrived changes the checksum).  Numbers from it mean nothing."""

    def __init__(self, P, pic, rng, depth, band_rows=0):
        import torch
        self.torch = torch
        self.ms = P.MotionSearch(pic.w64, pic.h64, rng, depth, torch.device("cpu"), want_surf=False)
        self.lcb = None
        rows = pic.h64 // 64
        self.bands = [(r, min(band_rows, rows - r)) for r in range(0, rows, band_rows)] if band_rows else [(0, rows)]
        self.geo = (pic.stride, pic.t.numel() // pic.stride, pic.stride_c, pic.c[0].numel() // pic.stride_c)
        self.out = [torch.zeros_like(t) for t in pic.planes()]
        self.final = self.out

    def _mix(self, cur, ref, row0, n):
        st, rows, sc, rows_c = self.geo
        my, myc = (rows - self.ms.h64) // 2, (rows_c - self.ms.h64 // 2) // 2
        last = (row0 + n) * 64 == self.ms.h64
        spans = [((0 if row0 == 0 else my + row0 * 64) * st, (rows if last else my + (row0 + n) * 64) * st)] + \
                [((0 if row0 == 0 else myc + row0 * 32) * sc, (rows_c if last else myc + (row0 + n) * 32) * sc)] * 2
        for o, c, r, (a, b) in zip(self.out, cur.planes(), ref.planes(), spans):
            o[a:b] = (c[a:b] >> 1) + (r[a:b] >> 1)

    # whole-picture interface (stages.FramePipeline)
    def run(self, cur, ref, mark=None):
        self._mix(cur, ref, 0, self.ms.h64 // 64)
        self.final = self.out
        return self.out[0]

    def final_planes(self):
        return self.final

    def swap_output(self, spare):
        outs, self.out = self.out, list(spare)
        return outs

    def launch_lookahead_costs(self):
        pass

    def checksum(self):
        return {"recon_%s" % n: int(p.to(self.torch.int64).sum().item()) for n, p in zip(("y", "cb", "cr"), self.final)}

    # banded interface (stages.BandedFramePipeline)
    def begin_frame(self, cur):
        self.cur = cur

    def run_band(self, b, cur, ref):
        self._mix(cur, ref, *self.bands[b])

    def end_frame(self):
        self.final = self.out

    def band_context(self, b):
        import contextlib
        return contextlib.nullcontext()

    def capture(self, cur, ref):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # ~0.5 s of timed region at 4K (round 1 timed 46 ms: too short for the driver's sampler)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)      # BASELINE metric: "4K preset=slow" = configs[2]
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--range", type=int, default=57)          # reference default merange (param.cpp:198)
    ap.add_argument("--subme", type=int, default=3)           # preset slow (param.cpp:397-539); medium = 2
    ap.add_argument("--level", type=int, default=2)           # 32x32 blocks in the reconstruction stage
    ap.add_argument("--qp", type=int, default=27)
    ap.add_argument("--depth", type=int, default=8, choices=[8, 10], help="10 = the Main10 configurations (configs[3], [4])")
    ap.add_argument("--surface", action="store_true",
                    help="the search launch also WRITES the SAD surfaces of all 85 PUs (11.9 GB algorithmic / 4.9 GB physical per 4K picture, rounds 1 - 3's default). "
                         "Nothing in this closed loop reads them - its stages consume the per-PU minima - so since round 4 the default search keeps the minima only "
                         "(round-3 verdict, next 6: stop writing bytes nobody reads); the surfaces are the product of the consumer services (x265hip_me_stream), "
                         "measured in the encoder leg")
    ap.add_argument("--no-surface", action="store_true", help="(the default since round 4; kept so that older command lines still parse)")
    ap.add_argument("--sharding", choices=["ring", "gop"], default="ring",
                    help="N > 1: ring = the reference's frame parallelism with its real dependency (frame f on rank f %% N searches frame f - 1, "
                         "handed on band by band; DESIGN.md section 6); gop = every rank encodes its own closed group of pictures with its own "
                         "reference chain (segment-parallel encoding: no data-path exchange at all, an upper bound, not what x265 -F does)")
    ap.add_argument("--ring-gop", type=int, default=5,
                    help="N > 1, ring: mini-GOP length G (round 6).  Frames that are multiples of G are anchors (they read the anchor before them); the G - 1 "
                         "pictures between two anchors are non-referenced and read the anchor before them, so their ranks are independent of each other and "
                         "the chain that bounds the ring is the anchors' - x265's default structure, bframes 4 (common/param.cpp:166-168), with the dependents "
                         "predicted from one side.  Every picture is the same step.  0 = round 5's P-only chain (frame f reads frame f - 1)")
    ap.add_argument("--split", type=int, default=1,
                    help="search -> sub-pel refinement -> reconstruction in this many parts of whole CTU rows, part k refined / reconstructed on a side "
                         "stream while part k + 1 is searched (same results; needs --parallel-planes 1; 1 = the picture in one piece - the default: "
                         "next to other kernels the record-per-lane search loses more than the overlap hides, 2.94 against 2.29 ms at 4K)")
    ap.add_argument("--parallel-planes", type=int, default=1, choices=[0, 1],
                    help="1 = after the sub-pel stage Y, Cb and Cr run their reconstruction -> deblocking -> SAO -> border chains on three HIP "
                         "streams and the lookahead runs next to the search (same launches, same outputs); 0 = every launch on one stream")
    ap.add_argument("--subpel-planes", type=int, default=1, choices=[0, 1],
                    help="1 = the sub-pel stage reads its candidates from the reference picture's 15 phase planes (one x265hip_phase_planes launch "
                         "per frame, inside the timed step); 0 = it interpolates every candidate tile itself")
    ap.add_argument("--surf-format", choices=["packed_b", "packed_t", "packed", "i32"], default="packed_b",
                    help="SAD surface records: packed = u16 for the 8x8/16x16 levels (X265HIP_SURF_PACKED), packed_t = the same records chunk-major "
                         "inside a motion-vector row (X265HIP_SURF_PACKED_T), packed_b = the same records in contiguous blocks of 64 "
                         "(X265HIP_SURF_PACKED_B: one block per wavefront step of the record-per-lane kernel), i32 = all int32")
    ap.add_argument("--search", choices=["full", "dia", "hex", "umh", "star", "sea"], default="full",
                    help="full = exhaustive search (SAD surfaces + best mv) + sub-pel stage; dia/hex/umh/star/sea = the reference's pattern "
                         "searches run by the device-side search driver (x265hip_me_search), predictor (0,0)")
    ap.add_argument("--lookahead-batch", type=int, default=0,
                    help="pictures per launch of the lookahead's P-frame cost estimate, which runs ahead on a side stream (0 = stage off)")
    ap.add_argument("--band-rows", type=int, default=0,
                    help="CTU rows per band of the frame-parallel ring (N > 1, or --banded): a band is searched / reconstructed / filtered as a slice "
                         "of its own and handed to the next rank as soon as it is final.  0 = chosen from the number of ranks (pick_band_rows)")
    ap.add_argument("--band-streams", type=int, default=3, help="banded pipeline: band b runs on HIP stream b %% this, so a band's search overlaps the "
                    "previous band's reconstruction / loop filters (1: every band on the caller's stream)")
    ap.add_argument("--band-graphs", type=int, default=0, help="banded pipeline: replay each band's launches as one HIP graph (0: launch by launch)")
    ap.add_argument("--banded", action="store_true", help="run the banded pipeline on one GPU too (measures what the band granularity costs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1, ring: skip the second timed pass without exchange (`replicas` of the line)")
    ap.add_argument("--no-verify", action="store_true", help="skip the bit-exact comparison of one whole frame with the oracle chain")
    ap.add_argument("--sao-decision", default="rdo", choices=["rdo", "standin"],
                    help="rdo (default): the reference's own rate-distortion decision of the SAO parameters on the device (x265hip_sao_rdo = "
                         "SAO::rdoSaoUnitCu: offset iteration, CABAC bit counts, merge candidates; ~0.3 ms of serial CTU-row walk at 4K); standin: round 2's "
                         "distortion-only choice (x265hip_sao_decide, 0.01 ms) - not what x265 decides")
    ap.add_argument("--ref-handoff", choices=["swap", "copy"], default="swap",
                    help="one GPU: how the picture a step produced becomes the next step's reference - swap = the decoded-picture buffer "
                         "ping-pongs (no copy), copy = three plane copies per step (rounds 1-2)")
    ap.add_argument("--no-encoder", action="store_true",
                    help="skip the encoder-level leg (tier T3): the REAL reference encoder (oracle/_ref) on BASELINE configs[2] - 4K, preset slow, "
                         "--me star - with its own C table and with the stage-level seams (integer-search SADs from x265hip_me_cache surfaces, the "
                         "lookahead's frame cost / intra estimates from x265hip_lowres_cost_host / x265hip_lowres_intra_host); fps + bitstream md5")
    ap.add_argument("--encoder", default="cfg3,cfg3f,cfg4,cfg5,cfg2", help="configurations of the encoder-level leg (tools/encoder_bench.py: cfg1,cfg2,cfg3,cfg3f,cfg4,cfg5); the "
                    "default shows BASELINE configs[2] (the metric's), the same on a fade (weighted references), configs[3] (4K 10-bit), configs[4] (8K 10-bit, 3 frames) and configs[1] (1080p)")
    ap.add_argument("--encoder-frames", type=int, default=0, help="frames of every encoder-level leg (0 = per configuration: 48 for cfg3 - more than the "
                    "25-picture lookahead of preset slow, so that the lookahead runs ahead of the frame encoders the way it does in a real encode - 24 for cfg3f / cfg4, 96 for cfg2)")
    ap.add_argument("--encoder-frame-threads", type=int, default=0, help="--frame-threads of the encoder legs (0 = what x265 picks itself for 16 cores at that "
                    "picture size: 5 at 4K, 3 at 1080p; the row-granular seams serve at any value)")
    ap.add_argument("--encoder-tables", default="c,csplit,seam",
                    help="c = reference C table; csplit = HOST-ONLY control: the C table with sad_x3 / sad_x4 answered by N calls of its own sad (what the seam stubs do "
                         "for a candidate they cannot serve - g++ vectorises the single-reference loop and not the multi-reference ones, so this alone is faster); "
                         "seam = the services behind the stage-level seams; hip = per-call stubs (slow)")
    ap.add_argument("--prims", action="store_true",
                    help="instead of the pipeline line, print the per-family table of the batch-layer kernels with the CPU paths timed beside "
                         "them (tools/bench_prims.py: bench.py's cpu_baseline leg at primitive level)")
    ap.add_argument("--search-probe", action="store_true",
                    help="instead of the pipeline line, time the search drivers over every PU of a frame with the CPU restatement beside "
                         "them (tools/search_probe.py)")
    ap.add_argument("--lookahead-probe", action="store_true",
                    help="instead of the pipeline line, time the lookahead P-frame cost estimate (x265hip_lowres_cost) for one picture and "
                         "for batches of independent pictures on separate streams, CPU restatement beside it (tools/lookahead_probe.py)")
    args, rest = ap.parse_known_args()
    if args.prims or args.search_probe or args.lookahead_probe:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sys.argv = [sys.argv[0]] + rest + ([] if args.prims else [str(args.width), str(args.height)])
        mod = importlib.import_module("bench_prims" if args.prims else ("search_probe" if args.search_probe else "lookahead_probe"))
        if hasattr(mod, "main"):
            mod.main()
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # X265HIP_BENCH_STUB=1: the N > 1 control flow on CPU tensors over gloo with a stand-in for the stages (_StubPipeline; tests/test_dist_cpu.py) -
    # a dry run of THIS FILE's logic, never a measurement and never a fallback: without the variable a missing GPU is fatal
    stub = os.environ.get("X265HIP_BENCH_STUB") == "1"
    if not torch.cuda.is_available() and not stub:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # X265HIP_BENCH_BACKEND=gloo: functional dry run of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the bands
    # travel through host memory) - for checking the ring / band code on real kernels, never for numbers
    backend = "gloo" if stub else os.environ.get("X265HIP_BENCH_BACKEND", "nccl")
    if stub:
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a, **k: None          # the stand-in has nothing to wait for
    else:
        local = local % torch.cuda.device_count() if backend != "nccl" else local
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a hand-off that never completes must fail the run within minutes, not hold the node until the driver's limit
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=240))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=240))

    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
    S = importlib.import_module("x265-yuuki-asuna_amd.stages")
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")

    sao_rdo = None
    if args.sao_decision == "rdo":
        HT = importlib.import_module("x265-yuuki-asuna_amd.host_tables")
        tabs = HT.load()                                   # host-built tables (entropy bit costs, lambda): handed to the stage, never recomputed there
        cu_qp = max(args.qp - 6 * (args.depth - 8), 0)     # --qp is the quantiser's QP (+ QP_BD_OFFSET); SAO prices bits at the CU's own QP
        cm, ct = HT.sao_contexts(HT.SLICE_P, cu_qp)
        sao_rdo = {"lambdas": HT.sao_lambdas(tabs, cu_qp), "ctx_merge": cm, "ctx_type": ct, "entropy_bits": tabs["entropy_bits"]}
    nclip = 4
    clip = F.synth_clip(args.width, args.height, nclip, depth=args.depth, seed=265 + rank)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    pipe = _StubPipeline(P, pics[0], args.range, args.depth) if stub else S.FramePipeline(pics[0].w64, pics[0].h64, args.depth, dev, rng=args.range, subme=args.subme, level=args.level,
                           qp=args.qp, want_surf=args.surface, packed=({"packed": True, "packed_t": "t", "packed_b": "b"}.get(args.surf_format, False) if args.depth == 8 else False),
                           lookahead=(args.width, args.height), search=args.search, deblock=True, sao=True, lookahead_cost_batch=args.lookahead_batch,
                           chroma=True, sao_apply=True, sign_hide=True, subpel_planes=bool(args.subpel_planes),
                           parallel_planes=bool(args.parallel_planes), split=args.split, sao_rdo=sao_rdo)
    ref_pic = pics[0].like([p.clone() for p in pics[0].planes()])     # the reference every rank searches in (starts as frame 0): Y, Cb, Cr
    gop = world > 1 and args.sharding == "gop"
    fp = P.FrameParallel(rank, 1 if gop else world)          # gop: the hand-off is this rank's own copy
    banded = (world > 1 and not gop) or args.banded
    ring_gop = max(0, args.ring_gop) if (world > 1 and not gop) else 0
    if banded and not args.band_rows:
        kw = dict(ctu_rows=(args.height + 63) // 64, lag_rows_luma=args.range + 16, depth=args.depth, width=args.width)
        args.band_rows = (pick_band_rows_gop(world, ring_gop, **kw) if ring_gop else pick_band_rows(world, **kw)) if world > 1 else 4
    if banded:
        # N > 1: the reference's real frame-parallel dependency - frame f (rank f % N) searches frame f - 1, band by band (pipeline.FrameParallelRing)
        bp = _StubPipeline(P, pics[0], args.range, args.depth, band_rows=args.band_rows) if stub else S.BandedFramePipeline(pics[0].w64, pics[0].h64, args.depth, dev, band_rows=args.band_rows, rng=args.range, subme=args.subme, level=args.level,
                                   qp=args.qp, want_surf=args.surface,
                                   # a small band's grid (4 CTU rows = 240 workgroups) is too small for the record-per-lane kernel's 4-wavefront
                                   # workgroups (one per CU): such bands use the record-contiguous packed format of the row-walking kernel;
                                   # from 5 rows on the record-per-lane kernel wins (6 rows: 2.58 against 2.81 ms per picture)
                                   packed=(args.surf_format != "i32" and args.depth == 8) and
                                          (("b" if args.surf_format == "packed_b" else "t") if args.band_rows >= int(os.environ.get("X265HIP_BAND_T_ROWS", "5")) and args.surf_format in ("packed_t", "packed_b") else True),
                                   lookahead=(args.width, args.height), deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True,
                                   graphs=bool(args.band_graphs), streams=args.band_streams, sao_rdo=sao_rdo)
        # (a band's Cb / Cr chains on side streams next to Y, like the whole-picture step, LOSE at every band size: 17 rows 1.97 -> 2.79 ms, 4 rows 2.35 -> 5.90 ms, profiles/r06_band_ab.txt)
        # (bands keep every launch on one stream: side streams for the chroma chains change nothing at band size - 3.93 vs 3.97 ms at 4 rows)
        geom = (pics[0].stride, F.MARGIN_Y, pics[0].stride_c, F.CHROMA_MARGIN_Y)
        # The band hand-off goes through the library's C ABI (x265hip_comm_* + x265hip_recon_publish_rows: pipeline.AbiTransport), so the code
        # RCCL executes here is the code a C++ host would run; X265HIP_RING_TRANSPORT=dist selects torch.distributed's own point-to-point
        # operations instead (also the fallback when the C-ABI communicators cannot be built - decided collectively, never by one rank alone).
        transport_name = "dist"
        transport = None
        comm_init_s = None
        # X265HIP_RING_TRANSPORT=bcast (round 6; needs mini-GOPs): ONE communicator over all ranks, an anchor's band = ncclBroadcast rooted at its rank (pipeline.AbiBcastTransport,
        # x265hip_recon_publish_rows with peer = -1) - the transport `north_star` names, as an A/B switch next to the point-to-point flows; dist_bcast = its torch.distributed twin
        want = os.environ.get("X265HIP_RING_TRANSPORT", "abi")
        if want in ("bcast", "dist_bcast") and not ring_gop:
            want = "abi" if want == "bcast" else "dist"
        if world > 1 and backend == "nccl" and want in ("abi", "bcast"):
            ok = 1
            t_comm = time.perf_counter()
            try:
                # mini-GOPs: an anchor's bands go to every rank that encodes one of the next G pictures - flows of distance 1 .. min(G, N - 1)
                transport = P.AbiBcastTransport(rank, world, dev, args.depth, geom, pics[0].h64) if want == "bcast" else \
                    P.AbiTransport(rank, world, dev, args.depth, geom, pics[0].h64, refs=min(ring_gop, world - 1) if ring_gop else 1)
                transport.setup(dev)
            except Exception as e:          # noqa: BLE001 - any failure means "use the other transport", on every rank
                sys.stderr.write(f"bench.py rank {rank}: C-ABI ring transport unavailable ({e!r}); torch.distributed point-to-point instead\n")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            comm_init_s = time.perf_counter() - t_comm       # unique ids + ncclCommInitRank of every directed flow (+ the agreement all-reduce)
            if int(flag.item()):
                transport_name = want
            else:
                if transport is not None:
                    transport.close()
                transport = None
        if transport is None and world > 1 and want in ("bcast", "dist_bcast") and ring_gop:
            transport = P.DistBcastTransport(rank, world, stage_through_host=backend != "nccl")
            transport_name = "dist_bcast"
        ring = P.FrameParallelRing(rank, world, bp.bands, lag_rows_luma=args.range + 16,      # search window + 8-tap interpolation + sub-pel drift
                                   stage_through_host=backend != "nccl", transport=transport, gop=ring_gop)
        if transport is None or transport_name == "dist_bcast":
            ring.make_groups(device=dev)
        total_frames = (args.warmup + args.steps) * world
        ranks_seen = 1
        if world > 1:
            # diagnostics for the first hardware runs (round-3 verdict, next 8): how many ranks does the collective layer really see?
            seen = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(seen)
            ranks_seen = int(seen.item())
        bp.begin_frame(pics[1])                         # allocates the output planes
        if args.band_graphs:                            # set-up: every (band, source picture) of the resident clip recorded as a HIP graph
            for pc in pics[1:]:
                bp.capture(pc, ref_pic)
            torch.cuda.synchronize()

    def step(i):
        cur = pics[1 + i % (nclip - 1)]
        if banded:
            ring.finish()                               # the previous picture's bands have left before its planes are rewritten
            bp.begin_frame(cur)
            ring.run_frame(i, geom, ref_pic.planes(), bp.final_planes(), lambda b, row0, n: bp.run_band(b, cur, ref_pic), total_frames=total_frames,
                           band_context=bp.band_context)
            bp.end_frame()
            if world == 1:
                fp.exchange(ref_pic.planes(), bp.final_planes())
            return
        pipe.run(cur, ref_pic)
        # the filtered reconstruction (Y, Cb, Cr) becomes the next reference.  One GPU: the decoded-picture buffer ping-pongs - the planes the
        # step wrote ARE the next reference and the replaced reference's planes receive the next picture (three plane copies and the gaps
        # around them were 50 - 70 us of the step, profiles/r03_step_timeline.txt; --ref-handoff copy keeps them).  N > 1: the frame-parallel
        # hand-off, the last rank's picture becomes everyone's reference.
        new = pipe.swap_output(ref_pic.planes()) if (world == 1 and args.ref_handoff == "swap") else None
        if new is not None:
            ref_pic.t, ref_pic.c = new[0], (list(new[1:3]) if len(new) >= 3 else None)
        else:
            fp.exchange(ref_pic.planes(), pipe.final_planes())

    # The interpreter's cycle collector is parked for the warm-up AND the timed loop: one full collection over torch's object graph is a 30 - 60 ms
    # pause of the launching thread (seen in the dispatch timeline, profiles/r02_band_timeline.txt: a single 38 ms hole in a 10-step
    # run) - nothing the encoder-side caller of the C ABI would have, and with a few dozen steps it decides the average.  It runs BEFORE the
    # warm-up steps (round 5): collecting between warm-up and timed loop left the GPU idle for those 30 - 60 ms and the first timed steps ran on a
    # device that had clocked down - 1.69 ms per step over the driver's 20 steps against 1.63 over 200.
    import gc
    gc.collect()
    gc.disable()
    for i in range(args.warmup):
        step(i)
    if banded and world > 1 and not stub:
        ring.time_waits()                            # two device events per band: how long its stream waits for the reference rows it reads
    pipe.launch_lookahead_costs()                    # no lookahead work of the warm-up frames leaks into the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if banded:
        ring.drain(geom, ref_pic.planes(), total_frames)     # a broadcast transport: the anchors this rank has not joined yet (nothing to do for point-to-point flows)
        ring.finish()
    pipe.launch_lookahead_costs()                    # flush the incomplete batch: all K pictures are scored inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    replicas = None
    if world > 1 and not gop and not args.no_replicas:
        # the comparison line beside the ring (round-4 verdict, next 6): the same K steps with NO exchange - every rank its own closed
        # group of pictures (what --sharding gop times) - so the first hardware record shows ring and replicas side by side
        rp = [p.clone() for p in ref_pic.planes()]
        rpic = pics[0].like(rp)

        def rstep(i):
            pipe.run(pics[1 + i % (nclip - 1)], rpic)
            new = pipe.swap_output(rpic.planes())
            if new is not None:
                rpic.t, rpic.c = new[0], (list(new[1:3]) if len(new) >= 3 else None)
        for i in range(min(args.warmup, 3)):
            rstep(i)
        pipe.launch_lookahead_costs()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tr = time.perf_counter()
        for i in range(args.steps):
            rstep(i)
        pipe.launch_lookahead_costs()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        rt = torch.tensor([time.perf_counter() - tr], dtype=torch.float64, device=dev)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        replicas = {"value": round(world * args.steps / float(rt.item()), 3), "unit": "frames/s", "ms_per_step": round(1000.0 * float(rt.item()) / args.steps, 4),
                    "what": "--sharding gop: every rank its own closed group of pictures, no exchange - the upper bound the ring is compared with"}
    gc.enable()
    ring_wait = None
    if banded and world > 1:                         # every rank: the maximum over ranks is a collective
        wms, _ = (0.0, 0) if stub else ring.wait_ms()
        wt = torch.tensor([wms / max(1, args.steps)], dtype=torch.float64, device=dev)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        ring_wait = round(float(wt.item()), 4)
    if banded:
        csum = {"recon_%s" % n: int(p.view(torch.uint8).to(torch.int64).sum().item()) for n, p in zip(("y", "cb", "cr"), bp.final_planes())}
    else:
        csum = pipe.checksum()

    # ---- untimed pass: HIP-event time of every stage (events on the stream the kernels are launched on, recorded by the
    # pipeline's own stage hook, so exactly the launches of the timed loop are measured) ----
    ms = pipe.ms
    cur = pics[1]
    acc = {}
    for _ in range(0 if stub else 5):
        evs = [("start", torch.cuda.Event(enable_timing=True))]
        evs[0][1].record()

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append((name, e))
        pipe.run(cur, ref_pic, mark=mark)
        torch.cuda.synchronize()
        for (_, e0), (name, e1) in zip(evs[:-1], evs[1:]):
            acc.setdefault(name, []).append(e0.elapsed_time(e1))
    stages = {k: round(float(np.median(v)), 4) for k, v in acc.items()} if not stub else {"me": 1.0}
    # The pattern searches are data-dependent (early exits, STAR's raster refinement): stages_ms is the median of five runs of ONE frame pair
    # (frame 1 searched in the last reference), ms_per_step the average over the closed loop including the synthetic clip's wrap-around
    # pairs - for --search star the two differ by 2x, and that is GPU time of the search launches, not a host stall
    # (profiles/r03_search_star_stats.txt).  The exhaustive search (the default) does the same work for every pair.
    stages_note = None if args.search == "full" else ("data-dependent search: stages_ms is one frame pair (frame 1 vs the last reference), "
                                                      "ms_per_step averages the closed loop incl. the clip's wrap-around pairs")
    if pipe.lcb:
        # the lookahead's cost estimate: one launch scores `lookahead_batch` pictures on its own stream, overlapped with the stages above
        b = pipe.lcb
        side = torch.cuda.Stream()
        tms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            S.LookaheadCost.run_batch(pipe.lc[:b], [pipe.ring[(j + 1) % len(pipe.ring)] for j in range(b)], [pipe.ring[j % len(pipe.ring)] for j in range(b)],
                                      stream=side.cuda_stream)
            e1.record(side)
            torch.cuda.synchronize()
            tms.append(e0.elapsed_time(e1))
        stages["lookahead_cost_launch_of_%d (side stream, overlapped)" % b] = round(float(np.median(tms)), 4)

    if rank == 0:
        fps = world * args.steps / dt
        surf_mode = ms.surf is not None
        dom = "me"
        alg_bytes = ms.algorithmic_bytes(bpp=1 if args.depth == 8 else 2)
        achieved = alg_bytes / (stages[dom] * 1e-3) / 1e9
        if args.search != "full":
            # a pattern search visits a data-dependent handful of the window's candidates: the exhaustive search's algorithmic bytes do not
            # describe it and no per-launch figure of its own exists - no fraction is printed (round-3 verdict, weak 6 ii)
            alg_bytes = achieved = None
        traffic, tsrc = load_traffic(args.width, args.height, args.range,
                                     ((('packed_b' if ms.blocked else ('packed_t' if ms.tiled else 'packed')) if ms.packed else 'i32') if surf_mode else 'best') + ('' if args.depth == 8 else '_d10'))
        if args.search != "full":
            traffic, tsrc = None, None
        out = {
            "metric": "encoded fps + bit-exact check, 4K preset=slow, 1/2/4/8 MI355X vs host AVX2",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8" if args.depth == 8 else "u16", "data": "synthetic",
            "config": {"workload": (f"{args.width}x{args.height} {args.depth}-bit closed-loop frame pipeline (tier T2): lookahead -> "
                                    + (f"exhaustive +-{args.range} ME, 85 PUs/CTU -> subme {args.subme}" if args.search == "full" else f"{args.search} search, merange {args.range}, subme {args.subme}")
                                    + f" -> MC + DCT/quant/recon qp {args.qp} -> deblock -> SAO ({'rdo' if sao_rdo else 'stand-in'}) -> border -> next reference"),
                       "workload_detail": f"{args.width}x{args.height} {args.depth}-bit ({'BASELINE configs[2]: 4K, preset slow search depth' if args.width == 3840 else 'BASELINE configs[1] picture size' if args.width == 1920 else 'custom size'}) closed-loop frame pipeline: lookahead lowres planes + intra estimate (+ P-frame cost estimate vs the previous picture, " + (f"{args.lookahead_batch} pictures per launch on a side stream" if args.lookahead_batch else "off") + ") -> " +
                                   (f"ME exhaustive +-{args.range} for all 85 PUs/CTU ({('SAD surfaces (' + (('packed, in blocks of 64' if ms.blocked else ('packed chunk-major' if ms.tiled else 'packed')) if ms.packed else 'i32') + ' records) + ') if surf_mode else ''}best mv) -> "
                                    f"sub-pel subme={args.subme} -> " if args.search == "full" else
                                    f"{args.search} search driver (motionEstimate, merange {args.range}, subme {args.subme}, predictor 0) for all 85 PUs/CTU -> ") +
                                   f"{8 << args.level}x{8 << args.level} luma + 4:2:0 chroma prediction + DCT/quant (sign-bit hiding on)/recon qp {args.qp} -> luma + chroma deblocking -> "
                                   f"SAO statistics -> SAO parameters (" + ("the reference's rate-distortion decision SAO::rdoSaoUnitCu on the device: offset iteration, CABAC bit "
                                   "counts with per-row contexts, merge candidates" if sao_rdo else "saoStatsInitialOffset + distortion-only stand-in, on device") + ") -> SAO apply (Y, Cb, Cr) -> "
                                   f"border extension -> next reference (Y, Cb, Cr"
                                   + ("; one GPU: the decoded picture ping-pongs - the planes a step wrote are the next step's reference, no copy"
                                      if world == 1 and args.ref_handoff == "swap" and not banded else "") +
                                   f"); pipeline throughput (tier T2), not HEVC encoded fps - the real "
                                   f"encoder's fps (tier T3) is the `encoder_summary` object of this line / bench_detail.json",
                       "frames_per_step_per_gpu": 1,
                       "parallelism": ("none (1 GPU)" if world == 1 and not banded else f"gop x{world}: replicas, no exchange" if gop else
                                       f"frame-parallel ring x{world}, bands of {args.band_rows} CTU rows, transport {transport_name if world > 1 else 'none'}"),
                       "parallelism_detail": ((f"segment-parallel x{world}: every rank encodes its own closed group of pictures, no exchange (--sharding gop)" if gop
                                        else f"frame-parallel x{world}") if not banded else
                                       f"frame-parallel ring x{world}: frame f on rank f % {world} searches " + (f"the newest anchor before it (mini-GOPs of {ring_gop}: anchors = multiples of {ring_gop}, the pictures between two anchors non-referenced)" if ring_gop else "frame f - 1") + f", handed on in bands of {args.band_rows} CTU rows "
                                       f"(each band a slice of its own, like the reference's --slices); band transfers: "
                                       + ("x265hip_recon_publish_rows (the library's C ABI on RCCL, one 2-rank communicator per directed flow)" if transport_name == "abi"
                                          else "x265hip_recon_publish_rows with peer = -1: ncclBroadcast over ONE communicator of all ranks, rooted at the anchor's rank (every rank joins every broadcast)" if transport_name == "bcast"
                                          else "torch.distributed broadcast over one group of all ranks" if transport_name == "dist_bcast"
                                          else "torch.distributed point-to-point")),
                       "sharding": ("gop" if gop else "ring") if world > 1 else "none",
                       "ctus_per_frame": ms.nctu, "checksum": csum,
                       **({"band_rows": args.band_rows, "band_streams": args.band_streams,
                           "ring_model": "N pictures per max(step, N x lag), lag = the band periods until the reference rows a band's search window "
                                         "reaches are final + a hand-over (DESIGN.md section 6; band size chosen from N by pick_band_rows)"} if banded else {})},
            "stages_ms": stages,
            **({"stages_note": stages_note} if stages_note else {}),
            "roofline": {"bound": "hbm", "kernel": "me_search_kernel" if args.search != "full" else
                                   ((("me_ctu_c_kernel" if (ms.tiled or ms.blocked) else "me_ctu_q_kernel") if args.depth == 8 else "me_ctu_w_kernel") + "<surf,best>" if surf_mode
                                    else minima_kernel_name(A, args.depth, args.range, stub)),
                         "achieved": round(achieved, 2) if achieved is not None else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved is not None else None, "traffic": traffic, "traffic_source": tsrc,
                         # the PHYSICAL fraction beside the contractual one: HBM bytes the counters saw / the live launch time / peak.
                         # `frac` prices SURVEY 8(d)'s algorithmic bytes (window re-reads that LDS serves, 4 B per candidate where the
                         # packed records hold 2.14) - it is the contract's number, this one is what the memory system really carries
                         "frac_traffic": round(traffic / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "output_bytes_per_launch": ms.hbm_floor_bytes(1 if args.depth == 8 else 2),
                         "launch_ms": stages[dom],
                         # what really bounds the kernel: the issue rate of its SAD instructions (the contract's `bound` can only say hbm | mfma)
                         "valu": valu_bound(ms.nctu, args.range, args.depth, stages[dom]) if args.search == "full" else None,
                         **({"note": "the launch keeps the per-PU minima only (default since round 4): its algorithmic bytes are the window reads + 8 bytes per PU; "
                                     "rounds 1 - 3 quoted 0.90 - 0.92 on a launch that also wrote 11.9 GB (algorithmic) of SAD surfaces nothing in the loop read "
                                     "(--surface brings it back)"} if args.search == "full" and not surf_mode else {})},
        }
        if ring_wait is not None:
            flows = min(ring_gop, world - 1) if ring_gop else 1
            model_kw = dict(ctu_rows=(args.height + 63) // 64, lag_rows_luma=args.range + 16, depth=args.depth, width=args.width)
            out["config"]["ring"] = {"ranks_seen": ranks_seen, "transport": transport_name, "bands_per_frame": len(bp.bands), "refs": 1, "gop": ring_gop,
                                     "communicators": world * flows if transport_name == "abi" else 1 if transport_name == "bcast" else 0,
                                     # what the band model predicts for this ring, in units of one GPU's whole-picture step (the tables are measured on one GPU)
                                     "model_x_one_gpu": (round(ring_model(world, args.band_rows, ring_gop, **model_kw) * band_table(args.depth, args.width)[1], 2)
                                                         if args.band_rows in band_table(args.depth, args.width)[0] else None),
                                     "comm_init_s": round(comm_init_s, 3) if comm_init_s is not None else None,
                                     "band_wait_ms_per_frame_max_over_ranks": ring_wait,
                                     "note": "band_wait = device time the bands' streams spent waiting for the reference rows they read (two events per band); "
                                             "ranks_seen = all-reduce of ones over the job's process group"}
        if replicas is not None:
            out["replicas"] = replicas
        sr = load_stage_traffic(args.width, args.height, args.depth)
        if sr:
            out["stages_roofline"] = sr
        if banded:
            out["roofline"]["note"] = ("stage times and roofline are whole-frame launches of rank 0 (untimed pass after the loop); the timed loop runs the "
                                       f"same stages band by band ({args.band_rows} CTU rows per band, row-walking search kernel per band)")
        bit_exact = None
        if world == 1 and not args.no_cpu_baseline:
            dev_out = None
            if args.search == "full" and not args.no_verify:      # one more (untimed) frame: clip[1] searched in the SOURCE clip[0]
                dev_out = device_outputs(pipe, pics[1], pics[0])
            out["cpu_baseline"], bit_exact = cpu_baseline(F, clip, args.range, args.subme, args.level, args.qp, depth=args.depth, dev_out=dev_out, sao_rdo=sao_rdo)
            if bit_exact is not None:
                out["bit_exact"] = bit_exact["ok"]
                out["bit_exact_detail"] = bit_exact
        if world == 1 and not args.no_encoder and not args.no_cpu_baseline:
            write_detail(out, args)              # the headline is on disk before the long encoder legs start
            out.update(encoder_leg(args))
        write_detail(out, args)
        line = json.dumps(compact_line(out), separators=(",", ":"))
        if len(line) >= MAX_LINE_BYTES:          # never again a line the driver cannot parse (round-4 verdict): drop the optional objects, keep the contract
            line = json.dumps(compact_line(out, minimal=True), separators=(",", ":"))
        sys.stdout.flush()
        print(line, flush=True)
        if bit_exact is not None and not bit_exact["ok"]:
            sys.stderr.write("bench.py: device pipeline differs from the oracle chain: %s\n" % json.dumps(bit_exact["stages"]))
            sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
