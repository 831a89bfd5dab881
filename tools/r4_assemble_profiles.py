#!/usr/bin/env python3
"""gpurun_out/<tag>/ of tools/r4_profile.sh -> the committed evidence under profiles/:

  profiles/r04_bench_kernel_stats_<cfg>.txt, r04_bench_pmc_<cfg>.txt, r04_bench_mfma_<cfg>.txt     the rocprofv3 summaries as they came
  profiles/r04_inst_counters_<cfg>.txt                                                              SQ instruction counters of the search kernels
  profiles/r04_valu_rates_ubench.txt + profiles/valu_rates.json                                     issue cost of the SAD instructions ALONE
  profiles/stage_traffic.json   {"configs": {"<W>x<H>_d<depth>": {source, kernels: {name: fetch / write bytes, avg_us, GB/s, frac_of_8tb, mfma_busy_frac}}}}
  profiles/traffic.json         + the dominant kernel's HBM bytes per launch for every profiled configuration (bench.py's roofline.traffic)

  python tools/r4_assemble_profiles.py gpurun_out/r4p
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_to_traffic as PT          # noqa: E402

CLOCK_GHZ, SIMDS = 2.4, 1024
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"          # file-name prefix of the round the visit belongs to (python tools/r4_assemble_profiles.py gpurun_out/r5p r05)


def mfma_fracs(path):
    out = {}
    if not os.path.exists(path):
        return out
    for ln in open(path):
        m = re.match(r"^(.*?)\s+SQ_VALU_MFMA_BUSY_CYCLES\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", ln)
        if m and PT.short_kernel(m.group(1)):
            busy, dur_ns = float(m.group(3)), float(m.group(6))
            out[PT.short_kernel(m.group(1))] = round(busy / (dur_ns * CLOCK_GHZ * SIMDS), 5) if dur_ns else None
    return out


def one_config(src, key, fill_bytes):
    pmc = PT.parse_pmc(os.path.join(src, f"pmc_{key}.txt"))
    stats = PT.parse_stats(os.path.join(src, f"kernel_stats_{key}.txt"))
    mf = mfma_fracs(os.path.join(src, f"mfma_{key}.txt"))
    cal = 1.0
    f = pmc.get("fill_u64_kernel")
    if f and f.get("WRITE_SIZE"):
        cal = fill_bytes / (f["WRITE_SIZE"] * 1024.0)
    kernels = {}
    for k, e in sorted(pmc.items()):
        if k == "fill_u64_kernel":
            continue
        fetch = e.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
        write = e.get("WRITE_SIZE", 0.0) * 1024.0 * cal
        us = stats.get(k, sum(e["dur_ns"]) / len(e["dur_ns"]) / 1e3)
        # round 6: FETCH_SIZE calibrated per access pattern (profiles/r06_fetch_size_calibration.txt, tools/ubench/fetch_calibration.hip): the x2 holds for every CONTIGUOUS row a
        # wavefront reads (16 / 4 / 1 byte per lane, 32-byte tile rows: 128-byte requests tallied as 64); single 64-byte lines fetched one request each (4x4 tiles at arbitrary
        # positions) are counted at face value.  fetch_bytes keeps the x2 (exact for streaming kernels, an UPPER bound for kernels that gather tiles / halos),
        # fetch_bytes_min is the counter at face value (the lower bound; what a pure gather like subpel_refine_kernel moves)
        kernels[k] = {"fetch_bytes": int(fetch), "fetch_bytes_min": int(fetch / 2), "write_bytes": int(write), "avg_us": round(us, 3), "gbytes_per_s": round((fetch + write) / us / 1e3, 1),
                      "frac_of_8tb": round((fetch + write) / us / 1e3 / 8000.0, 4), "mfma_busy_frac": mf.get(k) if mf.get(k) else None}
    return {"source": f"profiles/{RND}_bench_pmc_{key}.txt + profiles/{RND}_bench_kernel_stats_{key}.txt + profiles/{RND}_bench_mfma_{key}.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / "
                      f"--pmc SQ_VALU_MFMA_BUSY_CYCLES (separate passes, --kernel-trace only), per-dispatch averages; KiB -> bytes, FETCH_SIZE x 2 (gfx950), WRITE_SIZE x {cal:.4f} "
                      f"(calibrated on fill_u64_kernel: {fill_bytes} B written); mfma_busy_frac = busy cycles / (duration x {CLOCK_GHZ} GHz x {SIMDS} SIMDs)",
            "write_calibration": round(cal, 5), "kernels": kernels}


def main():
    src = sys.argv[1]
    prof = os.path.join(ROOT, "profiles")
    configs = {"3840x2160_d8": (3840, 2160, 8, 2040), "3840x2160_d10": (3840, 2160, 10, 2040), "7680x4320_d10": (7680, 4320, 10, 8160), "3840x2160_d8_surface": (3840, 2160, 8, 2040)}
    st_path = os.path.join(prof, "stage_traffic.json")
    st = json.load(open(st_path)) if os.path.exists(st_path) else {"configs": {}}          # configurations this visit did not re-profile keep their entries
    st.setdefault("configs", {})
    tr_path = os.path.join(prof, "traffic.json")
    tr = json.load(open(tr_path)) if os.path.exists(tr_path) else {}
    for key, (w, h, d, nctu) in configs.items():
        if not os.path.exists(os.path.join(src, f"pmc_{key}.txt")):
            continue
        for kind in ("kernel_stats", "pmc", "mfma"):
            p = os.path.join(src, f"{kind}_{key}.txt")
            if os.path.exists(p):
                shutil.copy(p, os.path.join(prof, f"{RND}_bench_{kind}_{key}.txt"))
        p = os.path.join(src, f"bench_{key}.json")
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(prof, f"{RND}_bench_under_rocprof_{key}.json"))
        c = one_config(src, key, nctu * 85 * 8)
        if not key.endswith("_surface"):
            st["configs"][key] = c
        # the dominant kernel's traffic per launch -> traffic.json (the keys bench.py's load_traffic builds)
        for k, v in c["kernels"].items():
            if k.startswith("me_ctu"):
                surf = "true" in k.split("<")[1].split(",")[0]
                fmt = ("packed_b" if surf else "best") + ("" if d == 8 else "_d10")
                if surf and d != 8:
                    fmt = "i32_d10"
                tr[f"me_{fmt}_{w}x{h}_r57"] = {"fetch_bytes": v["fetch_bytes"], "write_bytes": v["write_bytes"], "kernel": k, "avg_us": v["avg_us"],
                                               "source": f"profiles/{RND}_bench_pmc_{key}.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), "
                                                         f"per-dispatch average of {k}, KiB -> bytes; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE "
                                                         f"calibrated on fill_u64_kernel in the same pass"}
    json.dump(st, open(os.path.join(prof, "stage_traffic.json"), "w"), indent=1)
    json.dump(tr, open(tr_path, "w"), indent=1)
    for key in ("3840x2160_d10", "3840x2160_d8"):
        p = os.path.join(src, f"inst_{key}.txt")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(prof, f"{RND}_inst_counters_{key}.txt"))
    fc = os.path.join(src, "fetch_size_calibration.txt")
    if os.path.exists(fc):
        shutil.copy(fc, os.path.join(prof, f"{RND}_fetch_size_calibration.txt"))
    vr = os.path.join(src, "valu_rates.txt")
    if os.path.exists(vr):
        shutil.copy(vr, os.path.join(prof, f"{RND}_valu_rates_ubench.txt"))
        ns = {}
        for ln in open(vr):
            m = re.match(r"^(\S.*?)\s+threads/WG\s+1024:.*->\s+([\d.]+) ns/instr/SIMD-slot", ln)
            if m:
                ns[m.group(1).strip()] = float(m.group(2))
        if "v_qsad_pk_u16_u8 alone" in ns:
            json.dump({"v_qsad_pk_u16_u8_ns": ns["v_qsad_pk_u16_u8 alone"], "v_sad_u16_ns": ns["v_sad_u16 alone"], "v_sad_u8_ns": ns.get("v_sad_u8 alone"),
                       "plain_valu_ns": ns.get("v_add+xor (2 ops)", 0) / 2 or None,
                       "source": f"profiles/{RND}_valu_rates_ubench.txt (tools/ubench/valu_rates.hip, 4 wavefronts per SIMD, 8 independent chains): wall ns per wave-instruction per SIMD "
                                 "of the instruction ALONE (OPs 17 - 19; rounds 1 - 3 quoted OPs 0 - 2, whose iterations also issue a v_or_b32 per 32-bit operand)"},
                      open(os.path.join(prof, "valu_rates.json"), "w"), indent=1)
    print(json.dumps({k: {n: (v["avg_us"], v["frac_of_8tb"], v["mfma_busy_frac"]) for n, v in c["kernels"].items()} for k, c in st["configs"].items()}, indent=0)[:6000])


if __name__ == "__main__":
    main()
