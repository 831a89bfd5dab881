#!/usr/bin/env python3
"""profiles/<round>_pmc.txt (+ the kernel-trace summary of the same command) -> profiles/stage_traffic.json: for every x265hip kernel of
the default bench.py step the HBM bytes one launch moves (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, corrected as
/opt/skills/guides/MI355X_MICROARCH.md's HBM section prescribes: KiB -> bytes, FETCH_SIZE doubled on gfx950, WRITE_SIZE calibrated on
fill_u64_kernel whose written bytes are known) and its average duration.  bench.py prints them as `stages_roofline` so the weakest
kernel is in the bench line, not in prose (round-2 verdict, next 3).

  python tools/pmc_to_traffic.py profiles/r03_bench_pmc.txt profiles/r03_bench_kernel_stats.txt [--fill-bytes 1387200] > profiles/stage_traffic.json
"""
import json
import re
import sys


def short_kernel(name):
    m = re.search(r"x265hip::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def parse_pmc(path):
    out = {}
    for ln in open(path):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", ln)
        if not m:
            continue
        k = short_kernel(m.group(1))
        if k:
            e = out.setdefault(k, {})
            e[m.group(2)] = float(m.group(4))
            e["n"] = int(m.group(3))
            e.setdefault("dur_ns", []).append(float(m.group(7)))
    return out


def parse_stats(path):
    out = {}
    for ln in open(path):
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m and short_kernel(m.group(1)):
            out[short_kernel(m.group(1))] = float(m.group(4))       # avg_us
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    fill_bytes = 1387200           # bench.py's 4K run: fill_u64_kernel writes 2040 CTUs x 85 PUs x 8 B
    if "--fill-bytes" in sys.argv:
        fill_bytes = int(sys.argv[sys.argv.index("--fill-bytes") + 1])
    pmc = parse_pmc(args[0])
    stats = parse_stats(args[1]) if len(args) > 1 else {}
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
        kernels[k] = {"fetch_bytes": int(fetch), "write_bytes": int(write), "avg_us": round(us, 3),
                      "launches_per_step": None, "gbytes_per_s": round((fetch + write) / us / 1e3, 1), "frac_of_8tb": round((fetch + write) / us / 1e3 / 8000.0, 4)}
    print(json.dumps({"source": f"{args[0]}" + (f" + {args[1]}" if len(args) > 1 else "") +
                                ": rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), per-dispatch averages; KiB -> bytes, "
                                f"FETCH_SIZE x 2 (gfx950), WRITE_SIZE x {cal:.4f} (calibrated on fill_u64_kernel: {fill_bytes} B written)",
                      "write_calibration": round(cal, 5), "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
