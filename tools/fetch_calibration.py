#!/usr/bin/env python3
"""FETCH_SIZE per access pattern against known byte counts (round-5 verdict, next 4c).  On the GPU box:

    python tools/fetch_calibration.py gpurun_out/<tag>

builds tools/ubench/fetch_calibration, runs it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (counters only, as the pool requires) and prints, per
pattern, requested bytes, the bytes of the distinct 64-byte / 128-byte lines touched, FETCH_SIZE x 1024 and the factor that turns the counter into each of
them.  The table goes to profiles/r06_fetch_size_calibration.txt; tools/r4_assemble_profiles.py's x2 is the `cal_stream_b128` row."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fetchcal"))
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(ROOT, "tools", "ubench", "fetch_calibration")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", exe + ".hip", "-o", exe])
    env = dict(os.environ, TMPDIR="/tmp")
    known = subprocess.run(["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", os.path.join(out, "pmc"), "-o", "f", "--", exe], cwd="/tmp", env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    rows = {ln.split()[0]: [int(v) for v in ln.split()[1:4]] for ln in known.stdout.splitlines() if ln.startswith("cal_")}
    db = subprocess.check_output(["find", os.path.join(out, "pmc"), "-name", "*.db"], text=True).split()[0]
    summ = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), "pmc", db], text=True)
    lines = ["# FETCH_SIZE (rocprofv3 --pmc FETCH_SIZE --kernel-trace, gfx950, ROCm 7.2) against KNOWN bytes per access pattern: tools/ubench/fetch_calibration.hip over a 1 GiB",
             "# buffer (4 x the Infinity Cache), 3 launches each, per-dispatch average.  factor_* = known bytes / (FETCH_SIZE x 1024): what the counter must be multiplied by.",
             f"{'pattern':18s} {'requested_B':>14s} {'lines64_B':>14s} {'lines128_B':>14s} {'FETCH_SIZE_B':>14s} {'factor_req':>10s} {'factor_64':>10s} {'factor_128':>10s} {'avg_us':>9s} {'GB/s_req':>9s}"]
    for ln in summ.splitlines():
        m = re.match(r"^(?:void\s+)?(cal_\w+)\(.*?\s+FETCH_SIZE\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", ln)
        if not m or m.group(1) not in rows:
            continue
        req, l64, l128 = rows[m.group(1)]
        fetch, dur = float(m.group(3)) * 1024.0, float(m.group(6))
        lines.append(f"{m.group(1):18s} {req:14d} {l64:14d} {l128:14d} {fetch:14.0f} {req / fetch:10.3f} {l64 / fetch:10.3f} {l128 / fetch:10.3f} {dur / 1e3:9.1f} {req / dur:9.1f}")
    text = "\n".join(lines)
    print(text)
    open(os.path.join(out, "fetch_size_calibration.txt"), "w").write(text + "\n")
    subprocess.call(["find", out, "-name", "*.db", "-delete"])


if __name__ == "__main__":
    main()
