"""Random soak of x265hip_me_fullsearch (all three record formats, both 8-bit kernels, windows of 1 .. 70) against the oracle: picture
sizes, ranges and formats drawn at random with a fixed seed.  Test infrastructure (uses tests/test_gpu_me.py's checker)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_me as T
    n = int(os.environ.get("N", 60))
    rng = np.random.default_rng(int(os.environ.get("SEED", 2026)))
    done = {}
    for it in range(n):
        w, h = 64 * int(rng.integers(1, 5)), 64 * int(rng.integers(1, 4))
        r = int(rng.integers(1, 71))
        fmt = [False, True, "t"][int(rng.integers(0, 3))]
        kernel = ["", "cand", "rows"][int(rng.integers(0, 3))] if fmt != "t" else ""
        if kernel:
            os.environ["X265HIP_ME_KERNEL"] = kernel
        else:
            os.environ.pop("X265HIP_ME_KERNEL", None)
        T.A.lib().x265hip_me_env_refresh()          # the library reads its switches once per process; this soak flips one per iteration
        extreme = "flat" if rng.integers(0, 8) == 0 else None
        try:
            T._run(w, h, r, 8, seed=int(rng.integers(1, 1000)), extreme=extreme, packed=fmt)
            key = (str(fmt), kernel or "auto")
            done[key] = done.get(key, 0) + 1
        except Exception as e:                      # a window the chosen format / kernel cannot serve must fail loudly, not wrongly
            msg = str(e)
            if "X265HIP_SURF_PACKED" in msg or "LDS" in msg or "needs" in msg:
                done[("refused", str(fmt))] = done.get(("refused", str(fmt)), 0) + 1
                continue
            print(f"FAIL it {it}: {w}x{h} R={r} fmt={fmt} kernel={kernel} extreme={extreme}: {msg[:300]}", flush=True)
            raise
    print("soak ok:", {str(k): v for k, v in sorted(done.items(), key=str)})


if __name__ == "__main__":
    main()
