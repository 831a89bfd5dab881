"""Repeats a small encode with the lookahead seam on the GPU provider and the oracle re-scoring every estimate (gpu+verify): hunts
intermittent differences between x265hip_lowres_cost_host / _intra_host and the CPU restatement.  Test infrastructure."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_seam_cpu as T
    n = int(os.environ.get("N", 20))
    bad = 0
    for it in range(n):
        for depth, preset, extra in ((8, "medium", [("bframes", "0")]), (8, "medium", []), (8, "slow", [])):
            opts = [("pools", "4"), ("frame-threads", "1"), ("crf", "24"), ("lookahead-slices", "1")] + extra
            base, got, rep = T.run_pair(depth, 320, 192, 12, preset, opts, "gpu", rng=16, verify=False, wait=True, lookahead="gpu+verify", seed=41 + (it % 3 if os.environ.get("VARY") else 0))
            la = rep["lookahead_seam"]
            ok = got[0] == base[0] and la["verify_mismatches"] == 0
            bad += not ok
            print(f"iter {it} {preset} {extra}: md5 equal {got[0] == base[0]}, estimates {la['frame_cost_estimates_served']}, mismatches {la['verify_mismatches']}", flush=True)
    print("BAD RUNS:", bad)


if __name__ == "__main__":
    main()
