#!/usr/bin/env python3
"""Randomised soak of the loop-filter stages on the GPU against the oracle: the SAO passes (statistics, parameters, application, luma and
chroma footprints, fused launches) and the deblocking tests of tests/test_gpu_sao.py / test_gpu_deblock.py with random picture sizes,
bit depths, block sizes, QPs and offsets.  python tools/filter_soak.py [seconds] [seed]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_deblock as D
import test_gpu_sao as S


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0, done, short = time.time(), {}, 0
    while time.time() - t0 < budget:
        depth = int(rng.choice([8, 8, 10, 12]))
        w, h = 2 * int(rng.integers(33, 260)), 2 * int(rng.integers(33, 180))
        w8, h8 = 8 * int(rng.integers(9, 60)), 8 * int(rng.integers(9, 40))
        cases = [("sao passes", lambda: S.test_sao_passes_match_oracle(depth, w, h)),
                 ("sao chroma planes", lambda: S.test_sao_chroma_planes_match_oracle(depth, w8, h8)),
                 ("sao decide", lambda: S.test_sao_decide_matches_oracle_and_closes_the_loop(depth, w, h)),
                 ("sao fused planes", lambda: S.test_sao_planes_fused_equals_the_single_plane_entries(depth, w8 * 2, h8 * 2)),
                 ("deblock random strengths", lambda: D.test_deblock_random_strengths_qp_map_and_offsets(int(rng.choice([8, 10])))),
                 ("deblock intra + chroma", lambda: D.test_deblock_with_intra_blocks_and_chroma(depth, int(rng.integers(0, 3)), min(51, int(rng.integers(28, 44)) + 2 * (depth - 8)),
                                                                                             (int(rng.integers(-6, 7)), int(rng.integers(-6, 7))))),
                 ("boundary strengths", lambda: D.test_boundary_strengths_multi_reference_and_b_pictures(int(rng.integers(0, 4)), int(rng.integers(0, 2)),
                                                                                                        str(rng.choice(["both", "noref1", "ref0only", "none"])), seed=int(rng.integers(1, 1 << 30))))]
        for name, fn in cases:
            try:
                fn()
                done[name] = done.get(name, 0) + 1
            except AssertionError as e:
                line = traceback.extract_tb(e.__traceback__)[-1].line or ""
                if "array_equal" not in line and " == " not in line and ("> 0" in line or ".any()" in line or "!=" in line):
                    short += 1          # a coverage assertion of the test (the random case did not exercise something)
                    continue
                print(f"MISMATCH in {name}: depth {depth} {w}x{h} / {w8}x{h8}: {line.strip()[:160]} {str(e)[:300]}", flush=True)
                sys.exit(1)
    print("filter soak: " + ", ".join(f"{k} {v}" for k, v in done.items()) + f" random cases equal to the oracle ({short} missed a test's own coverage check; {time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
