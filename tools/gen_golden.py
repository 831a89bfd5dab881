#!/usr/bin/env python3
"""Generate tests/golden/prims_d{8,10}.npz from the REAL reference build (oracle/_ref, i.e. x265 3.5's C
primitives compiled from /root/reference by oracle/Makefile).  Each entry is data only: explicit input
arrays and the reference's outputs for one primitive call.  Needs this container (/root/reference); the
fixtures themselves are committed so every later run - CPU oracle and GPU kernels - can be checked
against reference results without the reference being present.

Naming: <slot path>|<case>|in<k> / out<k> / ret, args stored as <...>|args (int64 vector).
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H                      # noqa: E402
import golden_cases as G                 # noqa: E402


def main():
    for depth in (8, 10):
        ref = H.load_reference(depth, ROOT)
        if ref is None:
            raise SystemExit("oracle/_ref missing: run `make -C oracle ref` first (needs /root/reference)")
        store = {}
        for case in G.cases(depth):
            G.run_case(ref, case, store, record=True)
        path = os.path.join(ROOT, "tests", "golden", f"prims_d{depth}.npz")
        np.savez_compressed(path, **store)
        print(path, len(store), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
