#!/usr/bin/env python3
"""Repeats the row-by-row scenario of tests/test_gpu_seam.py::test_phase_stream_planes_equal_the_picture_granular_planes and, on a mismatch,
says WHERE the planes differ (plane, phases, lines, columns) - the test only says that they do.  GPU box only."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from tools import seam_driver as SD
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    depth, width, height = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    reps, sleep = int(sys.argv[4]), float(sys.argv[5])
    geo = SD.geometry(width, height)
    dt = np.uint8 if depth == 8 else np.uint16
    bad = 0
    for rep in range(reps):
        rng = np.random.default_rng(3 + depth + rep)
        src = [rng.integers(0, 1 << depth, (geo["rows"], geo["stride"])).astype(dt), rng.integers(0, 1 << depth, (geo["rows_c"], geo["stride_c"])).astype(dt),
               rng.integers(0, 1 << depth, (geo["rows_c"], geo["stride_c"])).astype(dt)]
        prov = SD.StreamGpuPhaseProvider(depth, geo, slots=2)
        L = prov.L
        L.x265hip_phase_stream_open.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.x265hip_phase_stream_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.x265hip_phase_stream_planes.restype = ctypes.c_void_p
        L.x265hip_phase_stream_planes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.x265hip_phase_stream_progress.restype = ctypes.c_void_p
        L.x265hip_phase_stream_progress.argtypes = [ctypes.c_void_p, ctypes.c_int]
        ctu_rows = geo["height"] // 64
        for slot in (1, 0):
            gen = L.x265hip_phase_stream_open(prov.handle, slot)
            prog = np.ctypeslib.as_array((ctypes.c_uint64 * 2).from_address(L.x265hip_phase_stream_progress(prov.handle, slot)))
            for r in range(ctu_rows):
                assert L.x265hip_phase_stream_rows(prov.handle, slot, gen, src[0].ctypes.data, src[1].ctypes.data, src[2].ctypes.data, r, 1) == 0
                if sleep:
                    time.sleep(sleep)
            want = [(gen << 32) | (geo["rows"] - 8), (gen << 32) | (geo["rows_c"] - 8)]
            t0 = time.time()
            while (int(prog[0]) != want[0] or int(prog[1]) != want[1]) and time.time() - t0 < 20:
                time.sleep(0.005)
            assert [int(prog[0]), int(prog[1])] == want, prov.report()
            for pl in range(3):
                k = min(pl, 1)
                rows, stride, nph = (geo["rows"], geo["stride"], 15) if k == 0 else (geo["rows_c"], geo["stride_c"], 63)
                n = nph * rows * stride
                got = np.ctypeslib.as_array((ctypes.c_uint8 * (n * dt().itemsize)).from_address(L.x265hip_phase_stream_planes(prov.handle, slot, pl))).view(dt).reshape(nph, rows, stride).copy()
                dev = torch.device("cuda:0")
                tsrc = torch.from_numpy(src[pl].view(np.int16 if depth > 8 else np.uint8)).to(dev)
                tdst = torch.zeros(n, dtype=tsrc.dtype, device=dev)
                A.phase_planes(depth, tsrc, 0, tdst, stride, rows, chroma=bool(k))
                torch.cuda.synchronize()
                exp = tdst.cpu().numpy().view(dt).reshape(nph, rows, stride)
                g, e = got[:, 4:rows - 8, 8:stride - 8], exp[:, 4:rows - 8, 8:stride - 8]
                if not np.array_equal(g, e):
                    bad += 1
                    d = np.argwhere(g != e)
                    ph, ln, col = d[:, 0], d[:, 1] + 4, d[:, 2] + 8
                    print(f"rep {rep} slot {slot} plane {pl}: {len(d)} samples differ; phases {sorted(set(ph.tolist()))[:20]} lines {ln.min()}..{ln.max()} "
                          f"(distinct {len(set(ln.tolist()))}) columns {col.min()}..{col.max()}; first got {g[tuple(d[0])]} expected {e[tuple(d[0])]}; "
                          f"zeros among got {int((g[g != e] == 0).sum())}", flush=True)
        prov.close()
    print(f"{reps} repetitions x 2 slots x 3 planes: {bad} mismatching planes")


if __name__ == "__main__":
    main()
