#!/usr/bin/env python3
"""Latency / concurrency probe of x265hip_lowres_cost_host (the lookahead seam's provider): one frame cost estimate per call from 1, 4 and
16 host threads at 1080p and 4K source sizes, B pictures, planes through the shared device cache.  Prints ms per call and estimates/s."""
import ctypes
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
import oracle_api as O


class HP(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_int), ("stride", ctypes.c_ssize_t), ("width_in_cu", ctypes.c_int), ("height_in_cu", ctypes.c_int),
                ("lines", ctypes.c_int), ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int), ("cur", ctypes.c_void_p),
                ("ref", ctypes.c_void_p * 4), ("ref1", ctypes.c_void_p * 4), ("ref_bi", ctypes.c_void_p * 4),
                ("intra_cost", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p), ("cost_q", ctypes.c_void_p), ("cost_q_half", ctypes.c_int),
                ("bframe_bias", ctypes.c_int), ("do_search", ctypes.c_int * 2), ("mvs", ctypes.c_void_p * 2), ("mv_costs", ctypes.c_void_p * 2),
                ("lowres_costs", ctypes.c_void_p), ("row_satds", ctypes.c_void_p), ("frame", ctypes.c_void_p),
                ("plane_key_cur", ctypes.c_uint64), ("plane_key_ref", ctypes.c_uint64), ("plane_key_ref1", ctypes.c_uint64), ("plane_key_ref_bi", ctypes.c_uint64)]


def main():
    f = A.lib().x265hip_lowres_cost_host
    f.argtypes = [ctypes.POINTER(HP)]
    for (W, Hh) in ((1920, 1080), (3840, 2160)):
        clip = F.synth_clip(W, Hh, 3, depth=8, seed=5)
        lw, lh = ((W // 2 + 7) >> 3) * 8, ((Hh // 2 + 7) >> 3) * 8
        stride = (lw + 2 * F.MARGIN_X + 31) & ~31
        org = stride * F.MARGIN_Y + F.MARGIN_X
        rows = lh + 2 * F.MARGIN_Y
        planes = []
        for y, _, _ in clip:
            buf, st, og, _, _ = F.pad_plane(y)
            planes.append(O.lowres_init(8, buf, st, og, stride, org, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y))
        wcu, hcu = lw // 8, lh // 8
        n = wcu * hcu
        icost, _, _ = O.lowres_intra(8, planes[1][0], stride, org, wcu, hcu, 5, nthreads=8)
        half = 4 * max(lw, lh) + 1024
        cq, qoff = F.qpel_cost_table(16, lam=1.0, qmax=half)

        def make(bidir):
            q = HP()
            keep = [np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint16),
                    np.zeros(hcu, np.int32), np.zeros(4, np.int64)]
            q.depth, q.stride, q.width_in_cu, q.height_in_cu, q.lines, q.margin_x, q.margin_y = 8, stride, wcu, hcu, lh, F.MARGIN_X, F.MARGIN_Y
            q.cur = planes[1][0].ctypes.data + org
            for i in range(4):
                q.ref[i] = planes[0][i].ctypes.data + org
                q.ref1[i] = planes[2][i].ctypes.data + org if bidir else None
            q.intra_cost = icost.ctypes.data
            q.cost_q, q.cost_q_half = cq.ctypes.data + 2 * qoff, half
            q.do_search[0], q.do_search[1] = 1, int(bidir)
            q.mvs[0], q.mvs[1], q.mv_costs[0], q.mv_costs[1] = keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data
            q.lowres_costs, q.row_satds, q.frame = keep[4].ctypes.data, keep[5].ctypes.data, keep[6].ctypes.data
            q.plane_key_cur, q.plane_key_ref, q.plane_key_ref1 = 2, 1, 3
            return q, keep
        for bidir in (False, True):
            for nthreads in (1, 4, 16):
                reps = 6
                qs = [make(bidir) for _ in range(nthreads)]
                A.check(f(ctypes.byref(qs[0][0])), "warm")          # planes into the cache, kernels loaded

                def work(q):
                    for _ in range(reps):
                        A.check(f(ctypes.byref(q)), "x265hip_lowres_cost_host")
                ts = [threading.Thread(target=work, args=(q,)) for q, _ in qs]
                t0 = time.perf_counter()
                for t in ts: t.start()
                for t in ts: t.join()
                dt = time.perf_counter() - t0
                print(f"{W}x{Hh} {'B' if bidir else 'P'} picture, {nthreads:2d} host threads: {1e3 * dt / reps:7.2f} ms per call per thread, "
                      f"{nthreads * reps / dt:8.1f} estimates/s", flush=True)
        t0 = time.perf_counter()
        O.lowres_cost(8, planes[1][0], planes[0], stride, org, wcu, hcu, cq, qoff, icost)
        print(f"{W}x{Hh} P picture, oracle C on one CPU thread: {1e3 * (time.perf_counter() - t0):.1f} ms per estimate", flush=True)


if __name__ == "__main__":
    main()
