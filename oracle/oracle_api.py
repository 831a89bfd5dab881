"""ctypes access to the oracle (oracle/_build/libx265oracle*.so) - TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does (tests/test_no_oracle_in_product.py enforces it).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


TU_INTRA_SLICE, TU_SIGN_HIDE = 1, 2       # flag bits of the TU stages' `intra_slice` argument (x265hip.h: X265HIP_TU_*)


def set_tu_tables(depth, quant_coeff=None, dequant_coeff=None, nr_offset=None, nr_sum=None, avx2=False):
    """Scaling-list coefficients / denoiser tables for the oracle's TU stages (numpy arrays or None; kept alive by the caller).  Call with
    no tables to switch back to flat lists."""
    fn = getattr(lib(avx2), f"x265oracle_set_tu_tables_d{depth}")
    fn.argtypes = [ctypes.c_void_p] * 4
    fn(*[None if a is None else a.ctypes.data for a in (quant_coeff, dequant_coeff, nr_offset, nr_sum)])


def set_tu_capture(depth, dct_coeff=None, delta_u=None, avx2=False):
    """Capture buffers of the oracle's TU stages (int16 / int32 numpy arrays shaped like the levels, or None): the coefficients handed
    to the quantiser and the quantiser's deltaU."""
    fn = getattr(lib(avx2), f"x265oracle_set_tu_capture_d{depth}")
    fn.argtypes = [ctypes.c_void_p] * 2
    fn(*[None if a is None else a.ctypes.data for a in (dct_coeff, delta_u)])


def host_has_avx2() -> bool:
    try:
        return " avx2 " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


def lib(avx2: bool = False) -> ctypes.CDLL:
    key = bool(avx2)
    if key not in _libs:
        name = "libx265oracle_avx2.so" if avx2 else "libx265oracle.so"
        path = os.path.join(_HERE, "_build", name)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle oracle` (or __graft_entry__.build())")
        _libs[key] = ctypes.CDLL(path)
    return _libs[key]


def me_fullsearch(depth, fenc, fenc_stride, fenc_org, fref, fref_stride, fref_org, width, height, rng,
                  ctu_begin, ctu_end, cost_x, cost_y, want_surf=True, want_best=True, levels=(0, 1, 2, 3),
                  nthreads=0, avx2=False):
    """Run the CPU restatement of the exhaustive search on padded host planes (numpy).
    Returns (surf, best) with the same layouts as the HIP ABI (int32 [ctu][mvy][mvx/4][85][4] and uint64
    [ctu][85]; full-frame sized arrays, only CTUs [ctu_begin, ctu_end) are filled)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_me_fullsearch_d{depth}")
    nctu = (width // 64) * (height // 64)
    nc = 2 * rng + 1
    ng = (nc + 3) // 4
    surf = np.zeros(nctu * nc * ng * 340, dtype=np.int32) if want_surf else None
    best = np.full(nctu * 85, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64) if want_best else None
    mask = sum(1 << l for l in levels)
    es = fenc.itemsize
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    cx = np.ascontiguousarray(cost_x, dtype=np.uint16)
    cy = np.ascontiguousarray(cost_y, dtype=np.uint16)
    fn(fenc.ctypes.data + fenc_org * es, fenc_stride, fref.ctypes.data + fref_org * es, fref_stride,
       width, height, rng, ctu_begin, ctu_end,
       surf.ctypes.data if surf is not None else None, best.ctypes.data if best is not None else None,
       cx.ctypes.data, cy.ctypes.data, mask, nthreads)
    return surf, best


def subpel_refine(depth, fenc, fenc_stride, fenc_org, fref, fref_stride, fref_org, width, height, rng,
                  ctu_begin, ctu_end, best_in, cost_q, qoff, subme, nthreads=0, avx2=False):
    """CPU restatement of the sub-pel refinement stage; returns int32 [nctu*85, 2] = {cost, qx | qy << 16}."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_subpel_refine_d{depth}")
    nctu = (width // 64) * (height // 64)
    out = np.zeros((nctu * 85, 2), dtype=np.int32)
    es = fenc.itemsize
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    b = np.ascontiguousarray(best_in, dtype=np.uint64)
    cq = np.ascontiguousarray(cost_q, dtype=np.uint16)
    fn(fenc.ctypes.data + fenc_org * es, fenc_stride, fref.ctypes.data + fref_org * es, fref_stride,
       width, height, rng, ctu_begin, ctu_end, b.ctypes.data, cq.ctypes.data, qoff, subme, out.ctypes.data, nthreads)
    return out


def inter_recon(depth, fenc, fenc_stride, fenc_org, fref, fref_stride, fref_org, width, height, level, mv, qp,
                intra_slice=0, ctu_begin=0, ctu_end=None, nthreads=0, avx2=False):
    """CPU restatement of the fused prediction + residual round trip.  Returns (recon plane, levels, num_sig, dist)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_inter_recon_d{depth}")
    nctu = (width // 64) * (height // 64)
    if ctu_end is None:
        ctu_end = nctu
    n = 8 << level
    nblk = (64 // n) ** 2
    recon = np.zeros_like(fenc)
    levels = np.zeros(nctu * nblk * n * n, dtype=np.int16)
    num_sig = np.zeros(nctu * nblk, dtype=np.uint32)
    dist = np.zeros(nctu * nblk, dtype=np.uint64)
    es = fenc.itemsize
    m = np.ascontiguousarray(mv, dtype=np.int32)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    fn(fenc.ctypes.data + fenc_org * es, fenc_stride, fref.ctypes.data + fref_org * es, fref_stride,
       recon.ctypes.data + fenc_org * es, fenc_stride, width, height, level, m.ctypes.data, qp, intra_slice,
       levels.ctypes.data, num_sig.ctypes.data, dist.ctypes.data, ctu_begin, ctu_end, nthreads)
    return recon, levels, num_sig, dist


def lowres_init(depth, src, src_stride, src_org, stride, org, rows, width, lines, margin_x, margin_y, avx2=False):
    """CPU restatement of Lowres::init's pixel work.  Returns the four padded lowres planes (flat arrays of rows * stride)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_lowres_init_d{depth}")
    planes = [np.zeros(rows * stride, dtype=src.dtype) for _ in range(4)]
    es = src.itemsize
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_void_p] * 4 + [ctypes.c_ssize_t] + [ctypes.c_int] * 4
    fn(src.ctypes.data + src_org * es, src_stride, *[p.ctypes.data + org * es for p in planes], stride, width, lines, margin_x, margin_y)
    return planes


def lowres_intra(depth, plane, stride, org, width_in_cu, height_in_cu, intra_penalty, nthreads=0, avx2=False):
    """CPU restatement of LookaheadTLD::lowresIntraEstimate's per-block work.  Returns (intra_cost, intra_mode, lowres_costs)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_lowres_intra_d{depth}")
    n = width_in_cu * height_in_cu
    cost, mode, lc = np.zeros(n, np.int32), np.zeros(n, np.uint8), np.zeros(n, np.uint16)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_int]
    fn(plane.ctypes.data + org * plane.itemsize, stride, width_in_cu, height_in_cu, intra_penalty, cost.ctypes.data, mode.ctypes.data,
       lc.ctypes.data, nthreads)
    return cost, mode, lc


def cutree_propagate(depth, width_in_cu, height_in_cu, propagate_in, intra_cost, lowres_costs, inv_qscale, mvs0, mvs1, fps_factor, bipred_weight,
                     ref_cost0, ref_cost1=None, avx2=False):
    """CPU restatement of Lookahead::estimateCUPropagate + primitives.propagateCost.  Returns the updated copies (ref_cost0, ref_cost1)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_cutree_propagate_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    pi = None if propagate_in is None else np.ascontiguousarray(propagate_in, np.uint16)
    ic, lc, iq = np.ascontiguousarray(intra_cost, np.int32), np.ascontiguousarray(lowres_costs, np.uint16), np.ascontiguousarray(inv_qscale, np.int32)
    m0 = np.ascontiguousarray(mvs0, np.int32)
    m1 = None if mvs1 is None else np.ascontiguousarray(mvs1, np.int32)
    r0 = np.ascontiguousarray(ref_cost0, np.uint16).copy()
    r1 = None if ref_cost1 is None else np.ascontiguousarray(ref_cost1, np.uint16).copy()
    ptr = lambda a: None if a is None else a.ctypes.data
    fn(width_in_cu, height_in_cu, ptr(pi), ic.ctypes.data, lc.ctypes.data, iq.ctypes.data, m0.ctypes.data, ptr(m1), float(fps_factor), int(bipred_weight),
       r0.ctypes.data, ptr(r1))
    return r0, r1


def cutree_finish(depth, intra_cost, inv_qscale, propagate_cost, qp_aq_offset, fps_factor_q8, weight_delta, strength, qp_cutree_offset, avx2=False):
    """CPU restatement of Lookahead::cuTreeFinish (qgSize >= 16, hevcAq off).  Returns the updated copy of qp_cutree_offset."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_cutree_finish_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    ic, iq = np.ascontiguousarray(intra_cost, np.int32), np.ascontiguousarray(inv_qscale, np.int32)
    pc, qa = np.ascontiguousarray(propagate_cost, np.uint16), np.ascontiguousarray(qp_aq_offset, np.float64)
    out = np.ascontiguousarray(qp_cutree_offset, np.float64).copy()
    fn(len(ic), ic.ctypes.data, iq.ctypes.data, pc.ctypes.data, qa.ctypes.data, int(fps_factor_q8), float(weight_delta), float(strength), out.ctypes.data)
    return out


def frame_cost_recalculate(depth, width_in_cu, height_in_cu, lowres_costs, qp_cutree_offset, avx2=False):
    """CPU restatement of Lookahead::frameCostRecalculate (P pictures).  Returns (score, row_satds)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_frame_cost_recalculate_d{depth}")
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lc, qp = np.ascontiguousarray(lowres_costs, np.uint16), np.ascontiguousarray(qp_cutree_offset, np.float64)
    rows = np.zeros(height_in_cu, np.int32)
    return int(fn(width_in_cu, height_in_cu, lc.ctypes.data, qp.ctypes.data, rows.ctypes.data)), rows


def cutree_finish_hevc_aq(depth, width, height, part, blocks_in_row, intra_cost, inv_qscale, propagate_cost, fps_factor_q8, weight_delta, strength, qp_offset,
                          avx2=False):
    """CPU restatement of Lookahead::computeCUTreeQpOffset (qgSize >= 16) for one layer; returns dCuTreeOffset float64 [partitions]."""
    fn = getattr(lib(avx2), f"x265oracle_cutree_finish_hevc_aq_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    ic, iq, pc = np.ascontiguousarray(intra_cost, np.int32), np.ascontiguousarray(inv_qscale, np.int32), np.ascontiguousarray(propagate_cost, np.uint16)
    qo = np.ascontiguousarray(qp_offset, np.float64)
    out = np.zeros_like(qo)
    fn(width, height, part, blocks_in_row, ic.ctypes.data, iq.ctypes.data, pc.ctypes.data, int(fps_factor_q8), float(weight_delta), float(strength), qo.ctypes.data,
       out.ctypes.data)
    return out


def cutree_finish_qg8(depth, width_in_cu, height_in_cu, intra_cost, inv_qscale8x8, propagate_cost, qp_aq_offset, fps_factor_q8, weight_delta, strength,
                      qp_cutree_offset, avx2=False):
    """The --qg-size 8 branch of cuTreeFinish: offsets on the full-resolution grid [2 * height_in_cu, 2 * width_in_cu]."""
    fn = getattr(lib(avx2), f"x265oracle_cutree_finish_qg8_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    ic, iq = np.ascontiguousarray(intra_cost, np.int32), np.ascontiguousarray(inv_qscale8x8, np.int32)
    pc, qa = np.ascontiguousarray(propagate_cost, np.uint16), np.ascontiguousarray(qp_aq_offset, np.float64)
    out = np.ascontiguousarray(qp_cutree_offset, np.float64).copy()
    fn(width_in_cu, height_in_cu, ic.ctypes.data, iq.ctypes.data, pc.ctypes.data, qa.ctypes.data, int(fps_factor_q8), float(weight_delta), float(strength), out.ctypes.data)
    return out


def frame_cost_recalculate_qg8(depth, width_in_cu, height_in_cu, lowres_costs, qp_cutree_offset, avx2=False):
    fn = getattr(lib(avx2), f"x265oracle_frame_cost_recalculate_qg8_d{depth}")
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lc, qp = np.ascontiguousarray(lowres_costs, np.uint16), np.ascontiguousarray(qp_cutree_offset, np.float64)
    rows = np.zeros(height_in_cu, np.int32)
    return int(fn(width_in_cu, height_in_cu, lc.ctypes.data, qp.ctypes.data, rows.ctypes.data)), rows


def aq_frame(depth, y, stride, org, width, height, cb=None, cr=None, stride_c=0, org_c=0, qg_size=16, aq_mode=2, aq_strength=1.0, weightp=True,
             avx2=False):
    """CPU restatement of LookaheadTLD::calcAdaptiveQuantFrame.  y / cb / cr: padded planes (flat arrays, sample (0,0) at org / org_c).
    Returns (energy uint32 [blocks], qp_aq_offset float64 [blocks], inv_qscale int32 [blocks], wp_sum uint64 [3], wp_ssd uint64 [3])."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_aq_frame_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_ssize_t] * 2 + [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 5
    n = ((width + qg_size - 1) // qg_size) * ((height + qg_size - 1) // qg_size)
    energy, qp, inv = np.zeros(n, np.uint32), np.zeros(n, np.float64), np.zeros(n, np.int32)
    sm, ssd = np.zeros(3, np.uint64), np.zeros(3, np.uint64)
    es = y.itemsize
    fn(y.ctypes.data + org * es, None if cb is None else cb.ctypes.data + org_c * es, None if cr is None else cr.ctypes.data + org_c * es,
       stride, stride_c, width, height, qg_size, aq_mode, float(aq_strength), int(bool(weightp)),
       energy.ctypes.data, qp.ctypes.data, inv.ctypes.data, sm.ctypes.data, ssd.ctypes.data)
    return energy, qp, inv, sm, ssd


AQ_LAYER_DEPTH = {64: (1, 0, 1, 0), 32: (1, 1, 1, 0), 16: (1, 1, 1, 0), 8: (1, 1, 1, 1)}      # lowres.h:123-129, 64x64 CTUs, by qgSize


def aq_hevc_quadrants(depth, y, stride, org, width, height, part, avx2=False):
    """The integer half of LookaheadTLD::xPreanalyze: uint64 [partitions, 4 quadrants, (sum, sum of squares)]."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_aq_hevc_quadrants_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    n = ((width + part - 1) // part) * ((height + part - 1) // part)
    sums = np.zeros((n, 4, 2), np.uint64)
    fn(y.ctypes.data + org * y.itemsize, stride, width, height, part, sums.ctypes.data)
    return sums


def aq_hevc_frame(depth, y, stride, org, width, height, cb=None, cr=None, stride_c=0, org_c=0, qg_size=16, qp_adaptation_range=1.0, weightp=True,
                  avx2=False):
    """CPU restatement of calcAdaptiveQuantFrame with rc.hevcAq (xPreanalyze / xPreanalyzeQp).  Returns (layer_parts int32 [4],
    activity, qp_offset float64 [sum of the enabled layers' partitions], avg_activity float64 [4], inv_qscale int32 [deepest layer's
    partitions], wp_sum, wp_ssd uint64 [3])."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_aq_hevc_frame_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_ssize_t] * 2 + [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 7
    total = sum(((width + (64 >> d) - 1) // (64 >> d)) * ((height + (64 >> d) - 1) // (64 >> d)) for d in range(4) if AQ_LAYER_DEPTH[qg_size][d])
    parts, act, qp, avg = np.zeros(4, np.int32), np.zeros(total, np.float64), np.zeros(total, np.float64), np.zeros(4, np.float64)
    inv = np.zeros(total, np.int32)
    sm, ssd = np.zeros(3, np.uint64), np.zeros(3, np.uint64)
    es = y.itemsize
    fn(y.ctypes.data + org * es, None if cb is None else cb.ctypes.data + org_c * es, None if cr is None else cr.ctypes.data + org_c * es,
       stride, stride_c, width, height, 64, qg_size, float(qp_adaptation_range), int(bool(weightp)),
       parts.ctypes.data, act.ctypes.data, qp.ctypes.data, avg.ctypes.data, inv.ctypes.data, sm.ctypes.data, ssd.ctypes.data)
    deepest = max(d for d in range(4) if parts[d])
    return parts, act, qp, avg, inv[:parts[deepest]], sm, ssd


def lowres_weight_cost(depth, fenc, ref, stride, org, width, lines, intra_cost, weight, avx2=False):
    """CPU restatement of LookaheadTLD::weightCostLuma: weight = None (unweighted) or (scale, denom, offset)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_lowres_weight_cost_d{depth}")
    fn.restype = ctypes.c_uint32
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 4
    ic = np.ascontiguousarray(intra_cost, dtype=np.int32)
    es = fenc.itemsize
    w = (0, 0, 0, 0) if weight is None else (1,) + tuple(int(v) for v in weight)
    return int(fn(fenc.ctypes.data + org * es, ref.ctypes.data + org * es, stride, width, lines, ic.ctypes.data, *w))


def weights_analyse(depth, fenc, ref, stride, org, width, lines, intra_cost, wp_ssd, wp_sum, avx2=False):
    """CPU restatement of LookaheadTLD::weightsAnalyse.  wp_ssd / wp_sum: (current, reference).  Returns (weight or None, minscore,
    origscore) with weight = (scale, denom, offset)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_weights_analyse_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    ic = np.ascontiguousarray(intra_cost, dtype=np.int32)
    ssd, sm = np.asarray(wp_ssd, dtype=np.uint64), np.asarray(wp_sum, dtype=np.uint64)
    out = np.zeros(6, np.int64)
    es = fenc.itemsize
    fn(fenc.ctypes.data + org * es, ref.ctypes.data + org * es, stride, width, lines, ic.ctypes.data, ssd.ctypes.data, sm.ctypes.data, out.ctypes.data)
    return (tuple(int(v) for v in out[1:4]) if out[0] else None), int(out[4]), int(out[5])


def weight_plane(depth, plane, weight):
    """primitives.weight_pp over a whole buffer the way weightsAnalyse applies the chosen weight (slicetype.cpp:943-952)."""
    scale, denom, offset = weight
    corr = 14 - depth
    rnd = ((1 << (denom - 1)) if denom else 0) << corr
    v = (scale * (plane.astype(np.int64) << corr) + rnd) >> (denom + corr)
    return np.clip(v + (offset << (depth - 8)), 0, (1 << depth) - 1).astype(plane.dtype)


def motion_estimate(depth, fenc, fref, stride, org, method, subme, merange, cost_q, qoff, mvmin, mvmax, jobs, nthreads=0, avx2=False,
                    mvc=None, num_mvc=None):
    """CPU restatement of MotionEstimate::motionEstimate over a job array (numpy structured array with the fields of
    x265hip_me_search_job); results are written into a copy that is returned."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_motion_estimate_mvc_d{depth}")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + \
                  [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    mv = np.ascontiguousarray(mvc, dtype=np.int32) if mvc is not None else None
    nm = np.ascontiguousarray(num_mvc, dtype=np.int32) if num_mvc is not None else None
    out = np.ascontiguousarray(jobs).copy()
    cq = np.ascontiguousarray(cost_q, dtype=np.uint16)
    es = fenc.itemsize
    rc = fn(fenc.ctypes.data + org * es, fref.ctypes.data + org * es, stride, method, subme, merange, cq.ctypes.data, qoff,
            mvmin[0], mvmin[1], mvmax[0], mvmax[1], out.ctypes.data, len(out), nthreads,
            mv.ctypes.data if mv is not None else None, nm.ctypes.data if nm is not None else None)
    if rc:
        raise RuntimeError("x265oracle_motion_estimate: unsupported method or PU size")
    return out


def deblock_bs_inter(depth, width, height, level, mv, num_sig, avx2=False, intra=None):
    """CPU restatement of getBoundaryStrength for a picture of square blocks (intra: optional uint8 flags per block, Bs 2 on their
    edges).  Returns (bs_ver, bs_hor)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_deblock_bs_d{depth}")
    bv = np.zeros((height // 4) * (width // 8), np.uint8)
    bh = np.zeros((height // 8) * (width // 4), np.uint8)
    m = np.ascontiguousarray(mv, dtype=np.int32)
    ns = np.ascontiguousarray(num_sig, dtype=np.uint32)
    it = None if intra is None else np.ascontiguousarray(intra, dtype=np.uint8)
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    fn(width, height, level, m.ctypes.data, ns.ctypes.data, None if it is None else it.ctypes.data, bv.ctypes.data, bh.ctypes.data)
    return bv, bh


def deblock_bs_b(depth, width, height, level, mv0, mv1, ref0, ref1, num_sig, slice_b=True, intra=None, avx2=False):
    """CPU restatement of the whole of getBoundaryStrength (deblock.cpp:191-247): per block a reference picture id per list
    (int8, -1 = list unused) and an mv record array per list; slice_b selects the B-picture comparison.  Returns (bs_ver, bs_hor)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_deblock_bs_b_d{depth}")
    bv = np.zeros((height // 4) * (width // 8), np.uint8)
    bh = np.zeros((height // 8) * (width // 4), np.uint8)
    m0 = np.ascontiguousarray(mv0, dtype=np.int32)
    m1 = None if mv1 is None else np.ascontiguousarray(mv1, dtype=np.int32)
    r0 = None if ref0 is None else np.ascontiguousarray(ref0, dtype=np.int8)
    r1 = None if ref1 is None else np.ascontiguousarray(ref1, dtype=np.int8)
    ns = np.ascontiguousarray(num_sig, dtype=np.uint32)
    it = None if intra is None else np.ascontiguousarray(intra, dtype=np.uint8)
    ptr = lambda a: None if a is None else a.ctypes.data
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 8
    fn(width, height, level, int(bool(slice_b)), ptr(m0), ptr(m1), ptr(r0), ptr(r1), ptr(ns), ptr(it), bv.ctypes.data, bh.ctypes.data)
    return bv, bh


def deblock_chroma(depth, cb, cr, stride_c, org_c, width, height, bs_ver, bs_hor, qp, qp_map=None, cb_qp_offset=0, cr_qp_offset=0,
                   tc_offset_div2=0, avx2=False):
    """CPU restatement of edgeFilterChroma over the two chroma planes of a 4:2:0 picture (width / height = luma size); returns the
    filtered copies of cb, cr (flat padded planes, sample (0,0) at element org_c)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_deblock_chroma_d{depth}")
    ob, orr = cb.copy(), cr.copy()
    es = cb.itemsize
    qm = None if qp_map is None else np.ascontiguousarray(qp_map, dtype=np.int8)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    fn(ob.ctypes.data + org_c * es, orr.ctypes.data + org_c * es, stride_c, width, height, bs_ver.ctypes.data, bs_hor.ctypes.data, qp,
       None if qm is None else qm.ctypes.data, cb_qp_offset, cr_qp_offset, tc_offset_div2)
    return ob, orr


def deblock_luma(depth, rec, stride, org, width, height, bs_ver, bs_hor, qp, qp_map=None, beta_offset_div2=0, tc_offset_div2=0, avx2=False):
    """CPU restatement of edgeFilterLuma over a picture; returns the filtered copy of `rec` (padded plane)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_deblock_luma_d{depth}")
    out = rec.copy()
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    qm = np.ascontiguousarray(qp_map, dtype=np.int8) if qp_map is not None else None
    fn(out.ctypes.data + org * out.itemsize, stride, width, height, bs_ver.ctypes.data, bs_hor.ctypes.data, qp,
       qm.ctypes.data if qm is not None else None, beta_offset_div2, tc_offset_div2)
    return out


def intra_recon(depth, n, fenc, fenc_stride, nb, recon_len, recon_stride, qp, intra_slice, jobs, nthreads=0, avx2=False, chroma=False):
    """CPU restatement of the intra TU candidate set (search.cpp:335-373 through the oracle primitives); chroma=True: the 4:2:0
    chroma flavour (predIntraChromaAng: unfiltered neighbours, no edge smoothing; DCT for 4x4).
    jobs: numpy records {off[4], arg[4]}.  Returns (recon flat array, levels, num_sig, dist)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_intra_recon{'_chroma' if chroma else ''}_d{depth}")
    njobs = len(jobs)
    recon = np.zeros(recon_len, dtype=fenc.dtype)
    levels = np.zeros(njobs * n * n, dtype=np.int16)
    num_sig = np.zeros(njobs, dtype=np.uint32)
    dist = np.zeros(njobs, dtype=np.uint64)
    j = np.ascontiguousarray(jobs)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rc = fn(fenc.ctypes.data, fenc_stride, nb.ctypes.data, recon.ctypes.data, recon_stride, n, qp, intra_slice,
            j.ctypes.data, njobs, levels.ctypes.data, num_sig.ctypes.data, dist.ctypes.data, nthreads)
    assert rc == 0
    return recon, levels, num_sig, dist


def lowres_cost(depth, cur, ref_planes, stride, org, width_in_cu, height_in_cu, cost_q, qoff, intra_cost, inv_qscale=None, avx2=False,
                ref1_planes=None, do_search=(1, 1), bframe_bias=0, mvs_in=None, mv_costs_in=None, ref_bi_planes=None):
    """CPU restatement of CostEstimateGroup::estimateFrameCost (slicetype.cpp:3115-3388) for a P picture (ref1_planes None) or a B
    picture.  cur: the current picture's plane 0; ref_planes / ref1_planes: the list-0 / list-1 reference's four planes (flat arrays,
    pixel (0,0) at element `org`).  P: returns (mvs int32 [n, 2], mv_costs, lowres_costs, row_satds, frame int64 [3] = costEst,
    costEstAq, intraMbs).  B: returns ((mvs0, mvs1), (mv_costs0, mv_costs1), lowres_costs, row_satds, frame int64 [4] with the
    returned score last).  ref_bi_planes: --weightp for B pictures - ref_planes are then the weighted list-0 planes, ref_bi_planes the
    unweighted ones the bi-directional candidates keep."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_lowres_cost_wp_d{depth}")
    n = width_in_cu * height_in_cu
    mvs = [np.zeros((n, 2), np.int32) if mvs_in is None or mvs_in[i] is None else np.ascontiguousarray(mvs_in[i], np.int32).copy() for i in range(2)]
    mvc = [np.zeros(n, np.int32) if mv_costs_in is None or mv_costs_in[i] is None else np.ascontiguousarray(mv_costs_in[i], np.int32).copy() for i in range(2)]
    lc, rows, frame = np.zeros(n, np.uint16), np.zeros(height_in_cu, np.int32), np.zeros(4, np.int64)
    es = cur.itemsize
    cq = np.ascontiguousarray(cost_q, dtype=np.uint16)
    ic = np.ascontiguousarray(intra_cost, dtype=np.int32)
    iq = None if inv_qscale is None else np.ascontiguousarray(inv_qscale, dtype=np.int32)
    r0 = (ctypes.c_void_p * 4)(*[p.ctypes.data + org * es for p in ref_planes])
    r1 = None if ref1_planes is None else (ctypes.c_void_p * 4)(*[p.ctypes.data + org * es for p in ref1_planes])
    rb = None if ref_bi_planes is None else (ctypes.c_void_p * 4)(*[p.ctypes.data + org * es for p in ref_bi_planes])
    ds = (ctypes.c_int * 2)(*do_search)
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8
    rc = fn(cur.ctypes.data + org * es, r0, r1, stride, width_in_cu, height_in_cu, cq.ctypes.data, qoff, ic.ctypes.data,
            None if iq is None else iq.ctypes.data, ds, bframe_bias, mvs[0].ctypes.data, mvc[0].ctypes.data, mvs[1].ctypes.data, mvc[1].ctypes.data,
            lc.ctypes.data, rows.ctypes.data, frame.ctypes.data, rb)
    assert rc == 0
    if ref1_planes is None:
        return mvs[0], mvc[0], lc, rows, frame[:3]
    return (mvs[0], mvs[1]), (mvc[0], mvc[1]), lc, rows, frame


def sao_stats(depth, fenc, rec, stride, org, width, height, nthreads=0, avx2=False, ctu=(64, 64), plane_offset=0):
    """CPU restatement of SAO::calcSaoStatsCTU (sao.cpp:735-917) for every CTU of the luma plane.
    Returns (count, offset_org), int32 [numCtu, 5, 32] each (type order EO_0..EO_3, BO)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_sao_stats_plane_d{depth}")
    nctu = ((width + ctu[0] - 1) // ctu[0]) * ((height + ctu[1] - 1) // ctu[1])
    cnt, off = np.zeros((nctu, 5, 32), np.int32), np.zeros((nctu, 5, 32), np.int32)
    es = fenc.itemsize
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert fn(fenc.ctypes.data + org * es, rec.ctypes.data + org * es, stride, width, height, ctu[0], ctu[1], plane_offset,
              cnt.ctypes.data, off.ctypes.data, nthreads) == 0
    return cnt, off


def sao_decide(depth, count, offset_org, avx2=False):
    """CPU restatement of SAO::saoStatsInitialOffset (sao.cpp:1378-1433) + the distortion-only type choice (see x265hip_sao_decide).
    count / offset_org int32 [numCtu, 5, 32].  Returns (initial offsets int32 [numCtu, 5, 32], params int32 [numCtu, 7])."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_sao_decide_d{depth}")
    c = np.ascontiguousarray(count, dtype=np.int32).reshape(-1, 160)
    o = np.ascontiguousarray(offset_org, dtype=np.int32).reshape(-1, 160)
    init, params = np.zeros((c.shape[0], 5, 32), np.int32), np.zeros((c.shape[0], 7), np.int32)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(c.ctypes.data, o.ctypes.data, c.shape[0], init.ctypes.data, params.ctypes.data) == 0
    return init, params


def sao_rdo(depth, counts, offset_orgs, ctus_w, ctus_h, lambda_ctu, ctx_merge, ctx_type, entropy_bits, sao_flag=(1, 1), avx2=False):
    """CPU restatement of SAO::rdoSaoUnitCu over a picture (oracle/x265_oracle_pipeline6.c; sao.cpp:1225-1760): counts / offset_orgs =
    lists (1 or 3 planes) of int32 [numCtu, 5, 32]; lambda_ctu int64 [numCtu, 2]; ctx_* = the slice's initial context states;
    entropy_bits = the host's 128 per-state bit costs.  Returns (params: list of int32 [numCtu, 7] = typeIdx, bandPos, offset[4],
    mergeMode (0 none / 1 left / 2 up), numNoSao int32 [2])."""
    fn = getattr(lib(avx2), f"x265oracle_sao_rdo_d{depth}")
    planes = len(counts)
    nctu = ctus_w * ctus_h
    cs = [np.ascontiguousarray(c, dtype=np.int32).reshape(nctu, 160) for c in counts]
    os_ = [np.ascontiguousarray(o, dtype=np.int32).reshape(nctu, 160) for o in offset_orgs]
    lam = np.ascontiguousarray(lambda_ctu, dtype=np.int64).reshape(nctu, 2)
    bits = np.ascontiguousarray(entropy_bits, dtype=np.uint32)
    assert bits.size == 128
    params = [np.zeros((nctu, 7), np.int32) for _ in range(planes)]
    nos = np.zeros(2, np.int32)
    P3 = ctypes.c_void_p * planes
    flag = (ctypes.c_int * 2)(*[int(f) for f in sao_flag])
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(P3(*[c.ctypes.data for c in cs]), P3(*[o.ctypes.data for o in os_]), planes, ctus_w, ctus_h, lam.ctypes.data, int(ctx_merge), int(ctx_type),
            bits.ctypes.data, flag, P3(*[p.ctypes.data for p in params]), nos.ctypes.data)
    assert rc == 0
    return params, nos


def sao_apply(depth, src, stride, org, width, height, params, nthreads=0, avx2=False, ctu=(64, 64)):
    """CPU restatement of SAO::generateLumaOffsets / applyPixelOffsets (sao.cpp:572-630, 274-570) for every CTU; params int32
    [numCtu, 7] = typeIdx, bandPos, offset[4], mergeLeft.  Returns the offset picture (a copy of src outside the picture area)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_sao_apply_plane_d{depth}")
    dst = src.copy()
    es = src.itemsize
    pr = np.ascontiguousarray(params, dtype=np.int32)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    assert fn(src.ctypes.data + org * es, dst.ctypes.data + org * es, stride, width, height, ctu[0], ctu[1], pr.ctypes.data, nthreads) == 0
    return dst


def inter_recon_chroma(depth, fenc, fref, stride, org, width, height, level, mv, qp, intra_slice=0, nthreads=0, avx2=False):
    """CPU restatement of the chroma half of the inter TU stage for one plane of a 4:2:0 picture (flat padded planes, sample (0,0) at
    element `org`; width / height = luma size).  Returns (recon plane, levels, num_sig, dist)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_inter_recon_chroma_d{depth}")
    nctu = (width // 64) * (height // 64)
    nc = 4 << level
    nblk = (32 // nc) ** 2
    recon = np.zeros_like(fenc)
    levels = np.zeros(nctu * nblk * nc * nc, dtype=np.int16)
    num_sig = np.zeros(nctu * nblk, dtype=np.uint32)
    dist = np.zeros(nctu * nblk, dtype=np.uint64)
    es = fenc.itemsize
    m = np.ascontiguousarray(mv, dtype=np.int32)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert fn(fenc.ctypes.data + org * es, stride, fref.ctypes.data + org * es, stride, recon.ctypes.data + org * es, stride,
              width, height, level, m.ctypes.data, qp, intra_slice, levels.ctypes.data, num_sig.ctypes.data, dist.ctypes.data, nthreads) == 0
    return recon, levels, num_sig, dist


def inter_recon_bi(depth, fenc, stride, org, fref0, fref1, width, height, level, mv0, mv1, qp, dir_flags=None, intra_slice=0, nthreads=0, avx2=False,
                   weights=None):
    """CPU restatement of the bi-predictive inter TU stage (all planes share stride / org).  Returns (recon, levels, num_sig, dist).
    weights: (list 0, list 1), each None (no table) or (present, weight, offset, log2_denom) - explicit weighted prediction."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_inter_recon_bi_d{depth}")
    setw = getattr(L, f"x265oracle_set_pred_weights_d{depth}")
    setw.restype = None
    setw.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    wrec = [None if (weights is None or w is None) else np.asarray(w, dtype=np.int32) for w in (weights or (None, None))]
    setw(*[None if w is None else w.ctypes.data for w in wrec])
    nctu = (width // 64) * (height // 64)
    n = 8 << level
    nblk = (64 // n) ** 2
    recon = np.zeros_like(fenc)
    levels = np.zeros(nctu * nblk * n * n, dtype=np.int16)
    num_sig = np.zeros(nctu * nblk, dtype=np.uint32)
    dist = np.zeros(nctu * nblk, dtype=np.uint64)
    es = fenc.itemsize
    m0, m1 = np.ascontiguousarray(mv0, dtype=np.int32), np.ascontiguousarray(mv1, dtype=np.int32)
    d = None if dir_flags is None else np.ascontiguousarray(dir_flags, dtype=np.uint8)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert fn(fenc.ctypes.data + org * es, stride, fref0.ctypes.data + org * es, fref1.ctypes.data + org * es, stride,
              recon.ctypes.data + org * es, stride, width, height, level, m0.ctypes.data, m1.ctypes.data, None if d is None else d.ctypes.data,
              qp, intra_slice, levels.ctypes.data, num_sig.ctypes.data, dist.ctypes.data, nthreads) == 0
    setw(None, None)
    return recon, levels, num_sig, dist


class pred_capture:
    """Context manager: the inter stages called inside copy every block's PREDICTION into `plane` (unpadded 2-D array of the stage's
    geometry: width x height for the luma stages, half of that for the chroma stages)."""

    def __init__(self, depth, plane, avx2=False):
        self.fn = getattr(lib(avx2), f"x265oracle_set_pred_capture_d{depth}")
        self.fn.restype = None
        self.fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t]
        self.plane = plane

    def __enter__(self):
        self.fn(self.plane.ctypes.data, self.plane.shape[1])
        return self.plane

    def __exit__(self, *exc):
        self.fn(None, 0)


def inter_recon_chroma_bi(depth, fenc, fref0, fref1, stride, org, width, height, level, mv0, mv1, qp, dir_flags=None, intra_slice=0, nthreads=0,
                          avx2=False, weights=None):
    """CPU restatement of one chroma plane of the bi-predictive / weighted inter TU stage (planes share stride / org; width / height =
    luma size).  Returns (recon, levels, num_sig, dist)."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_inter_recon_chroma_bi_d{depth}")
    setw = getattr(L, f"x265oracle_set_pred_weights_d{depth}")
    setw.restype = None
    setw.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    wrec = [None if (weights is None or w is None) else np.asarray(w, dtype=np.int32) for w in (weights or (None, None))]
    setw(*[None if w is None else w.ctypes.data for w in wrec])
    nctu = (width // 64) * (height // 64)
    n = 4 << level
    nblk = (32 // n) ** 2
    recon = np.zeros_like(fenc)
    levels = np.zeros(nctu * nblk * n * n, dtype=np.int16)
    num_sig = np.zeros(nctu * nblk, dtype=np.uint32)
    dist = np.zeros(nctu * nblk, dtype=np.uint64)
    es = fenc.itemsize
    m0, m1 = np.ascontiguousarray(mv0, dtype=np.int32), np.ascontiguousarray(mv1, dtype=np.int32)
    d = None if dir_flags is None else np.ascontiguousarray(dir_flags, dtype=np.uint8)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert fn(fenc.ctypes.data + org * es, stride, fref0.ctypes.data + org * es, fref1.ctypes.data + org * es, stride,
              recon.ctypes.data + org * es, stride, width, height, level, m0.ctypes.data, m1.ctypes.data, None if d is None else d.ctypes.data,
              qp, intra_slice, levels.ctypes.data, num_sig.ctypes.data, dist.ctypes.data, nthreads) == 0
    setw(None, None)
    return recon, levels, num_sig, dist


def phase_planes(depth, src, stride, rows, chroma=False, avx2=False):
    """CPU restatement of x265hip_phase_planes on top of the oracle's interpolation primitives: src = flat padded plane (stride * rows
    samples).  Returns [15 or 63, rows, stride]; the 8-sample border of every plane is zero (undefined in the product)."""
    fn = getattr(lib(avx2), f"x265oracle_phase_planes_d{depth}")
    s = np.ascontiguousarray(src).reshape(-1)
    assert s.size == stride * rows and stride % 8 == 0 and rows % 8 == 0
    out = np.zeros((63 if chroma else 15, rows, stride), s.dtype)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn.restype = None
    fn(s.ctypes.data, stride, rows, int(bool(chroma)), out.ctypes.data)
    return out


class WaList(ctypes.Structure):
    """x265oracle_wa_list (oracle/x265_oracle_pipeline7.c)."""
    _fields_ = [("lowres", ctypes.c_void_p * 4), ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p), ("mvs", ctypes.c_void_p),
                ("wp_ssd", ctypes.c_uint64 * 3), ("wp_sum", ctypes.c_uint64 * 3)]


def weight_analyse(depth, cur, refs, pic_width, pic_height, intra_cost, avx2=False):
    """CPU restatement of weightAnalyse (encoder/weightPrediction.cpp:222-497), 4:2:0.  cur = dict(lowres=(array, org), lowres_stride, lowres_width,
    lowres_lines, cb=(array, org), cr=(array, org), stride_c, wp_ssd[3], wp_sum[3]); refs = 1 or 2 dicts(lowres=[(array, org)] * 4, cb, cr, mvs =
    int32 [blocks, 2] or None, wp_ssd, wp_sum).  Returns (weights int32 [2, 3, 4] = present / weight / denom / offset, denoms int32 [2, 2])."""
    L = lib(avx2)
    fn = getattr(L, f"x265oracle_weight_analyse_d{depth}")
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    es = cur["lowres"][0].itemsize
    at = lambda pair: pair[0].ctypes.data + pair[1] * es
    lists = (WaList * 2)()
    keep = []
    for i, r in enumerate(refs):
        for k in range(4):
            lists[i].lowres[k] = at(r["lowres"][k])
        lists[i].cb, lists[i].cr = at(r["cb"]), at(r["cr"])
        if r.get("mvs") is not None:
            m = np.ascontiguousarray(r["mvs"], np.int32)
            keep.append(m)
            lists[i].mvs = m.ctypes.data
        for k in range(3):
            lists[i].wp_ssd[k], lists[i].wp_sum[k] = int(r["wp_ssd"][k]), int(r["wp_sum"][k])
    ssd, sm = np.asarray(cur["wp_ssd"], np.uint64), np.asarray(cur["wp_sum"], np.uint64)
    ic = np.ascontiguousarray(intra_cost, np.int32)
    half = max(cur["lowres_stride"] * cur["lowres_lines"], cur["stride_c"] * (pic_height // 2)) + 64
    scratch = np.zeros(2 * half, cur["lowres"][0].dtype)
    out, den = np.zeros((2, 3, 4), np.int32), np.zeros((2, 2), np.int32)
    fn(at(cur["lowres"]), cur["lowres_stride"], cur["lowres_width"], cur["lowres_lines"], at(cur["cb"]), at(cur["cr"]), cur["stride_c"], pic_width, pic_height,
       ic.ctypes.data, ssd.ctypes.data, sm.ctypes.data, len(refs), ctypes.addressof(lists), scratch.ctypes.data, half, out.ctypes.data, den.ctypes.data)
    return out, den


# ---- round 6: sub-sample cost tables (oracle/x265_oracle_pipeline8.c) -----------------------------------------------------------------
def cost_pu_list(shapes, depth=8):
    """[n, 5] x, y, w, h, LumaPU enum of the PU list (squares / + 2NxN, Nx2N / + AMP)."""
    fn = getattr(lib(), f"x265oracle_cost_pu_list_d{depth}")
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    out = np.zeros((209, 5), np.int32)
    n = fn(shapes, out.ctypes.data)
    return out[:n].copy()


def cost_positions(subme, depth=8):
    """[n, 2] quarter-sample offsets a refinement of SubpelWorkload row `subme` can measure, raster order."""
    fn = getattr(lib(), f"x265oracle_cost_positions_d{depth}")
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    out = np.zeros((169, 2), np.int8)
    n = fn(subme, out.ctypes.data)
    return out[:n].copy()


def cost_record_bytes(subme, sad_costs=0):
    n = len(cost_positions(subme))
    return ((8 + 2 * n + 3) & ~3) + (((4 + 2 * n + 3) & ~3) if sad_costs else 0)


def cost_candidates(surf, centres, nctu, window, shapes, k, depth=8, avx2=False, mv_cost=None):
    """surf: int32 SAD rasters of the 85 squares (me_fullsearch's surfaces, I32 records); centres int16 [nctu, 2] or None.
    Returns int16 [nctu, npu, k, 2]."""
    fn = getattr(lib(avx2), f"x265oracle_cost_candidates_d{depth}")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = None
    npu = len(cost_pu_list(shapes))
    mc = None if mv_cost is None else np.ascontiguousarray(mv_cost, np.uint16)
    assert mc is None or mc.size == 2 * window + 1
    s = np.ascontiguousarray(surf, np.int32)
    c = None if centres is None else np.ascontiguousarray(centres, np.int16)
    out = np.zeros((nctu, npu, k, 2), np.int16)
    fn(s.ctypes.data, None if c is None else c.ctypes.data, nctu, window, shapes, k, out.ctypes.data, None if mc is None else mc.ctypes.data)
    return out


def cost_tables(depth, fenc, ref, stride, stride_c, margin_x, margin_y, margin_y_c, width, ctu_row0, ctu_rows, shapes, k, subme, chroma, cand, avx2=False, sad_costs=0):
    """fenc / ref: three flat padded planes each (allocation starts; the chroma ones may be None when chroma = 0).  cand: int16
    [ctu_rows * width / 64, npu, k, 2].  Returns uint8 [ctus, npu, k, record bytes]: the reference's own subpelCompare route per value; sad_costs = 1 appends
    the costs of the SAD-typed comparisons (cmp = the PU's sad) to every record."""
    fn = getattr(lib(avx2), f"x265oracle_cost_tables_d{depth}")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_ssize_t] + [ctypes.c_int] * 10 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    fn.restype = None
    f = [None if p is None else np.ascontiguousarray(p).reshape(-1) for p in fenc]
    r = [None if p is None else np.ascontiguousarray(p).reshape(-1) for p in ref]
    fp = (ctypes.c_void_p * 3)(*[None if p is None else p.ctypes.data for p in f])
    rp = (ctypes.c_void_p * 3)(*[None if p is None else p.ctypes.data for p in r])
    c = np.ascontiguousarray(cand, np.int16)
    npu, rec = len(cost_pu_list(shapes)), cost_record_bytes(subme, sad_costs)
    nctu = ctu_rows * (width // 64)
    assert c.size == nctu * npu * k * 2
    out = np.zeros((nctu, npu, k, rec), np.uint8)
    fn(fp, rp, stride, stride_c, margin_x, margin_y, margin_y_c, width, ctu_row0, ctu_rows, shapes, k, subme, int(bool(chroma)), c.ctypes.data, out.ctypes.data, int(bool(sad_costs)))
    return out
