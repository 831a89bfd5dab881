"""Driver of the stage-level seam (tier T3 with a batch-layer consumer): configures oracle/_ref/libx265ref<depth>_seam.so
(oracle/ref_seam.cpp - the reference-side binding) with a SAD-surface provider and hands back the table filler for
x265ref_encode.  Two providers with the same three entry points (x265hip_me_cache_submit / _surface / _ready signatures):

  GpuProvider     libx265hip.so's frame-granular cache (csrc/me_cache.hip): one exhaustive-search launch per (picture, reference),
                  surfaces streamed to pinned host memory CTU row by CTU row                          -> the product path
  OracleProvider  the oracle's exhaustive search on the CPU, synchronous (tests in the GPU-less container: proves the binding, the
                  lookup arithmetic and the partition decomposition against the real encoder)        -> checker only

Test infrastructure; the product is what GpuProvider calls."""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SUBMIT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)
SURFACE = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)
READY = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)
SURF_I32, SURF_PACKED, SURF_PACKED_T = 0, 1, 2


def geometry(width, height):
    """PicYuv layout for --ctu 64 (common/picyuv.cpp:87-114): whole CTUs, margins ctu + 32 / ctu + 16."""
    w64, h64 = (width + 63) // 64 * 64, (height + 63) // 64 * 64
    mx, my = 64 + 32, 64 + 16
    return dict(width=w64, height=h64, stride=w64 + 2 * mx, margin_x=mx, margin_y=my,
                rows=h64 + 2 * my, stride_c=w64 // 2 + 2 * mx, rows_c=h64 // 2 + 2 * (my >> 1))


def seam_lib(depth):
    path = os.path.join(ROOT, "oracle", "_ref", f"libx265ref{depth}_seam.so")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    lib = ctypes.CDLL(path)
    lib.x265ref_encode.restype = ctypes.c_long
    lib.x265ref_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    lib.x265ref_seam_configure.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_ssize_t] + [ctypes.c_int] * 4
    lib.x265ref_seam_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    return lib


STAT_NAMES = ("lookups_served", "outside_window", "row_not_ready", "motion_estimate_calls", "calls_with_lookup_context", "pair_submits",
              "verify_mismatches", "no_free_slot", "foreign_geometry", "verify")


def stats(lib):
    out = (ctypes.c_uint64 * 10)()
    lib.x265ref_seam_stats(out)
    d = dict(zip(STAT_NAMES, [int(v) for v in out]))
    tot = d["lookups_served"] + d["outside_window"] + d["row_not_ready"]
    d["lookup_hit_rate"] = round(d["lookups_served"] / tot, 4) if tot else None
    return d


class OracleProvider:
    """CPU stand-in for x265hip_me_cache (checker only): surfaces from oracle/x265_oracle_pipeline.c, int32 records."""

    def __init__(self, depth, geo, rng, slots):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_api
        self.O, self.depth, self.geo, self.range, self.slots = oracle_api, depth, geo, rng, slots
        self.es = 1 if depth == 8 else 2
        self.nctu = (geo["width"] // 64) * (geo["height"] // 64)
        nc = 2 * rng + 1
        self.words = self.nctu * nc * ((nc + 3) // 4) * 340
        self.surf = [np.zeros(self.words, np.int32) for _ in range(slots)]
        self.flags = [np.zeros(geo["height"] // 64, np.int32) for _ in range(slots)]
        self.gen = [0] * slots
        self.plane_elems = geo["stride"] * (geo["height"] + 2 * geo["margin_y"])
        self.org = geo["margin_y"] * geo["stride"] + geo["margin_x"]
        self.fills = 0
        self.format = SURF_I32
        self._cb = (SUBMIT(self._submit), SURFACE(self._surface), READY(self._ready))

    def _plane(self, ptr):
        dt = np.uint8 if self.depth == 8 else np.uint16
        raw = (ctypes.c_uint8 * (self.plane_elems * self.es)).from_address(ptr)
        return np.frombuffer(raw, dtype=dt)

    def _submit(self, ctx, slot, fenc, key, ref):
        g = self.geo
        zero = np.zeros(2 * self.range + 1, np.uint16)
        surf, _ = self.O.me_fullsearch(self.depth, self._plane(fenc), g["stride"], self.org, self._plane(ref), g["stride"], self.org,
                                       g["width"], g["height"], self.range, 0, self.nctu, zero, zero, want_surf=True, want_best=False)
        self.surf[slot][:] = surf
        self.gen[slot] += 1
        self.flags[slot][:] = self.gen[slot]
        self.fills += 1
        return self.gen[slot]

    def _surface(self, ctx, slot):
        return self.surf[slot].ctypes.data

    def _ready(self, ctx, slot):
        return self.flags[slot].ctypes.data

    def pointers(self):
        sub, surf, rdy = (ctypes.cast(c, ctypes.c_void_p) for c in self._cb)
        return None, sub, None, surf, rdy          # no batch entry: the binding falls back to single submits

    def report(self):
        return {"provider": "oracle (CPU checker)", "fills": self.fills}

    def close(self):
        pass


class CacheParams(ctypes.Structure):
    """x265hip_me_cache_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int), ("stride", ctypes.c_ssize_t),
                ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int), ("range", ctypes.c_int), ("surf_format", ctypes.c_int), ("slots", ctypes.c_int)]


class CacheStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("fills", "failed", "batches", "us_upload", "us_kernel", "us_download", "bytes_downloaded", "surface_bytes")]


class GpuProvider:
    """libx265hip.so's x265hip_me_cache: the product path."""

    def __init__(self, depth, geo, rng, slots, surf_format=None):
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        self.A, self.L = A, A.lib()
        # 8-bit default: record-contiguous packed rows from the row-walking kernel (surfaces alone: 0.31 ms against 0.39 ms for the
        # record-per-lane kernel's chunk-major rows at 4K +-24, profiles/r02_me_cand_ab.txt); SURF_PACKED_T on request
        self.format = surf_format if surf_format is not None else (SURF_PACKED if depth == 8 else SURF_I32)
        p = CacheParams(depth, geo["width"], geo["height"], geo["stride"], geo["margin_x"], geo["margin_y"], rng, self.format, slots)
        self.handle = ctypes.c_void_p()
        self.L.x265hip_me_cache_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(CacheParams)]
        A.check(self.L.x265hip_me_cache_create(ctypes.byref(self.handle), ctypes.byref(p)), "x265hip_me_cache_create")
        self.L.x265hip_me_cache_destroy.argtypes = [ctypes.c_void_p]
        self.L.x265hip_me_cache_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(CacheStats)]

    def pointers(self):
        L = self.L
        return (self.handle, ctypes.cast(L.x265hip_me_cache_submit, ctypes.c_void_p), ctypes.cast(L.x265hip_me_cache_submit_batch, ctypes.c_void_p),
                ctypes.cast(L.x265hip_me_cache_surface, ctypes.c_void_p), ctypes.cast(L.x265hip_me_cache_ready, ctypes.c_void_p))

    def report(self):
        st = CacheStats()
        self.L.x265hip_me_cache_stats(self.handle, ctypes.byref(st))
        n = max(1, st.batches)
        return {"provider": "x265hip_me_cache (one x265hip_me_fullsearch launch per (picture, reference); a picture's references as one batch, "
                            "surfaces downloaded row-interleaved)", "fills": int(st.fills), "failed": int(st.failed), "batches": int(st.batches),
                "ms_per_batch": {"uploads_and_kernels": round(st.us_kernel / n / 1e3, 3), "download": round(st.us_download / n / 1e3, 3)},
                "surface_mbytes_per_pair": round(st.surface_bytes / 1e6, 1),
                "download_gbytes_per_s": round(st.bytes_downloaded / max(1, st.us_download) / 1e3, 2)}

    def close(self):
        if self.handle:
            self.L.x265hip_me_cache_destroy(self.handle)
            self.handle = None


PH_SUBMIT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
PH_PLANES = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
PH_READY = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)
SUB_STAT_NAMES = ("subpel_compares_served", "passed_on_planes_not_arrived", "searches_without_context", "pictures_submitted", "verify_mismatches",
                  "searches_without_slot")


class OraclePhaseProvider:
    """CPU stand-in for x265hip_phase_cache (checker only): the oracle's phase planes, computed synchronously inside submit."""

    def __init__(self, depth, geo, slots):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_api
        self.O, self.depth, self.geo, self.slots = oracle_api, depth, geo, slots
        self.dt = np.uint8 if depth == 8 else np.uint16
        g = geo
        self.out = [[np.zeros((15, g["rows"], g["stride"]), self.dt)] + [np.zeros((63, g["rows_c"], g["stride_c"]), self.dt) for _ in range(2)]
                    for _ in range(slots)]
        self.flags = [np.zeros(2, np.int32) for _ in range(slots)]
        self.gen = [0] * slots
        self.fills = 0
        self._cb = (PH_SUBMIT(self._submit), PH_PLANES(self._planes), PH_READY(self._ready))

    def _view(self, ptr, n):
        raw = (ctypes.c_uint8 * (n * np.dtype(self.dt).itemsize)).from_address(ptr)
        return np.frombuffer(raw, dtype=self.dt)

    def _submit(self, ctx, slot, luma, cb, cr):
        g = self.geo
        self.out[slot][0][:] = self.O.phase_planes(self.depth, self._view(luma, g["stride"] * g["rows"]), g["stride"], g["rows"])
        for k, ptr in ((1, cb), (2, cr)):
            self.out[slot][k][:] = self.O.phase_planes(self.depth, self._view(ptr, g["stride_c"] * g["rows_c"]), g["stride_c"], g["rows_c"], chroma=True)
        self.gen[slot] += 1
        self.flags[slot][:] = self.gen[slot]
        self.fills += 1
        return self.gen[slot]

    def _planes(self, ctx, slot, plane):
        return self.out[slot][plane].ctypes.data

    def _ready(self, ctx, slot):
        return self.flags[slot].ctypes.data

    def pointers(self):
        return (None,) + tuple(ctypes.cast(c, ctypes.c_void_p) for c in self._cb)

    def report(self):
        return {"provider": "oracle (CPU checker)", "fills": self.fills}

    def close(self):
        pass


class PhaseCacheParams(ctypes.Structure):
    """x265hip_phase_cache_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("stride", ctypes.c_ssize_t), ("rows", ctypes.c_int), ("stride_c", ctypes.c_ssize_t), ("rows_c", ctypes.c_int),
                ("slots", ctypes.c_int)]


class PhaseCacheStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("fills", "failed", "us_upload_kernel", "us_download", "bytes_downloaded", "bytes_per_picture")]


class GpuPhaseProvider:
    """libx265hip.so's x265hip_phase_cache: the product path."""

    def __init__(self, depth, geo, slots):
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        self.L = A.lib()
        p = PhaseCacheParams(depth, geo["stride"], geo["rows"], geo["stride_c"], geo["rows_c"], slots)
        self.handle = ctypes.c_void_p()
        self.L.x265hip_phase_cache_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(PhaseCacheParams)]
        A.check(self.L.x265hip_phase_cache_create(ctypes.byref(self.handle), ctypes.byref(p)), "x265hip_phase_cache_create")
        self.L.x265hip_phase_cache_destroy.argtypes = [ctypes.c_void_p]
        self.L.x265hip_phase_cache_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(PhaseCacheStats)]

    def pointers(self):
        L = self.L
        return (self.handle, ctypes.cast(L.x265hip_phase_cache_submit, ctypes.c_void_p), ctypes.cast(L.x265hip_phase_cache_planes, ctypes.c_void_p),
                ctypes.cast(L.x265hip_phase_cache_ready, ctypes.c_void_p))

    def report(self):
        st = PhaseCacheStats()
        self.L.x265hip_phase_cache_stats(self.handle, ctypes.byref(st))
        n = max(1, st.fills)
        return {"provider": "x265hip_phase_cache (15 luma + 2 x 63 chroma phase planes per reference picture, one x265hip_phase_planes launch per plane)",
                "fills": int(st.fills), "failed": int(st.failed), "mbytes_per_picture": round(st.bytes_per_picture / 1e6, 1),
                "ms_per_picture": {"upload_and_kernels": round(st.us_upload_kernel / n / 1e3, 3), "download": round(st.us_download / n / 1e3, 3)},
                "download_gbytes_per_s": round(st.bytes_downloaded / max(1, st.us_download) / 1e3, 2)}

    def close(self):
        if self.handle:
            self.L.x265hip_phase_cache_destroy(self.handle)
            self.handle = None


def install(depth, width, height, provider="gpu", rng=32, slots=8, min_pu=8, verify=False, wait=False, lookahead=None, subpel=None, subpel_slots=6,
            surf_format=None):
    """Returns (seam library, table filler pointer, report(), close()).  Encode with lib.x265ref_encode(..., filler, ...) using
    --frame-threads 1 and --ctu 64."""
    lib = seam_lib(depth)
    geo = geometry(width, height)
    prov = GpuProvider(depth, geo, rng, slots, surf_format) if provider == "gpu" else OracleProvider(depth, geo, rng, slots)
    ctx, submit, submit_batch, surface, ready = prov.pointers()
    rc = lib.x265ref_seam_configure(ctx, submit, submit_batch, surface, ready, rng, prov.format, slots, geo["width"], geo["height"], geo["stride"],
                                    geo["margin_x"], geo["margin_y"], min_pu, int(bool(verify)) | (2 if wait else 0))
    if rc:
        raise RuntimeError(f"x265ref_seam_configure failed ({rc})")
    filler = ctypes.cast(lib.x265ref_seam_fill_table, ctypes.c_void_p)
    # the lookahead seam (CostEstimateGroup::estimateFrameCost's block loop as one provider call): "gpu" = x265hip_lowres_cost_host,
    # "oracle" = the CPU restatement (checker; GPU-less tests), None = off.  Needs --lookahead-slices 1.
    lib.x265ref_lookahead_seam_configure.argtypes = [ctypes.c_void_p] * 4
    keep = None
    if lookahead in ("gpu", "gpu+verify"):
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        ocost = ointra = None
        if lookahead == "gpu+verify":          # the oracle re-scores every triple from the same inputs; mismatches are reported and counted
            keep = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle.so"))
            ocost = ctypes.cast(getattr(keep, f"x265oracle_lowres_cost_wp_d{depth}"), ctypes.c_void_p)
            ointra = ctypes.cast(getattr(keep, f"x265oracle_lowres_intra_d{depth}"), ctypes.c_void_p)
        lib.x265ref_lookahead_seam_configure(ctypes.cast(A.lib().x265hip_lowres_cost_host, ctypes.c_void_p), ocost,
                                             ctypes.cast(A.lib().x265hip_lowres_intra_host, ctypes.c_void_p), ointra)
    elif lookahead == "oracle":
        keep = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle.so"))
        lib.x265ref_lookahead_seam_configure(None, ctypes.cast(getattr(keep, f"x265oracle_lowres_cost_wp_d{depth}"), ctypes.c_void_p),
                                             None, ctypes.cast(getattr(keep, f"x265oracle_lowres_intra_d{depth}"), ctypes.c_void_p))
    else:
        lib.x265ref_lookahead_seam_configure(None, None, None, None)

    # the sub-sample seam (MotionEstimate::subpelCompare reads precomputed phase planes): "gpu" = x265hip_phase_cache, "oracle" = CPU checker
    lib.x265ref_subpel_seam_configure.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int]
    lib.x265ref_subpel_seam_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    sub = None
    if subpel:
        sub = (GpuPhaseProvider if subpel == "gpu" else OraclePhaseProvider)(depth, geo, subpel_slots)
        sctx, ssub, spl, srd = sub.pointers()
        rc = lib.x265ref_subpel_seam_configure(sctx, ssub, spl, srd, subpel_slots, geo["stride"], geo["rows"], geo["stride_c"], geo["rows_c"],
                                               int(bool(verify)) | (2 if wait else 0))
        if rc:
            raise RuntimeError(f"x265ref_subpel_seam_configure failed ({rc})")
    else:
        lib.x265ref_subpel_seam_configure(None, None, None, None, 0, 0, 0, 0, 0, 0)

    def report():
        d = stats(lib)
        d.update(prov.report())
        d.update({"range": rng, "slots": slots, "min_pu": min_pu})
        la = (ctypes.c_uint64 * 4)()
        lib.x265ref_lookahead_seam_stats(la)
        lib.x265ref_lookahead_seam_mismatches.restype = ctypes.c_uint64
        d["lookahead_seam"] = {"provider": lookahead, "frame_cost_estimates_served": int(la[0]), "passed_to_reference_loop": int(la[1]), "failed": int(la[2]),
                               "intra_estimates_served": int(la[3]), "verify_mismatches": int(lib.x265ref_lookahead_seam_mismatches())}
        if sub:
            so = (ctypes.c_uint64 * 6)()
            lib.x265ref_subpel_seam_stats(so)
            d["subpel_seam"] = dict(zip(SUB_STAT_NAMES, [int(v) for v in so]))
            d["subpel_seam"].update(sub.report())
        return d

    def close():
        lib.x265ref_seam_disable()
        prov.close()
        if sub:
            sub.close()
    close.keep = keep            # the oracle library must outlive the encode
    return lib, filler, report, close, prov
