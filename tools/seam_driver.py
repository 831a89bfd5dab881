"""Driver of the stage-level seam (tier T3 with a batch-layer consumer): configures oracle/_ref/libx265ref<depth>_seam.so
(binding/x265hip_x265_binding.cpp - the reference-side binding) with a SAD-surface provider and hands back the table filler for
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


def seam_lib(depth, build=""):
    path = os.path.join(ROOT, "oracle", "_ref", f"libx265ref{depth}{build}_seam.so")
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


# ---------------------------------------------------------------------------------------------------------------------------------
# ROW-GRANULAR providers (round 3): x265hip_me_stream / x265hip_phase_stream and their CPU stand-ins.  The binding's producer hook
# (FrameFilter::processPostRow) feeds them reconstructed CTU rows, so they serve under any --frame-threads.
PIC_ROWS = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
PAIR_OPEN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64)
PAIR_OPEN_W = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p)
PS_OPEN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint)
PS_ROWS = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
WEIGHTED_STAT_NAMES = ("pairs_opened_on_weighted_references", "motion_estimate_calls_with_context_on_weighted_references", "lookups_served_on_weighted_references",
                       "phase_views_opened_on_weighted_references", "subpel_compares_served_from_weighted_views", "lookups_on_saturated_16_bit_entries")
CENTRES = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)
LAYOUT_RECORDS, LAYOUT_PLANES = 0, 1


def planes_ctu_bytes(rng, min_level):
    """x265hip_stream_planes_ctu_bytes (include/x265hip.h)"""
    nc = 2 * rng + 1
    return nc * 4 * ((nc + 3) // 4) * ((0 if min_level else 128) + (0 if min_level > 1 else 32) + 20)


def records_to_planes(recs, nctu, rng, min_level):
    """int32 records [nctu * nc * ng][85][4] -> the PU-major planes layout (X265HIP_STREAM_PLANES) as bytes [nctu][ctu_bytes]: uint16
    (saturating) rasters of the 8x8 / 16x16 PUs, uint32 rasters of 32x32 / 64x64 - the checker-side twin of csrc/me_stream.hip's kernel"""
    nc = 2 * rng + 1
    ng = (nc + 3) // 4
    v = recs.reshape(nctu, nc, ng, 85, 4).transpose(0, 3, 1, 2, 4).reshape(nctu, 85, nc * ng * 4)
    lo = np.minimum(v[:, (80 if min_level > 1 else 64 if min_level else 0):80], 65535).astype(np.uint16)
    hi = v[:, 80:85].astype(np.uint32)
    return np.concatenate([lo.reshape(nctu, -1).view(np.uint8), hi.reshape(nctu, -1).view(np.uint8)], axis=1)


def weight_plane(plane, depth, w):
    """primitives.weight_pp (common/pixel.cpp:518-543) sample by sample; w = (w0, round, shift, offset) with the 14 - depth correction
    already in round and shift - what reference.cpp:154 passes.  Checker-side twin of the providers' device kernels."""
    w0, rnd, shift, off = w
    val = (plane.astype(np.int32) << (14 - depth)).astype(np.int16).astype(np.int32)
    return np.clip(((w0 * val + rnd) >> shift) + off, 0, (1 << depth) - 1).astype(plane.dtype)


def read_weights(ptr, n):
    """n x265hip_weight structs at ptr -> list of (w0, round, shift, offset)"""
    raw = (ctypes.c_int * (4 * n)).from_address(ptr)
    return [tuple(int(raw[4 * i + k]) for k in range(4)) for i in range(n)]
PS_PROGRESS = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)
STREAM_STAT_NAMES = ("recon_rows_to_sad_provider", "recon_rows_refused", "recon_rows_to_phase_provider", "lookups_dropped_slot_reopened")


class StreamOracleProvider:
    """CPU stand-in for x265hip_me_stream (checker only): pictures arrive row by row, every open pair's CTU rows are searched with the
    oracle as soon as the reference rows their windows reach are there - synchronously inside the calls.  int32 records; min_level 1
    keeps the 21 PUs of 16x16 and up like the product."""

    def __init__(self, depth, geo, rng, slots, min_level=0, layout=LAYOUT_RECORDS, centre_range=0):
        import threading
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_api
        self.O, self.depth, self.geo, self.range, self.slots, self.min_level = oracle_api, depth, geo, rng, slots, min_level
        self.dt = np.uint8 if depth == 8 else np.uint16
        self.ctus_w, self.ctus_h = geo["width"] // 64, geo["height"] // 64
        nc = 2 * rng + 1
        self.nc, self.ng = nc, (nc + 3) // 4
        self.rec_words = (21 if min_level else 85) * 4
        self.layout, self.centre_range = layout, centre_range
        self.row_words = self.ctus_w * (planes_ctu_bytes(rng, min_level) // 4 if layout else nc * self.ng * self.rec_words)
        self.surf = [np.zeros(self.row_words * self.ctus_h, np.int32) for _ in range(slots)]
        self.centres = [np.zeros((self.ctus_w * self.ctus_h, 2), np.int16) for _ in range(slots)] if centre_range else None
        self.max_c = (min(centre_range, geo["margin_x"] - rng - 12), min(centre_range, geo["margin_y"] - rng - 12))
        self.flags = [np.zeros(self.ctus_h, np.int32) for _ in range(slots)]
        self.pair = [None] * slots               # dict(f, r, gen, next)
        self.gen = [0] * slots
        self.pics = {}                           # key -> dict(plane, rows)
        self.plane_elems = geo["stride"] * geo["rows"]
        self.org = geo["margin_y"] * geo["stride"] + geo["margin_x"]
        # CTU rows of the reference a CTU row's searches reach: the centre search 63 + centre_range lines, the window round a centre another
        # max |cy| + range (the product's lagRows, me_stream.hip - round-4 advisor: both ignored centre_range)
        self.lag = (63 + (max(centre_range, self.max_c[1] + rng) if centre_range else rng)) // 64
        self.format = SURF_I32
        self.bands = self.rows_in = 0
        self.lock = threading.Lock()
        self.weighted_pairs = 0
        self._cb = (PIC_ROWS(self._picture_rows), PAIR_OPEN(self._pair_open), PAIR_OPEN_W(self._pair_open_w), SURFACE(self._surface), READY(self._ready),
                    CENTRES(self._centres))

    def _lines(self, r0, n):
        g = self.geo
        y0 = 0 if r0 == 0 else g["margin_y"] + r0 * 64
        y1 = g["rows"] if r0 + n == self.ctus_h else g["margin_y"] + (r0 + n) * 64
        return y0, y1

    def _picture_rows(self, ctx, key, buf, r0, n):
        with self.lock:
            g = self.geo
            pic = self.pics.setdefault(int(key), {"plane": np.zeros(self.plane_elems, self.dt), "rows": set()})
            if len(self.pics) > 64:              # drop the oldest pictures nobody refers to
                live = {q[k] for q in self.pair if q for k in ("f", "r")} | {int(key)}
                for k in list(self.pics):
                    if k not in live and len(self.pics) > 48:
                        del self.pics[k]
            y0, y1 = self._lines(r0, n)
            es = np.dtype(self.dt).itemsize
            raw = (ctypes.c_uint8 * ((y1 - y0) * g["stride"] * es)).from_address(buf + y0 * g["stride"] * es)
            pic["plane"][y0 * g["stride"]:y1 * g["stride"]] = np.frombuffer(raw, dtype=self.dt)
            pic["rows"].update(range(r0, r0 + n))
            self.rows_in += n
            self._advance()
        return 0

    def _pair_open(self, ctx, slot, fkey, rkey):
        return self._pair_open_w(ctx, slot, fkey, rkey, None)

    def _pair_open_w(self, ctx, slot, fkey, rkey, wptr):
        with self.lock:
            self.gen[slot] += 1
            self.flags[slot][:] = 0
            self.pair[slot] = {"f": int(fkey), "r": int(rkey), "gen": self.gen[slot], "next": 0, "w": read_weights(wptr, 1)[0] if wptr else None}
            self.weighted_pairs += bool(wptr)
            self._advance()
            return self.gen[slot]

    def _advance(self):
        g = self.geo
        zero = np.zeros(self.nc, np.uint16)
        for slot, q in enumerate(self.pair):
            if not q or q["next"] >= self.ctus_h or q["f"] not in self.pics or q["r"] not in self.pics:
                continue
            pf, pr = self.pics[q["f"]], self.pics[q["r"]]
            r0 = r1 = q["next"]
            while r1 < self.ctus_h and r1 in pf["rows"] and all(k in pr["rows"] for k in range(max(0, r1 - self.lag), min(self.ctus_h, r1 + self.lag + 1))):
                r1 += 1
            if r1 == r0:
                continue
            n = r1 - r0
            off = self.org + r0 * 64 * g["stride"]
            ref_plane = pr["plane"] if q["w"] is None else weight_plane(pr["plane"], self.depth, q["w"])      # rows not there yet weight to garbage nobody reads
            if self.centre_range:
                # where each CTU's 64x64 block went (minimum SAD of +-centre_range, ties to the first in raster order), clamped: the window's centre
                zc = np.zeros(2 * self.centre_range + 1, np.uint16)
                _, best = self.O.me_fullsearch(self.depth, pf["plane"], g["stride"], off, ref_plane, g["stride"], off, g["width"], n * 64, self.centre_range,
                                               0, self.ctus_w * n, zc, zc, want_surf=False, want_best=True)
                idx = (best.reshape(-1, 85)[:, 84] & 0xffffffff).astype(np.int64)
                ncb = 2 * self.centre_range + 1
                cen = np.stack([np.clip(idx % ncb - self.centre_range, -self.max_c[0], self.max_c[0]),
                                np.clip(idx // ncb - self.centre_range, -self.max_c[1], self.max_c[1])], axis=1).astype(np.int16)
                self.centres[slot][r0 * self.ctus_w:r1 * self.ctus_w] = cen
                parts = []
                for c in range(self.ctus_w * n):           # one CTU at a time: each has its own window
                    o = off + (c // self.ctus_w) * 64 * g["stride"] + (c % self.ctus_w) * 64
                    sc, _ = self.O.me_fullsearch(self.depth, pf["plane"], g["stride"], o, ref_plane, g["stride"], o + int(cen[c, 1]) * g["stride"] + int(cen[c, 0]),
                                                 64, 64, self.range, 0, 1, zero, zero, want_surf=True, want_best=False)
                    parts.append(sc)
                surf = np.concatenate(parts)
            else:
                surf, _ = self.O.me_fullsearch(self.depth, pf["plane"], g["stride"], off, ref_plane, g["stride"], off, g["width"], n * 64, self.range,
                                               0, self.ctus_w * n, zero, zero, want_surf=True, want_best=False)
            recs = surf.reshape(-1, 85, 4)
            if self.layout:
                self.surf[slot][r0 * self.row_words:r1 * self.row_words] = records_to_planes(recs, self.ctus_w * n, self.range, self.min_level).reshape(-1).view(np.int32)
            else:
                if self.min_level:
                    recs = recs[:, 64:, :]
                self.surf[slot][r0 * self.row_words:r1 * self.row_words] = recs.reshape(-1)
            self.flags[slot][r0:r1] = q["gen"]
            q["next"] = r1
            self.bands += 1

    def _surface(self, ctx, slot):
        return self.surf[slot].ctypes.data

    def _ready(self, ctx, slot):
        return self.flags[slot].ctypes.data

    def _centres(self, ctx, slot):
        return self.centres[slot].ctypes.data if self.centres else None

    def pointers(self):
        return (None,) + tuple(ctypes.cast(c, ctypes.c_void_p) for c in self._cb)

    def report(self):
        return {"provider": "oracle, row-granular (CPU checker)", "bands": self.bands, "rows_in": self.rows_in, "weighted_pairs": self.weighted_pairs}

    def close(self):
        pass


class StreamParams(ctypes.Structure):
    """x265hip_me_stream_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int), ("stride", ctypes.c_ssize_t),
                ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int), ("range", ctypes.c_int), ("surf_format", ctypes.c_int), ("min_level", ctypes.c_int),
                ("slots", ctypes.c_int), ("pictures", ctypes.c_int), ("band_rows", ctypes.c_int), ("layout", ctypes.c_int), ("centre_range", ctypes.c_int),
                ("device_plus_1", ctypes.c_int)]


class StreamStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("pairs_opened", "pairs_completed", "bands", "rows_searched", "rows_uploaded", "failed", "stale_pairs",
                                               "us_busy", "bytes_downloaded", "bytes_uploaded", "surface_bytes", "rows_weighted", "weighted_pairs")]


class StreamGpuProvider:
    """libx265hip.so's x265hip_me_stream: the product path under frame threads."""

    def __init__(self, depth, geo, rng, slots, min_level=1, pictures=24, band_rows=0, layout=LAYOUT_RECORDS, centre_range=0, device=None):
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        self.A, self.L = A, A.lib()
        self.format = SURF_PACKED if depth == 8 else SURF_I32
        self.min_level, self.layout, self.centre_range = min_level, layout, centre_range
        p = StreamParams(depth, geo["width"], geo["height"], geo["stride"], geo["margin_x"], geo["margin_y"], rng, self.format, min_level, slots, pictures, band_rows,
                         layout, centre_range, 0 if device is None else device + 1)
        self.handle = ctypes.c_void_p()
        L = self.L
        L.x265hip_me_stream_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(StreamParams)]
        A.check(L.x265hip_me_stream_create(ctypes.byref(self.handle), ctypes.byref(p)), "x265hip_me_stream_create")
        L.x265hip_me_stream_destroy.argtypes = [ctypes.c_void_p]
        L.x265hip_me_stream_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(StreamStats)]

    def pointers(self):
        L = self.L
        return (self.handle,) + tuple(ctypes.cast(f, ctypes.c_void_p) for f in (L.x265hip_me_stream_picture_rows, L.x265hip_me_stream_pair_open,
                                                                                  L.x265hip_me_stream_pair_open_weighted, L.x265hip_me_stream_surface,
                                                                                  L.x265hip_me_stream_ready, L.x265hip_me_stream_centres))

    def report(self):
        st = StreamStats()
        self.L.x265hip_me_stream_stats(self.handle, ctypes.byref(st))
        d = {n: int(getattr(st, n)) for n, _ in StreamStats._fields_}
        d["provider"] = ("x265hip_me_stream (reconstructed CTU rows in as the reference publishes them; every open (picture, reference) pair searched "
                         "row by row behind the producer; " + ("32x32-and-up " if self.min_level > 1 else "16x16-and-up " if self.min_level else "all ") +
                         ("PU-major planes" if self.layout else "records") + " downloaded per band" +
                         (f"; windows centred on each CTU's displacement within +-{self.centre_range}" if self.centre_range else "") + ")")
        d["surface_mbytes_per_pair"] = round(st.surface_bytes / 1e6, 1)
        d["worker_busy_ms"] = round(st.us_busy / 1e3, 1)
        return d

    def close(self):
        if self.handle:
            self.L.x265hip_me_stream_destroy(self.handle)
            self.handle = None


class StreamOraclePhaseProvider:
    """CPU stand-in for x265hip_phase_stream (checker only): pictures arrive row by row under a key; a slot is a VIEW of one picture,
    optionally weighted plane by plane before the interpolation; the oracle's phase planes grow line by line as rows arrive."""

    def __init__(self, depth, geo, slots):
        import threading
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_api
        self.O, self.depth, self.geo, self.slots = oracle_api, depth, geo, slots
        self.dt = np.uint8 if depth == 8 else np.uint16
        g = geo
        self.ctu_rows = g["height"] // 64
        self.dims = [(g["rows"], g["stride"], g["margin_y"], 64, 15), (g["rows_c"], g["stride_c"], g["margin_y"] >> 1, 32, 63)]
        self.pics = {}                           # key -> dict(src [3 planes], staged, next)
        self.out = [[np.zeros((self.dims[min(k, 1)][4], self.dims[min(k, 1)][0], self.dims[min(k, 1)][1]), self.dt) for k in range(3)] for _ in range(slots)]
        self.progress = [np.zeros(2, np.uint64) for _ in range(slots)]
        self.view = [None] * slots               # dict(key, gen, w [3] or None per plane, seen, done)
        self.gen = [0] * slots
        self.opened = self.bands = self.weighted_views = 0
        self.lock = threading.Lock()
        self._cb = (PS_OPEN(self._open), PS_ROWS(self._rows), PH_PLANES(self._planes), PS_PROGRESS(self._progress))

    def _pic(self, key):
        if key not in self.pics:
            if len(self.pics) > 48:              # drop the oldest pictures no view is fed from
                live = {v["key"] for v in self.view if v} | {key}
                for k in list(self.pics):
                    if k not in live and len(self.pics) > 32:
                        del self.pics[k]
            self.pics[key] = {"src": [np.zeros((self.dims[min(k, 1)][0], self.dims[min(k, 1)][1]), self.dt) for k in range(3)], "staged": set(), "next": 0}
        return self.pics[key]

    def _open(self, ctx, slot, key, wptr, mask):
        with self.lock:
            self.gen[slot] += 1
            self.progress[slot][:] = 0
            w3 = read_weights(wptr, 3) if wptr and mask else [None] * 3
            self.view[slot] = {"key": int(key), "w": [w3[c] if (mask >> c) & 1 else None for c in range(3)], "seen": 0, "done": [8, 8]}
            self.opened += 1
            self.weighted_views += bool(wptr and mask)
            self._pic(int(key))
            self._advance()
            return self.gen[slot]

    def _rows(self, ctx, key, luma, cb, cr, r0, n):
        with self.lock:
            pic = self._pic(int(key))
            es = np.dtype(self.dt).itemsize
            for pl, ptr in enumerate((luma, cb, cr)):
                rows, stride, margin, cl, _ = self.dims[min(pl, 1)]
                y0 = 0 if r0 == 0 else margin + r0 * cl
                y1 = rows if r0 + n == self.ctu_rows else margin + (r0 + n) * cl
                raw = (ctypes.c_uint8 * ((y1 - y0) * stride * es)).from_address(ptr + y0 * stride * es)
                pic["src"][pl][y0:y1] = np.frombuffer(raw, dtype=self.dt).reshape(y1 - y0, stride)
            pic["staged"].update(range(r0, r0 + n))
            while pic["next"] in pic["staged"]:
                pic["next"] += 1
            self._advance()
        return 0

    def _advance(self):
        for slot, v in enumerate(self.view):
            if not v or v["key"] not in self.pics:
                continue
            pic = self.pics[v["key"]]
            r1 = pic["next"]
            if r1 <= v["seen"]:
                continue
            v["seen"] = r1
            for k in range(2):
                rows, stride, margin, cl, nph = self.dims[k]
                y1 = rows if r1 == self.ctu_rows else margin + r1 * cl
                b0, b1 = v["done"][k], y1 - 8
                if b1 - b0 < 8:
                    continue
                for pl in ((0,) if k == 0 else (1, 2)):
                    band = pic["src"][pl][b0 - 8:b1 + 8]
                    if v["w"][pl] is not None:
                        band = weight_plane(band, self.depth, v["w"][pl])
                    ph = self.O.phase_planes(self.depth, np.ascontiguousarray(band), stride, band.shape[0], chroma=bool(k))
                    self.out[slot][pl][:, b0:b1] = ph[:, 8:band.shape[0] - 8]
                v["done"][k] = b1
                self.progress[slot][k] = (self.gen[slot] << 32) | b1
            self.bands += 1

    def _planes(self, ctx, slot, plane):
        return self.out[slot][plane].ctypes.data

    def _progress(self, ctx, slot):
        return self.progress[slot].ctypes.data

    def pointers(self):
        return (None,) + tuple(ctypes.cast(c, ctypes.c_void_p) for c in self._cb)

    def report(self):
        return {"provider": "oracle, row-granular (CPU checker)", "views_opened": self.opened, "bands": self.bands, "weighted_views": self.weighted_views}

    def close(self):
        pass


class PhaseStreamParams(ctypes.Structure):
    """x265hip_phase_stream_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("stride", ctypes.c_ssize_t), ("rows", ctypes.c_int), ("margin_y", ctypes.c_int),
                ("stride_c", ctypes.c_ssize_t), ("rows_c", ctypes.c_int), ("margin_y_c", ctypes.c_int), ("ctu_rows", ctypes.c_int), ("slots", ctypes.c_int),
                ("pictures", ctypes.c_int), ("device_plus_1", ctypes.c_int)]


class PhaseStreamStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("opened", "completed", "bands", "failed", "us_busy", "bytes_downloaded", "bytes_uploaded", "bytes_per_picture",
                                               "weighted_views", "lines_weighted")]


class StreamGpuPhaseProvider:
    """libx265hip.so's x265hip_phase_stream: the product path under frame threads."""

    def __init__(self, depth, geo, slots, pictures=0, device=None):
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        self.L = L = A.lib()
        p = PhaseStreamParams(depth, geo["stride"], geo["rows"], geo["margin_y"], geo["stride_c"], geo["rows_c"], geo["margin_y"] >> 1, geo["height"] // 64, slots,
                              pictures, 0 if device is None else device + 1)
        self.handle = ctypes.c_void_p()
        L.x265hip_phase_stream_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(PhaseStreamParams)]
        A.check(L.x265hip_phase_stream_create(ctypes.byref(self.handle), ctypes.byref(p)), "x265hip_phase_stream_create")
        L.x265hip_phase_stream_destroy.argtypes = [ctypes.c_void_p]
        L.x265hip_phase_stream_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(PhaseStreamStats)]

    def pointers(self):
        L = self.L
        return (self.handle,) + tuple(ctypes.cast(f, ctypes.c_void_p) for f in (L.x265hip_phase_stream_view_open, L.x265hip_phase_stream_picture_rows,
                                                                                  L.x265hip_phase_stream_planes, L.x265hip_phase_stream_progress))

    def report(self):
        st = PhaseStreamStats()
        self.L.x265hip_phase_stream_stats(self.handle, ctypes.byref(st))
        d = {n: int(getattr(st, n)) for n, _ in PhaseStreamStats._fields_}
        d["provider"] = ("x265hip_phase_stream (a view per (reference picture, weights) opened by the first search that refers to it; its phase planes grow "
                         "line by line behind the reconstruction)")
        d["mbytes_per_picture"] = round(st.bytes_per_picture / 1e6, 1)
        d["worker_busy_ms"] = round(st.us_busy / 1e3, 1)
        d["download_gbytes_per_s_while_busy"] = round(st.bytes_downloaded / max(1, st.us_busy) / 1e3, 2)
        return d

    def close(self):
        if self.handle:
            self.L.x265hip_phase_stream_destroy(self.handle)
            self.handle = None


# ---------------------------------------------------------------------------------------------------------------------------------
# COST-TABLE providers (round 6): x265hip_cost_stream and its CPU stand-in - the sub-sample half of motionEstimate served as values.
CS_ROWS = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
CS_OPEN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p)
CS_TABLES = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)
CS_READY = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)
COST_STAT_NAMES = ("comparisons_served_from_records", "passed_on_other_vector_or_position", "passed_on_records_not_arrived", "passed_on_saturated_delta", "dropped_slot_reopened",
                   "motion_estimate_calls_seen", "calls_without_context", "pairs_opened", "pair_requests_without_slot", "recon_rows_to_provider", "recon_rows_refused",
                   "verify_mismatches", "pairs_on_weighted_references", "sad_typed_comparisons_served_from_records")
PRESET_SUBME = {"ultrafast": 0, "superfast": 1, "veryfast": 1, "faster": 2, "fast": 2, "medium": 2, "slow": 3, "slower": 4, "veryslow": 4, "placebo": 5}      # common/param.cpp:397-539
PRESET_SHAPES = {"slow": 1, "slower": 2, "veryslow": 2, "placebo": 2}                                                                                             # --rect from slow, --amp from slower


def cost_config(preset, opts, centre_range=57, window=8, candidates=1, slots=24, pictures=40, views=12, band_rows=8, mv_cost=True, set_subme=None, sad_costs=False):
    """The cost-table service's parameters for an encode: the refinement's position set and the chroma flag follow --subme (bChromaSATD: subme > 2,
    motion.cpp:212), the PU list follows --rect / --amp."""
    o = dict((k, v) for k, v in opts)
    subme = int(o.get("subme", PRESET_SUBME.get(preset, 2)))
    shapes = PRESET_SHAPES.get(preset, 0)
    if "rect" in o: shapes = max(shapes, 1)
    if "amp" in o: shapes = 2
    if "no-rect" in o: shapes = 0
    elif "no-amp" in o: shapes = min(shapes, 1)
    # set_subme: the records may hold the position set of a HIGHER workload row than the encode's (a superset: refinements that start from a fractional predictor
    # leave the --subme 3 set of 49 positions more often than --subme 4's 85); the chroma flag stays the encode's
    return dict(centre_range=centre_range, window=window, candidates=candidates, shapes=shapes, subme=max(subme, set_subme or 0), host_subme=subme, chroma=int(subme > 2), slots=slots, pictures=pictures, views=views,
                band_rows=band_rows, mv_cost=bool(mv_cost), sad_costs=int(bool(sad_costs)))


class StreamGpuCostProvider:
    """libx265hip.so's x265hip_cost_stream: the product path."""

    def __init__(self, depth, geo, cfg, device=None):
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        self.A, self.L, self.cfg = A, A.lib(), cfg
        L = self.L
        p = A.CostStreamParams(depth, geo["width"], geo["height"], geo["stride"], geo["margin_x"], geo["margin_y"], geo["stride_c"], geo["margin_y"] >> 1,
                               cfg["centre_range"], cfg["window"], cfg["candidates"], cfg["shapes"], cfg["subme"], cfg["chroma"], cfg.get("sad_costs", 0), cfg["slots"], cfg["pictures"], cfg["views"],
                               cfg["band_rows"], 0 if device is None else device + 1)
        self.handle = ctypes.c_void_p()
        L.x265hip_cost_stream_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(A.CostStreamParams)]
        A.check(L.x265hip_cost_stream_create(ctypes.byref(self.handle), ctypes.byref(p)), "x265hip_cost_stream_create")
        L.x265hip_cost_stream_destroy.argtypes = [ctypes.c_void_p]
        L.x265hip_cost_stream_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(A.CostStreamStats)]
        self.rects, self.positions = A.cost_pu_list(cfg["shapes"]), A.cost_positions(cfg["subme"])
        self.record_bytes = A.cost_record_bytes(cfg["subme"], cfg.get("sad_costs", 0))
        self.ctu_bytes = A.cost_ctu_bytes(cfg["subme"], cfg["shapes"], cfg["candidates"], cfg.get("sad_costs", 0))

    def pointers(self):
        L = self.L
        return (self.handle,) + tuple(ctypes.cast(f, ctypes.c_void_p) for f in (L.x265hip_cost_stream_picture_rows, L.x265hip_cost_stream_pair_open,
                                                                                  L.x265hip_cost_stream_tables, L.x265hip_cost_stream_ready))

    def report(self):
        st = self.A.CostStreamStats()
        self.L.x265hip_cost_stream_stats(self.handle, ctypes.byref(st))
        d = {n: int(getattr(st, n)) for n, _ in st._fields_}
        d["provider"] = "x265hip_cost_stream (records of sub-sample SATD costs around each PU's best integer vectors; views and phase planes stay on the device)"
        d["worker_busy_ms"] = round(st.us_busy / 1e3, 1)
        d["config"] = dict(self.cfg)
        return d

    def close(self):
        if self.handle:
            self.L.x265hip_cost_stream_destroy(self.handle)
            self.handle = None


class StreamOracleCostProvider:
    """CPU stand-in for x265hip_cost_stream (checker only): pictures arrive by key (a reconstructed one row by row), a pair's CTU row is computed - the
    oracle's centre search, centred SAD rasters, candidates and tables (oracle/x265_oracle_pipeline8.c: the reference's own subpelCompare route per value) -
    once the reference rows <= r + 2 have arrived."""

    def __init__(self, depth, geo, cfg):
        import threading
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_api
        self.O, self.depth, self.geo, self.cfg = oracle_api, depth, geo, cfg
        self.dt = np.uint8 if depth == 8 else np.uint16
        g = geo
        self.ctus_w, self.ctu_rows = g["width"] // 64, g["height"] // 64
        self.nctu = self.ctus_w * self.ctu_rows
        self.dims = [(g["rows"], g["stride"], g["margin_y"], 64), (g["rows_c"], g["stride_c"], g["margin_y"] >> 1, 32)]
        self.rects, self.positions = oracle_api.cost_pu_list(cfg["shapes"])[:, :4].copy(), oracle_api.cost_positions(cfg["subme"])
        self.record_bytes = oracle_api.cost_record_bytes(cfg["subme"], cfg.get("sad_costs", 0))
        self.ctu_bytes = self.record_bytes * len(self.rects) * cfg["candidates"]
        self.pics = {}
        self.tables = [np.zeros(self.nctu * self.ctu_bytes, np.uint8) for _ in range(cfg["slots"])]
        self.flags = [np.zeros(self.ctu_rows, np.int32) for _ in range(cfg["slots"])]
        self.pair = [None] * cfg["slots"]
        self.gen = [0] * cfg["slots"]
        self.opened = self.rows_served = self.weighted = 0
        mx = g["margin_x"] - cfg["window"] - 12
        my = min(g["margin_y"], 2 * (g["margin_y"] >> 1)) - cfg["window"] - 20
        self.max_c = (min(mx, cfg["centre_range"]), min(my, cfg["centre_range"])) if cfg["centre_range"] else (mx, my)
        self.max_down = min(self.max_c[1], 42 - cfg["window"])          # a row is computed from reference rows <= r + 1: x265hip_cost_stream_create
        self.lock = threading.Lock()
        self._cb = (CS_ROWS(self._rows), CS_OPEN(self._open), CS_TABLES(self._tables), CS_READY(self._ready))

    def _pic(self, key):
        if key not in self.pics:
            if len(self.pics) > 48:
                live = {k for pr in self.pair if pr for k in (pr["fkey"], pr["rkey"])} | {key}
                for k in list(self.pics):
                    if k not in live and len(self.pics) > 32:
                        del self.pics[k]
            self.pics[key] = {"src": [np.zeros((self.dims[min(k, 1)][0], self.dims[min(k, 1)][1]), self.dt) for k in range(3)], "staged": set(), "next": 0}
        return self.pics[key]

    def _rows(self, ctx, key, luma, cb, cr, r0, n):
        with self.lock:
            pic = self._pic(int(key))
            es = np.dtype(self.dt).itemsize
            for pl, ptr in enumerate((luma, cb, cr)):
                rows, stride, margin, cl = self.dims[min(pl, 1)]
                y0 = 0 if r0 == 0 else margin + r0 * cl
                y1 = rows if r0 + n == self.ctu_rows else margin + (r0 + n) * cl
                raw = (ctypes.c_uint8 * ((y1 - y0) * stride * es)).from_address(ptr + y0 * stride * es)
                pic["src"][pl][y0:y1] = np.frombuffer(raw, dtype=self.dt).reshape(y1 - y0, stride)
            pic["staged"].update(range(r0, r0 + n))
            while pic["next"] in pic["staged"]:
                pic["next"] += 1
            self._advance()
        return 0

    def _open(self, ctx, slot, fkey, rkey, wptr, mask, cptr):
        with self.lock:
            self.gen[slot] += 1
            self.flags[slot][:] = 0
            w3 = read_weights(wptr, 3) if wptr and mask else [None] * 3
            n = 2 * self.cfg["window"] + 1
            mvc = np.frombuffer((ctypes.c_uint16 * n).from_address(cptr), dtype=np.uint16).copy() if cptr else None
            self.pair[slot] = {"fkey": int(fkey), "rkey": int(rkey), "w": [w3[c] if (mask >> c) & 1 else None for c in range(3)], "next": 0, "mv_cost": mvc}
            self.opened += 1
            self.weighted += bool(wptr and mask)
            self._pic(int(fkey)); self._pic(int(rkey))
            self._advance()
            return self.gen[slot]

    def _advance(self):
        O, g, c = self.O, self.geo, self.cfg
        for slot, pr in enumerate(self.pair):
            if not pr or pr["next"] >= self.ctu_rows or pr["fkey"] not in self.pics or pr["rkey"] not in self.pics:
                continue
            pf, rf = self.pics[pr["fkey"]], self.pics[pr["rkey"]]
            r1 = pr["next"] - 1
            while r1 + 1 < self.ctu_rows and pf["next"] > r1 + 1 and rf["next"] >= min(self.ctu_rows, r1 + 3):
                r1 += 1
            if r1 < pr["next"]:
                continue
            r0 = pr["next"]
            ref = [rf["src"][pl] if pr["w"][pl] is None else weight_plane(rf["src"][pl], self.depth, pr["w"][pl]) for pl in range(3)]
            fy, ry = pf["src"][0].reshape(-1), np.ascontiguousarray(ref[0]).reshape(-1)
            org = g["margin_y"] * g["stride"] + g["margin_x"]
            b, e = r0 * self.ctus_w, (r1 + 1) * self.ctus_w
            centres = np.zeros((self.nctu, 2), np.int16)
            if c["centre_range"]:
                zero = np.zeros(2 * c["centre_range"] + 1, np.uint16)
                _, best = O.me_fullsearch(self.depth, fy, g["stride"], org, ry, g["stride"], org, g["width"], g["height"], c["centre_range"], b, e, zero, zero, want_surf=False, want_best=True)
                idx = (best.reshape(self.nctu, 85)[b:e, 84] & np.uint64(0xffffffff)).astype(np.int64)
                ncb = 2 * c["centre_range"] + 1
                centres[b:e, 0] = np.clip(idx % ncb - c["centre_range"], -self.max_c[0], self.max_c[0])
                centres[b:e, 1] = np.clip(idx // ncb - c["centre_range"], -self.max_c[1], self.max_down)
            w = c["window"]
            rb = int(np.abs(centres[b:e]).max()) + w
            zero = np.zeros(2 * rb + 1, np.uint16)
            big, _ = O.me_fullsearch(self.depth, fy, g["stride"], org, ry, g["stride"], org, g["width"], g["height"], rb, b, e, zero, zero, want_surf=True, want_best=False)
            ncb, ngb = 2 * rb + 1, (2 * rb + 4) // 4
            big = big.reshape(self.nctu, ncb, ngb, 85, 4)[b:e].transpose(0, 1, 2, 4, 3).reshape(e - b, ncb, ngb * 4, 85)
            nc, ng = 2 * w + 1, (2 * w + 4) // 4
            surf = np.zeros((e - b, nc, ng * 4, 85), np.int32)
            for i in range(e - b):
                y0, x0 = int(centres[b + i, 1]) - w + rb, int(centres[b + i, 0]) - w + rb
                surf[i, :, :nc] = big[i, y0:y0 + nc, x0:x0 + nc]
            surf = np.ascontiguousarray(surf.reshape(e - b, nc, ng, 4, 85).transpose(0, 1, 2, 4, 3))
            cand = O.cost_candidates(surf, centres[b:e], e - b, w, c["shapes"], c["candidates"], depth=self.depth, mv_cost=pr["mv_cost"])
            tab = O.cost_tables(self.depth, [pf["src"][0], pf["src"][1], pf["src"][2]], ref, g["stride"], g["stride_c"], g["margin_x"], g["margin_y"], g["margin_y"] >> 1,
                                g["width"], r0, r1 - r0 + 1, c["shapes"], c["candidates"], c["subme"], c["chroma"], cand, sad_costs=c.get("sad_costs", 0))
            self.tables[slot][b * self.ctu_bytes:e * self.ctu_bytes] = tab.reshape(-1)
            self.flags[slot][r0:r1 + 1] = self.gen[slot]
            self.rows_served += r1 - r0 + 1
            pr["next"] = r1 + 1

    def _tables(self, ctx, slot):
        return self.tables[slot].ctypes.data

    def _ready(self, ctx, slot):
        return self.flags[slot].ctypes.data

    def pointers(self):
        return (None,) + tuple(ctypes.cast(c, ctypes.c_void_p) for c in self._cb)

    def report(self):
        return {"provider": "oracle, row-granular (CPU checker)", "pairs_opened": self.opened, "rows_served": self.rows_served, "weighted_pairs": self.weighted, "config": dict(self.cfg)}

    def close(self):
        pass


class MultiDeviceStreamProvider:
    """One x265hip_me_stream instance PER GPU behind the binding's single provider interface - the mapping a host that spreads its frame encoders
    over the GPUs of a node would use (encoder/encoder.cpp:304-321: frame encoder i -> pool i % numPools): slot s of the binding lives on instance
    s % n (its local slot s // n), every reconstructed CTU row is handed to EVERY instance (the one-to-many hand-over of SURVEY 8(e): any instance's
    pairs may refer to the picture).  devices = the device index of each instance; naming one device twice runs the same mapping on a one-GPU box."""

    def __init__(self, depth, geo, rng, slots, min_level, pictures, band_rows, layout, centre_range, devices):
        self.n = len(devices)
        per = -(-slots // self.n)
        self.inst = [StreamGpuProvider(depth, geo, rng, per, min_level, pictures, band_rows, layout, centre_range, device=d) for d in devices]
        self.format = self.inst[0].format
        L = self.L = self.inst[0].L
        L.x265hip_me_stream_picture_rows.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.x265hip_me_stream_pair_open.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]
        L.x265hip_me_stream_pair_open_weighted.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        for f in (L.x265hip_me_stream_surface, L.x265hip_me_stream_ready, L.x265hip_me_stream_centres):
            f.restype, f.argtypes = ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]
        self._cb = (PIC_ROWS(self._rows), PAIR_OPEN(self._open), PAIR_OPEN_W(self._open_w), SURFACE(self._surface), READY(self._ready), CENTRES(self._centres))

    def _at(self, slot):
        return self.inst[slot % self.n].handle, slot // self.n

    def _rows(self, ctx, key, buf, r0, n):
        rc = 0
        for i in self.inst:
            rc = rc or self.L.x265hip_me_stream_picture_rows(i.handle, key, buf, r0, n)
        return rc

    def _open(self, ctx, slot, fkey, rkey):
        h, s = self._at(slot)
        return self.L.x265hip_me_stream_pair_open(h, s, fkey, rkey)

    def _open_w(self, ctx, slot, fkey, rkey, wptr):
        h, s = self._at(slot)
        return self.L.x265hip_me_stream_pair_open_weighted(h, s, fkey, rkey, wptr)

    def _surface(self, ctx, slot):
        return self.L.x265hip_me_stream_surface(*self._at(slot))

    def _ready(self, ctx, slot):
        return self.L.x265hip_me_stream_ready(*self._at(slot))

    def _centres(self, ctx, slot):
        return self.L.x265hip_me_stream_centres(*self._at(slot))

    def pointers(self):
        return (None,) + tuple(ctypes.cast(f, ctypes.c_void_p) for f in self._cb)

    def report(self):
        reps = [i.report() for i in self.inst]
        d = {k: (sum(r[k] for r in reps) if isinstance(reps[0][k], int) and k != "surface_bytes" else reps[0][k]) for k in reps[0]}
        d["instances"] = [{k: r[k] for k in ("pairs_opened", "pairs_completed", "rows_searched", "rows_uploaded", "failed")} for r in reps]
        return d

    def close(self):
        for i in self.inst:
            i.close()


class MultiDevicePhaseProvider:
    """The same mapping for x265hip_phase_stream: view slot s on instance s % n, every reconstructed row to every instance."""

    def __init__(self, depth, geo, slots, devices):
        self.n = len(devices)
        per = -(-slots // self.n)
        self.inst = [StreamGpuPhaseProvider(depth, geo, per, device=d) for d in devices]
        L = self.L = self.inst[0].L
        L.x265hip_phase_stream_view_open.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint]
        L.x265hip_phase_stream_picture_rows.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.x265hip_phase_stream_planes.restype, L.x265hip_phase_stream_planes.argtypes = ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.x265hip_phase_stream_progress.restype, L.x265hip_phase_stream_progress.argtypes = ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]
        self._cb = (PS_OPEN(self._open), PS_ROWS(self._rows), PH_PLANES(self._planes), PS_PROGRESS(self._progress))

    def _open(self, ctx, slot, key, wptr, mask):
        return self.L.x265hip_phase_stream_view_open(self.inst[slot % self.n].handle, slot // self.n, key, wptr, mask)

    def _rows(self, ctx, key, y, cb, cr, r0, n):
        rc = 0
        for i in self.inst:
            rc = rc or self.L.x265hip_phase_stream_picture_rows(i.handle, key, y, cb, cr, r0, n)
        return rc

    def _planes(self, ctx, slot, plane):
        return self.L.x265hip_phase_stream_planes(self.inst[slot % self.n].handle, slot // self.n, plane)

    def _progress(self, ctx, slot):
        return self.L.x265hip_phase_stream_progress(self.inst[slot % self.n].handle, slot // self.n)

    def pointers(self):
        return (None,) + tuple(ctypes.cast(f, ctypes.c_void_p) for f in self._cb)

    def report(self):
        reps = [i.report() for i in self.inst]
        d = {k: (sum(r[k] for r in reps) if isinstance(reps[0][k], int) and k != "bytes_per_picture" else reps[0][k]) for k in reps[0]}
        d["instances"] = [{k: r[k] for k in ("opened", "completed", "bands", "failed")} for r in reps]
        return d

    def close(self):
        for i in self.inst:
            i.close()


def install(depth, width, height, provider="gpu", rng=32, slots=8, min_pu=8, verify=False, wait=False, lookahead=None, subpel=None, subpel_slots=6,
            surf_format=None, streamed=False, min_level=0, pictures=24, band_rows=0, weighted=True, layout=LAYOUT_RECORDS, centre_range=0, lookahead_min_blocks=0, min_ctus=0, build="", aq=None, aq_min_blocks=0,
            weight_analyse=None, weight_min_blocks=0, split_rest=False, devices=None, hit_rate_gate=(2000000, 50), cost=None, cost_cfg=None):
    """devices: [device index, ...] = one instance of each row-granular service PER entry (MultiDeviceStreamProvider / MultiDevicePhaseProvider).
    Returns (seam library, table filler pointer, report(), close()).  Encode with lib.x265ref_encode(..., filler, ...) and --ctu 64;
    the picture-granular providers need --frame-threads 1, streamed=True (row-granular providers) serves under any --frame-threads."""
    lib = seam_lib(depth, build)
    geo = geometry(width, height)
    # split_rest: whatever the services do not answer (partitions below min_pu, searches without a context, pictures below the size gates) uses the host-only
    # control's split sad_x3 / sad_x4 too (x265ref_split_fill_table) - an encode with the seams then differs from the control table by the services alone
    if split_rest:
        os.environ["X265REF_SEAM_SPLIT_REST"] = "1"
    else:
        os.environ.pop("X265REF_SEAM_SPLIT_REST", None)
    # the binding's own size gate of the two search seams (1000 CTUs: serve from 4K up) unless the caller names a threshold; tests on small pictures pass 0
    lib.x265ref_seam_min_ctus.argtypes = [ctypes.c_int]
    lib.x265ref_seam_min_ctus(1000 if min_ctus is None else min_ctus)
    # the SAD seam's hit-rate gate (window in lookups, percent): below that share of served lookups no new pairs are opened until a probe picture hits again
    lib.x265ref_seam_hit_rate_gate.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    lib.x265ref_seam_hit_rate_gate(*(hit_rate_gate or (0, 50)), None)
    # measurement mode (X265REF_PREDICT_PROBE=1): where does the host's integer search end, ranked by the window's SADs alone?  
    lib.x265ref_predict_probe.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    lib.x265ref_predict_probe(1 if os.environ.get("X265REF_PREDICT_PROBE") == "1" else 0, None)
    if streamed:
        prov = (MultiDeviceStreamProvider(depth, geo, rng, slots, min_level, pictures, band_rows, layout, centre_range, devices) if provider == "gpu" and devices
                else StreamGpuProvider(depth, geo, rng, slots, min_level, pictures, band_rows, layout, centre_range) if provider == "gpu"
                else StreamOracleProvider(depth, geo, rng, slots, min_level, layout, centre_range))
        ctx, pic_rows, pair_open, pair_open_w, surface, ready, centres = prov.pointers()
        lib.x265ref_seam_configure_streamed.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_ssize_t] + [ctypes.c_int] * 4
        rc = lib.x265ref_seam_configure_streamed(ctx, pic_rows, pair_open, pair_open_w if weighted else None, surface, ready, centres if centre_range else None, layout,
                                                 rng, prov.format, min_level, slots, geo["width"], geo["height"],
                                                 geo["stride"], geo["margin_x"], geo["margin_y"], min_pu, int(bool(verify)) | (2 if wait else 0))
    else:
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
    lib.x265ref_lookahead_seam_min_blocks.argtypes = [ctypes.c_int]
    lib.x265ref_lookahead_seam_min_blocks.restype = ctypes.c_uint64
    # the binding's own gate (16384 lowres blocks: serve from 4K up) unless the caller names a threshold; tests on small pictures pass 0
    lib.x265ref_lookahead_seam_min_blocks(16384 if lookahead_min_blocks is None else lookahead_min_blocks)
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

    # the adaptive-quantisation seam (LookaheadTLD::calcAdaptiveQuantFrame as one provider call per source picture): "gpu" = x265hip_aq_frame_host,
    # "oracle" = the CPU restatement, None = off; verify = the reference's own function runs after every served picture, arrays compared bit for bit.
    # aq_min_blocks None = 16384 quantisation groups (serve from 4K up at qg 16, like the lookahead seam's gate)
    lib.x265ref_aq_seam_configure.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    if aq == "gpu":
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        lib.x265ref_aq_seam_configure(ctypes.cast(A.lib().x265hip_aq_frame_host, ctypes.c_void_p), None, int(bool(verify)), 16384 if aq_min_blocks is None else aq_min_blocks)
    elif aq == "oracle":
        keep_aq = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle.so"))
        lib.x265ref_aq_seam_configure(None, ctypes.cast(getattr(keep_aq, f"x265oracle_aq_frame_d{depth}"), ctypes.c_void_p), int(bool(verify)),
                                      16384 if aq_min_blocks is None else aq_min_blocks)
    else:
        lib.x265ref_aq_seam_configure(None, None, 0, 0)

    # the weightAnalyse seam (the frame encoder's weighted-prediction analysis as one provider call per P / B slice): "gpu" =
    # x265hip_weight_analyse_host, "oracle" = the CPU restatement, None = off; verify = the reference's own function runs after every served slice
    lib.x265ref_weight_seam_configure.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    if weight_analyse == "gpu":
        A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
        lib.x265ref_weight_seam_configure(ctypes.cast(A.lib().x265hip_weight_analyse_host, ctypes.c_void_p), None, int(bool(verify)),
                                          16384 if weight_min_blocks is None else weight_min_blocks)
    elif weight_analyse == "oracle":
        keep_wa = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libx265oracle.so"))
        lib.x265ref_weight_seam_configure(None, ctypes.cast(getattr(keep_wa, f"x265oracle_weight_analyse_d{depth}"), ctypes.c_void_p), int(bool(verify)),
                                          16384 if weight_min_blocks is None else weight_min_blocks)
    else:
        lib.x265ref_weight_seam_configure(None, None, 0, 0)

    # the sub-sample seam (MotionEstimate::subpelCompare reads precomputed phase planes): "gpu" = x265hip_phase_cache, "oracle" = CPU checker
    lib.x265ref_subpel_seam_configure.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int]
    lib.x265ref_subpel_seam_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    sub = None
    if subpel and streamed:
        sub = (MultiDevicePhaseProvider(depth, geo, subpel_slots, devices) if subpel == "gpu" and devices
               else (StreamGpuPhaseProvider if subpel == "gpu" else StreamOraclePhaseProvider)(depth, geo, subpel_slots))
        sctx, sopen, srows, spl, sprog = sub.pointers()
        lib.x265ref_subpel_seam_configure_streamed.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_ssize_t,
                                                                                       ctypes.c_int, ctypes.c_int, ctypes.c_int]
        rc = lib.x265ref_subpel_seam_configure_streamed(sctx, sopen, srows, spl, sprog, subpel_slots, geo["stride"], geo["rows"], geo["stride_c"], geo["rows_c"],
                                                        geo["height"] // 64, int(bool(verify)) | (2 if wait else 0))
        if rc:
            raise RuntimeError(f"x265ref_subpel_seam_configure_streamed failed ({rc})")
    elif subpel:
        sub = (GpuPhaseProvider if subpel == "gpu" else OraclePhaseProvider)(depth, geo, subpel_slots)
        sctx, ssub, spl, srd = sub.pointers()
        rc = lib.x265ref_subpel_seam_configure(sctx, ssub, spl, srd, subpel_slots, geo["stride"], geo["rows"], geo["stride_c"], geo["rows_c"],
                                               int(bool(verify)) | (2 if wait else 0))
        if rc:
            raise RuntimeError(f"x265ref_subpel_seam_configure failed ({rc})")
    else:
        lib.x265ref_subpel_seam_configure(None, None, None, None, 0, 0, 0, 0, 0, 0)

    # the cost-table seam (MotionEstimate::subpelCompare's SATD comparisons answered from x265hip_cost_stream's records): "gpu" = the service, "oracle" = CPU checker;
    # cost_cfg = cost_config(preset, opts, ...) - the position set and the chroma flag must be the encode's --subme
    lib.x265ref_cost_seam_configure.argtypes = ([ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_ssize_t] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_int])
    lib.x265ref_cost_seam_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    cst = None
    if cost:
        cfg = dict(cost_cfg or cost_config("slow", []))
        cst = StreamGpuCostProvider(depth, geo, cfg) if cost == "gpu" else StreamOracleCostProvider(depth, geo, cfg)
        cctx, crows, copen, ctab, crdy = cst.pointers()
        rects = np.ascontiguousarray(cst.rects, np.int32)
        posn = np.ascontiguousarray(cst.positions, np.int8)
        import sys as _sys
        _sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_api as _O                     # the position lists only (host arithmetic); the GPU provider's own list is what the records are laid out by
        have = {tuple(p) for p in posn.tolist()}
        cover = sum(1 << sm for sm in range(8) if {tuple(p) for p in _O.cost_positions(sm).tolist()} <= have)
        rc = lib.x265ref_cost_seam_configure(cctx, crows, copen, ctab, crdy, cfg["slots"], geo["width"], geo["height"], geo["stride"], geo["stride_c"], geo["margin_x"], geo["margin_y"],
                                             cfg["candidates"], cfg["subme"], cfg["chroma"], rects.ctypes.data, len(rects), posn.ctypes.data, len(posn), cst.record_bytes, cst.ctu_bytes,
                                             cfg["window"], cover, ((8 + 2 * len(posn) + 3) & ~3) if cfg.get("sad_costs") else 0, int(bool(verify)) | (2 if wait else 0) | (4 if min_ctus == 0 else 0) | (0 if cfg.get("mv_cost", True) else 8))
        if rc:
            raise RuntimeError(f"x265ref_cost_seam_configure failed ({rc})")
    else:
        lib.x265ref_cost_seam_configure(None, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0)

    def report():
        d = stats(lib)
        d.update(prov.report())
        if cst:
            co = (ctypes.c_uint64 * 14)()
            lib.x265ref_cost_seam_stats(co)
            d["cost_seam"] = dict(zip(COST_STAT_NAMES, [int(v) for v in co]))
            got = d["cost_seam"]["comparisons_served_from_records"] + d["cost_seam"]["sad_typed_comparisons_served_from_records"]
            asked = got + sum(d["cost_seam"][k] for k in COST_STAT_NAMES[1:5])
            d["cost_seam"]["served_share_of_satd_comparisons_with_context"] = round(got / asked, 4) if asked else None
            d["cost_seam"].update(cst.report())
        d.update({"range": rng, "slots": slots, "min_pu": min_pu, "row_granular": bool(streamed), "layout": "planes" if layout else "records", "centre_range": centre_range,
                  "search_seams_left_off_by_the_size_gate": bool(lib.x265ref_seam_min_ctus(-1))})
        gate = (ctypes.c_uint64 * 4)()
        lib.x265ref_seam_hit_rate_gate(-1, -1, gate)
        d["hit_rate_gate"] = {"searches_left_to_the_host_while_closed": int(gate[0]), "times_closed": int(gate[1]), "closed_at_the_end": bool(gate[2]), "window_lookups": int(gate[3])}
        if os.environ.get("X265REF_PREDICT_PROBE") == "1":
            pp = (ctypes.c_uint64 * 7)()
            lib.x265ref_predict_probe(-1, pp)
            inside = max(1, int(pp[0]) - int(pp[1]) - int(pp[2]))
            d["integer_vector_predictor_probe"] = {"refinements": int(pp[0]), "without_sad_context": int(pp[1]), "ended_outside_the_window": int(pp[2]),
                                                   "ended_on_the_sad_minimum": int(pp[3]), "among_2_smallest": int(pp[4]), "among_4": int(pp[5]), "among_8": int(pp[6]),
                                                   "share_top1_of_inside": round(int(pp[3]) / inside, 4), "share_top4_of_inside": round(int(pp[5]) / inside, 4),
                                                   "share_top1_of_all": round(int(pp[3]) / max(1, int(pp[0])), 4)}
        if streamed:
            so4 = (ctypes.c_uint64 * 4)()
            lib.x265ref_seam_stream_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
            lib.x265ref_seam_stream_stats(so4)
            d["row_stream"] = dict(zip(STREAM_STAT_NAMES, [int(v) for v in so4]))
            so5 = (ctypes.c_uint64 * 6)()
            lib.x265ref_seam_weighted_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
            lib.x265ref_seam_weighted_stats(so5)
            d["weighted_references"] = dict(zip(WEIGHTED_STAT_NAMES, [int(v) for v in so5]))
        la = (ctypes.c_uint64 * 4)()
        lib.x265ref_lookahead_seam_stats(la)
        lib.x265ref_lookahead_seam_mismatches.restype = ctypes.c_uint64
        d["lookahead_seam"] = {"provider": lookahead, "frame_cost_estimates_served": int(la[0]), "passed_to_reference_loop": int(la[1]), "failed": int(la[2]),
                               "intra_estimates_served": int(la[3]), "verify_mismatches": int(lib.x265ref_lookahead_seam_mismatches()),
                               "left_to_the_reference_by_the_size_gate": int(lib.x265ref_lookahead_seam_min_blocks(-1))}
        if lookahead == "gpu":
            try:          # process-wide launch counts of x265hip_lowres_cost: { dependency-free flat launches (no list searched again), one-workgroup walks, split walks }
                A_ = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
                cnt = (ctypes.c_uint64 * 3)()
                A_.lib().x265hip_lowres_cost_launch_counts(cnt)
                d["lookahead_seam"]["launches_flat_walk_split"] = [int(cnt[0]), int(cnt[1]), int(cnt[2])]
            except (AttributeError, OSError):
                pass
        aqs = (ctypes.c_uint64 * 5)()
        lib.x265ref_aq_seam_stats(aqs)
        d["aq_seam"] = {"provider": aq, "pictures_served": int(aqs[0]), "passed_to_reference_loop": int(aqs[1]), "failed": int(aqs[2]), "verify_mismatches": int(aqs[3]),
                        "left_to_the_reference_by_the_size_gate": int(aqs[4])}
        was = (ctypes.c_uint64 * 6)()
        lib.x265ref_weight_seam_stats(was)
        d["weight_analyse_seam"] = {"provider": weight_analyse, "slices_served": int(was[0]), "passed_to_reference_loop": int(was[1]), "failed": int(was[2]),
                                    "verify_mismatches": int(was[3]), "left_to_the_reference_by_the_size_gate": int(was[4]), "served_slices_with_a_weight": int(was[5])}
        if sub:
            so = (ctypes.c_uint64 * 6)()
            lib.x265ref_subpel_seam_stats(so)
            d["subpel_seam"] = dict(zip(SUB_STAT_NAMES, [int(v) for v in so]))
            d["subpel_seam"].update(sub.report())
        return d

    def close():
        lib.x265ref_seam_disable()
        lib.x265ref_aq_seam_configure(None, None, 0, 0)
        lib.x265ref_weight_seam_configure(None, None, 0, 0)
        prov.close()
        if sub:
            sub.close()
        if cst:
            cst.close()
    close.keep = keep            # the oracle library must outlive the encode
    return lib, filler, report, close, prov
