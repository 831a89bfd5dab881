"""ctypes binding of libx265hip.so (the C ABI declared in include/x265hip.h).

PyTorch is used only as plumbing: device memory (tensors), streams and torch.distributed.
Every wrapper passes raw device pointers + the current torch stream through the C ABI - the
same entry points a C++ host (x265 itself) would call.  There is no CPU fallback: if the
library is missing or no gfx950 device is present, calls raise.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libx265hip.so")

CMP_SAD, CMP_SATD, CMP_SA8D, CMP_SSE_PP, CMP_PSY_COST = range(5)

_lib = None


class X265HipError(RuntimeError):
    pass


class MEParams(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int), ("range", ctypes.c_int),
        ("fenc", ctypes.c_void_p), ("fenc_stride", ctypes.c_ssize_t),
        ("fref", ctypes.c_void_p), ("fref_stride", ctypes.c_ssize_t),
        ("surf", ctypes.c_void_p), ("best", ctypes.c_void_p),
        ("cost_x", ctypes.c_void_p), ("cost_y", ctypes.c_void_p),
        ("surf_format", ctypes.c_int),
        ("centres", ctypes.c_void_p),
    ]


TU_INTRA_SLICE, TU_SIGN_HIDE = 1, 2        # X265HIP_TU_* flag bits of the TU stages' intra_slice field
SURF_I32, SURF_PACKED, SURF_PACKED_T, SURF_PACKED_B = 0, 1, 2, 3
SURF_GROUP_BYTES_I32, SURF_GROUP_BYTES_PACKED = 1360, 720


class SubpelParams(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int), ("range", ctypes.c_int),
        ("subme", ctypes.c_int),
        ("fenc", ctypes.c_void_p), ("fenc_stride", ctypes.c_ssize_t),
        ("fref", ctypes.c_void_p), ("fref_stride", ctypes.c_ssize_t),
        ("best_in", ctypes.c_void_p), ("cost_q", ctypes.c_void_p), ("qoff", ctypes.c_int), ("out", ctypes.c_void_p),
        ("phase_planes", ctypes.c_void_p), ("phase_plane_samples", ctypes.c_ssize_t),
    ]


class LowresInitParams(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_int), ("src", ctypes.c_void_p), ("src_stride", ctypes.c_ssize_t),
                ("plane", ctypes.c_void_p * 4), ("stride", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("lines", ctypes.c_int), ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int)]


class LowresIntraParams(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_int), ("plane", ctypes.c_void_p), ("stride", ctypes.c_ssize_t),
                ("width_in_cu", ctypes.c_int), ("height_in_cu", ctypes.c_int), ("intra_penalty", ctypes.c_int),
                ("intra_cost", ctypes.c_void_p), ("intra_mode", ctypes.c_void_p), ("lowres_costs", ctypes.c_void_p)]


class LowresCostPair(ctypes.Structure):
    """x265hip_lowres_cost_pair (include/x265hip.h)."""
    _fields_ = [("cur", ctypes.c_void_p), ("ref", ctypes.c_void_p * 4), ("ref1", ctypes.c_void_p * 4),
                ("intra_cost", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p),
                ("mvs", ctypes.c_void_p), ("mv_costs", ctypes.c_void_p), ("mvs1", ctypes.c_void_p), ("mv_costs1", ctypes.c_void_p),
                ("do_search", ctypes.c_int32 * 2),
                ("lowres_costs", ctypes.c_void_p), ("row_satds", ctypes.c_void_p), ("frame", ctypes.c_void_p),
                ("ref_bi", ctypes.c_void_p * 4)]


class LowresCostParams(ctypes.Structure):
    """x265hip_lowres_cost_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("stride", ctypes.c_ssize_t), ("width_in_cu", ctypes.c_int), ("height_in_cu", ctypes.c_int),
                ("cost_q", ctypes.c_void_p), ("qoff", ctypes.c_int), ("bframe_bias", ctypes.c_int),
                ("pairs", ctypes.POINTER(LowresCostPair)), ("npairs", ctypes.c_int), ("pairs_on_device", ctypes.c_int)]


class MESearchJob(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("px", "py", "w", "h", "qmvpx", "qmvpy", "out_qmvx", "out_qmvy", "out_cost")]


class MESearchParams(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_int), ("fenc", ctypes.c_void_p), ("fenc_stride", ctypes.c_ssize_t),
                ("fref", ctypes.c_void_p), ("fref_stride", ctypes.c_ssize_t),
                ("method", ctypes.c_int), ("subme", ctypes.c_int), ("merange", ctypes.c_int),
                ("cost_q", ctypes.c_void_p), ("qoff", ctypes.c_int),
                ("mvmin_x", ctypes.c_int), ("mvmin_y", ctypes.c_int), ("mvmax_x", ctypes.c_int), ("mvmax_y", ctypes.c_int),
                ("jobs", ctypes.c_void_p), ("njobs", ctypes.c_int), ("mvc", ctypes.c_void_p), ("num_mvc", ctypes.c_void_p),
                ("integral", ctypes.c_void_p * 12)]


class SeaIntegralParams(ctypes.Structure):
    """x265hip_sea_integral_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("ref", ctypes.c_void_p), ("stride", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int),
                ("planes", ctypes.c_void_p * 12)]


SEA_PLANE_DIMS = ((32, 32), (32, 24), (32, 8), (24, 32), (16, 16), (16, 12), (16, 4), (12, 16), (8, 32), (8, 8), (4, 16), (4, 4))   # framedata.h:171


ME_DIA, ME_HEX, ME_UMH, ME_STAR, ME_SEA, ME_FULL = range(6)


def me_search_job_dtype():
    import numpy as np
    return np.dtype([(n, "<i4") for n in ("px", "py", "w", "h", "qmvpx", "qmvpy", "out_qmvx", "out_qmvy", "out_cost")])


class DeblockBsParams(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("level", ctypes.c_int),
                ("mv", ctypes.c_void_p), ("num_sig", ctypes.c_void_p), ("bs_ver", ctypes.c_void_p), ("bs_hor", ctypes.c_void_p),
                ("intra", ctypes.c_void_p),
                ("slice_b", ctypes.c_int), ("mv1", ctypes.c_void_p), ("ref0", ctypes.c_void_p), ("ref1", ctypes.c_void_p)]


class DeblockChromaParams(ctypes.Structure):
    """x265hip_deblock_chroma_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p), ("stride", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("bs_ver", ctypes.c_void_p), ("bs_hor", ctypes.c_void_p),
                ("qp", ctypes.c_int), ("qp_map", ctypes.c_void_p),
                ("cb_qp_offset", ctypes.c_int), ("cr_qp_offset", ctypes.c_int), ("tc_offset_div2", ctypes.c_int)]


class DeblockParams(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_int), ("rec", ctypes.c_void_p), ("stride", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("bs_ver", ctypes.c_void_p), ("bs_hor", ctypes.c_void_p),
                ("qp", ctypes.c_int), ("qp_map", ctypes.c_void_p), ("beta_offset_div2", ctypes.c_int), ("tc_offset_div2", ctypes.c_int)]


def lib() -> ctypes.CDLL:
    """Load libx265hip.so (built in-tree by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise X265HipError(f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback for the product path")
        # One HIP runtime per process: torch bundles its own libamdhip64 / libhsa-runtime64 and whichever copy is loaded first serves
        # both.  If this library pulled in /opt/rocm's copies before torch was imported, torch's later initialisation found "No HIP GPUs".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.x265hip_version.restype = ctypes.c_char_p
        L.x265hip_last_error.restype = ctypes.c_char_p
        L.x265hip_table_calls.restype = ctypes.c_uint64 if hasattr(L, "x265hip_table_calls") else None
        L.x265hip_me_fullsearch.argtypes = [ctypes.POINTER(MEParams), ctypes.c_void_p]
        L.x265hip_me_best_reset.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.x265hip_pixelcmp_batch.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc < 0:
        raise X265HipError(f"{what} failed ({rc}): {lib().x265hip_last_error().decode()}")
    return rc


def exported_symbols():
    """Names include/x265hip.h declares; used by the CPU-only ABI test."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "x265hip.h")
    text = open(hdr).read()
    inline = set(re.findall(r"static inline [^\n(]*\b(x265hip_[a-z0-9_]+)\s*\(", text))      # header-only helpers are not exports
    return sorted(set(re.findall(r"\b(x265hip_[a-z0-9_]+)\s*\(", text)) - inline)


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def me_fullsearch(depth, width, height, rng, fenc, fenc_stride, fref, fref_stride,
                  surf=None, best=None, cost_x=None, cost_y=None,
                  fenc_off=0, fref_off=0, stream=None, surf_format=SURF_I32, centres=None):
    """fenc/fref: torch tensors holding the planes; *_off = element offset of pixel (0,0)."""
    es = 1 if depth == 8 else 2
    p = MEParams()
    p.surf_format = surf_format
    p.depth, p.width, p.height, p.range = depth, width, height, rng
    p.fenc, p.fenc_stride = fenc.data_ptr() + fenc_off * es, fenc_stride
    p.fref, p.fref_stride = fref.data_ptr() + fref_off * es, fref_stride
    p.surf, p.best = _p(surf), _p(best)
    p.cost_x, p.cost_y = _p(cost_x), _p(cost_y)
    p.centres = _p(centres)
    s = current_stream() if stream is None else stream
    check(lib().x265hip_me_fullsearch(ctypes.byref(p), s), "x265hip_me_fullsearch")


def subpel_refine(depth, width, height, rng, subme, fenc, fenc_stride, fref, fref_stride, best_in, cost_q, qoff, out,
                  fenc_off=0, fref_off=0, stream=None, phase_planes=None):
    """phase_planes: byte tensor with the 15 luma phase planes of the fref buffer (x265hip_phase_planes, same geometry as fref's tensor)."""
    es = 1 if depth == 8 else 2
    p = SubpelParams()
    p.depth, p.width, p.height, p.range, p.subme = depth, width, height, rng, subme
    p.fenc, p.fenc_stride = fenc.data_ptr() + fenc_off * es, fenc_stride
    p.fref, p.fref_stride = fref.data_ptr() + fref_off * es, fref_stride
    p.best_in, p.cost_q, p.qoff, p.out = best_in.data_ptr(), cost_q.data_ptr(), qoff, out.data_ptr()
    if phase_planes is not None:
        p.phase_planes, p.phase_plane_samples = phase_planes.data_ptr() + fref_off * es, fref.numel() * fref.element_size() // es
    s = current_stream() if stream is None else stream
    f = lib().x265hip_subpel_refine
    f.argtypes = [ctypes.POINTER(SubpelParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_subpel_refine")


def lowres_init(depth, src, src_stride, src_off, planes, stride, org, width, lines, margin_x, margin_y, stream=None):
    """src: padded full-resolution plane tensor (src_off = element offset of pixel (0,0)); planes: 4 tensors of the lowres
    geometry (org = element offset of their pixel (0,0))."""
    es = 1 if depth == 8 else 2
    p = LowresInitParams()
    p.depth, p.src, p.src_stride = depth, src.data_ptr() + src_off * es, src_stride
    for i in range(4):
        p.plane[i] = planes[i].data_ptr() + org * es
    p.stride, p.width, p.lines, p.margin_x, p.margin_y = stride, width, lines, margin_x, margin_y
    s = current_stream() if stream is None else stream
    f = lib().x265hip_lowres_init
    f.argtypes = [ctypes.POINTER(LowresInitParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_lowres_init")


def lowres_intra(depth, plane, stride, org, width_in_cu, height_in_cu, intra_penalty, intra_cost, intra_mode, lowres_costs, stream=None):
    es = 1 if depth == 8 else 2
    p = LowresIntraParams()
    p.depth, p.plane, p.stride = depth, plane.data_ptr() + org * es, stride
    p.width_in_cu, p.height_in_cu, p.intra_penalty = width_in_cu, height_in_cu, intra_penalty
    p.intra_cost, p.intra_mode, p.lowres_costs = intra_cost.data_ptr(), intra_mode.data_ptr(), lowres_costs.data_ptr()
    s = current_stream() if stream is None else stream
    f = lib().x265hip_lowres_intra
    f.argtypes = [ctypes.POINTER(LowresIntraParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_lowres_intra")


def lowres_cost_pair(depth, org, cur, ref_planes, intra_cost, mvs, mv_costs, lowres_costs, row_satds, frame, inv_qscale=None,
                     ref1_planes=None, mvs1=None, mv_costs1=None, do_search=(1, 1), ref_bi_planes=None):
    """One picture for lowres_cost (ref_bi_planes: --weightp on a B picture - ref_planes weighted, ref_bi_planes the unweighted list 0): cur = the current picture's plane 0, ref_planes = the list-0 reference's four phase planes
    (pixel (0,0) at element `org` of each tensor); ref1_planes / mvs1 / mv_costs1 = list 1 of a B picture."""
    es = 1 if depth == 8 else 2
    q = LowresCostPair()
    q.cur = cur.data_ptr() + org * es
    for i in range(4):
        q.ref[i] = ref_planes[i].data_ptr() + org * es
        q.ref1[i] = None if ref1_planes is None else ref1_planes[i].data_ptr() + org * es
        q.ref_bi[i] = None if ref_bi_planes is None else ref_bi_planes[i].data_ptr() + org * es
    q.intra_cost, q.inv_qscale = intra_cost.data_ptr(), _p(inv_qscale)
    q.mvs, q.mv_costs, q.mvs1, q.mv_costs1 = mvs.data_ptr(), mv_costs.data_ptr(), _p(mvs1), _p(mv_costs1)
    q.do_search[0], q.do_search[1] = do_search
    q.lowres_costs, q.row_satds, q.frame = lowres_costs.data_ptr(), row_satds.data_ptr(), frame.data_ptr()
    return q


def lowres_cost_table(pairs, device, stream=None):
    """Upload a list of LowresCostPair once (pinned staging, non-blocking copy on `stream`): returns (device tensor, staging tensor,
    kind) for lowres_cost(..., device_table=...).  Keep both tensors alive until the launch has run."""
    import numpy as np
    import torch
    arr = (LowresCostPair * len(pairs))(*pairs)
    host = torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy()).pin_memory()
    if stream is not None:
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            dev = host.to(device, non_blocking=True)
    else:
        dev = host.to(device, non_blocking=True)
    return dev, host, (2 if pairs[0].ref1[0] else 1)


def lowres_cost(depth, stride, width_in_cu, height_in_cu, cost_q, qoff, pairs, bframe_bias=0, stream=None, device_table=None):
    """Lookahead frame cost estimate (estimateFrameCost, slicetype.cpp:3115-3388) of a batch of independent pictures of one
    geometry and one kind (all P or all B); pairs: list of LowresCostPair (lowres_cost_pair), or None with
    device_table = lowres_cost_table(...) for a table that already sits on the device (no host-side blocking)."""
    p = LowresCostParams()
    p.depth, p.stride, p.width_in_cu, p.height_in_cu = depth, stride, width_in_cu, height_in_cu
    p.cost_q, p.qoff, p.bframe_bias = cost_q.data_ptr(), qoff, bframe_bias
    if device_table is not None:
        dev, _, kind = device_table
        p.pairs = ctypes.cast(ctypes.c_void_p(dev.data_ptr()), ctypes.POINTER(LowresCostPair))
        p.npairs, p.pairs_on_device = dev.numel() // ctypes.sizeof(LowresCostPair), kind
    else:
        arr = (LowresCostPair * len(pairs))(*pairs)
        p.pairs, p.npairs, p.pairs_on_device = arr, len(pairs), 0
    s = current_stream() if stream is None else stream
    f = lib().x265hip_lowres_cost
    f.argtypes = [ctypes.POINTER(LowresCostParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_lowres_cost")


def me_search(depth, fenc, fenc_stride, fenc_off, fref, fref_stride, fref_off, method, subme, merange, cost_q, qoff,
              mvmin, mvmax, jobs, njobs, stream=None, mvc=None, num_mvc=None, integral=None):
    """jobs: device uint8 tensor holding njobs x265hip_me_search_job records (me_search_job_dtype), updated in place.
    mvc / num_mvc: optional device int32 tensors [njobs][12][2] / [njobs] (motionEstimate's extra candidates).
    integral: X265_SEA only - the reference's block-sum planes, a (planes tensor, org) pair from sea_integral()."""
    es = 1 if depth == 8 else 2
    p = MESearchParams()
    p.depth = depth
    p.fenc, p.fenc_stride = fenc.data_ptr() + fenc_off * es, fenc_stride
    p.fref, p.fref_stride = fref.data_ptr() + fref_off * es, fref_stride
    p.method, p.subme, p.merange = method, subme, merange
    p.cost_q, p.qoff = cost_q.data_ptr(), qoff
    p.mvmin_x, p.mvmin_y, p.mvmax_x, p.mvmax_y = mvmin[0], mvmin[1], mvmax[0], mvmax[1]
    p.jobs, p.njobs = jobs.data_ptr(), njobs
    p.mvc, p.num_mvc = _p(mvc), _p(num_mvc)
    if integral is not None:
        planes, org = integral
        for k in range(12):
            p.integral[k] = planes[k].data_ptr() + org * 4
    s = current_stream() if stream is None else stream
    f = lib().x265hip_me_search
    f.argtypes = [ctypes.POINTER(MESearchParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_me_search")


class AqEnergyParams(ctypes.Structure):
    """x265hip_aq_energy_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("y", ctypes.c_void_p), ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p),
                ("stride", ctypes.c_ssize_t), ("stride_c", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("qg_size", ctypes.c_int),
                ("energy", ctypes.c_void_p), ("wp", ctypes.c_void_p)]


def aq_energy(depth, y, stride, org, width, height, qg_size, energy, wp, cb=None, cr=None, stride_c=0, org_c=0, stream=None):
    """acEnergyCu for every qg_size x qg_size block + the wp_sum / wp_ssd raw totals (x265hip_aq_energy).  energy: device int32 tensor
    [blocks] (uint32 bits); wp: device int64 tensor [6]."""
    es = 1 if depth == 8 else 2
    p = AqEnergyParams()
    p.depth, p.stride, p.stride_c, p.width, p.height, p.qg_size = depth, stride, stride_c, width, height, qg_size
    p.y = y.data_ptr() + org * es
    p.cb = None if cb is None else cb.data_ptr() + org_c * es
    p.cr = None if cr is None else cr.data_ptr() + org_c * es
    p.energy, p.wp = energy.data_ptr(), wp.data_ptr()
    s = current_stream() if stream is None else stream
    f = lib().x265hip_aq_energy
    f.argtypes = [ctypes.POINTER(AqEnergyParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_aq_energy")


class AqOffsetsParams(ctypes.Structure):
    """x265hip_aq_offsets_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("qg_size", ctypes.c_int), ("aq_mode", ctypes.c_int), ("aq_strength", ctypes.c_double),
                ("nblocks", ctypes.c_int), ("energy", ctypes.c_void_p), ("qp_aq_offset", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p)]


def aq_offsets(depth, qg_size, aq_mode, aq_strength, energy):
    """Host side of the AQ pass (x265hip_aq_offsets): energy = numpy uint32 [blocks]; returns (qp_aq_offset float64, inv_qscale int32)."""
    import numpy as np
    e = np.ascontiguousarray(energy, dtype=np.uint32)
    qp, inv = np.zeros(len(e), np.float64), np.zeros(len(e), np.int32)
    p = AqOffsetsParams()
    p.depth, p.qg_size, p.aq_mode, p.aq_strength, p.nblocks = depth, qg_size, aq_mode, float(aq_strength), len(e)
    p.energy, p.qp_aq_offset, p.inv_qscale = e.ctypes.data, qp.ctypes.data, inv.ctypes.data
    f = lib().x265hip_aq_offsets
    f.argtypes = [ctypes.POINTER(AqOffsetsParams)]
    check(f(ctypes.byref(p)), "x265hip_aq_offsets")
    return qp, inv


class AqFrameHostParams(ctypes.Structure):
    """x265hip_aq_frame_host_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("y", ctypes.c_void_p), ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p),
                ("stride", ctypes.c_ssize_t), ("stride_c", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("qg_size", ctypes.c_int), ("aq_mode", ctypes.c_int), ("aq_strength", ctypes.c_double),
                ("width_in_cu", ctypes.c_int), ("height_in_cu", ctypes.c_int), ("normalise_wp", ctypes.c_int),
                ("qp_aq_offset", ctypes.c_void_p), ("qp_cutree_offset", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p), ("inv_qscale_8x8", ctypes.c_void_p),
                ("energy", ctypes.c_void_p), ("wp_sum", ctypes.c_void_p), ("wp_ssd", ctypes.c_void_p)]


def aq_frame_host(depth, y, stride, org, width, height, qg_size, aq_mode, aq_strength, cb=None, cr=None, stride_c=0, org_c=0, normalise_wp=True, lowres_grid=None):
    """LookaheadTLD::calcAdaptiveQuantFrame behind host pointers (x265hip_aq_frame_host): y / cb / cr = numpy planes in HOST memory, sample
    (0,0) at org / org_c.  Returns dict(energy, qp_aq_offset, qp_cutree_offset, inv_qscale, inv_qscale_8x8 or None, wp_sum, wp_ssd)."""
    import numpy as np
    n = ((width + qg_size - 1) // qg_size) * ((height + qg_size - 1) // qg_size)
    out = {"energy": np.zeros(n, np.uint32), "qp_aq_offset": np.zeros(n, np.float64), "qp_cutree_offset": np.zeros(n, np.float64), "inv_qscale": np.zeros(n, np.int32),
           "inv_qscale_8x8": None, "wp_sum": np.zeros(3, np.uint64), "wp_ssd": np.zeros(3, np.uint64)}
    es = y.itemsize
    p = AqFrameHostParams()
    p.depth, p.stride, p.stride_c, p.width, p.height, p.qg_size, p.aq_mode, p.aq_strength = depth, stride, stride_c, width, height, qg_size, aq_mode, float(aq_strength)
    p.y = y.ctypes.data + org * es
    p.cb = None if cb is None else cb.ctypes.data + org_c * es
    p.cr = None if cr is None else cr.ctypes.data + org_c * es
    p.normalise_wp = int(bool(normalise_wp))
    if lowres_grid:
        p.width_in_cu, p.height_in_cu = lowres_grid
        out["inv_qscale_8x8"] = np.zeros(lowres_grid[0] * lowres_grid[1], np.int32)
        p.inv_qscale_8x8 = out["inv_qscale_8x8"].ctypes.data
    for k in ("energy", "qp_aq_offset", "qp_cutree_offset", "inv_qscale", "wp_sum", "wp_ssd"):
        setattr(p, k, out[k].ctypes.data)
    f = lib().x265hip_aq_frame_host
    f.argtypes = [ctypes.POINTER(AqFrameHostParams)]
    check(f(ctypes.byref(p)), "x265hip_aq_frame_host")
    return out


class WeightAnalyseRef(ctypes.Structure):
    """x265hip_weight_analyse_ref (include/x265hip.h)."""
    _fields_ = [("lowres", ctypes.c_void_p * 4), ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p), ("mvs", ctypes.c_void_p),
                ("wp_ssd", ctypes.c_uint64 * 3), ("wp_sum", ctypes.c_uint64 * 3), ("plane_key", ctypes.c_uint64)]


class WeightAnalyseHostParams(ctypes.Structure):
    """x265hip_weight_analyse_host_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("lowres", ctypes.c_void_p), ("lowres_stride", ctypes.c_ssize_t),
                ("lowres_width", ctypes.c_int), ("lowres_lines", ctypes.c_int), ("lowres_margin_x", ctypes.c_int), ("lowres_margin_y", ctypes.c_int),
                ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p), ("stride_c", ctypes.c_ssize_t), ("margin_xc", ctypes.c_int), ("margin_yc", ctypes.c_int),
                ("pic_width", ctypes.c_int), ("pic_height", ctypes.c_int), ("intra_cost", ctypes.c_void_p),
                ("wp_ssd", ctypes.c_uint64 * 3), ("wp_sum", ctypes.c_uint64 * 3), ("plane_key", ctypes.c_uint64), ("nlists", ctypes.c_int),
                ("ref", WeightAnalyseRef * 2), ("weights", ctypes.c_void_p), ("denoms", ctypes.c_void_p)]


def weight_analyse_host(depth, cur, refs, pic_width, pic_height, intra_cost, lowres_margin, chroma_margin, plane_keys=None):
    """weightAnalyse behind host pointers (x265hip_weight_analyse_host); cur = dict(lowres, lowres_stride, lowres_width, lowres_lines, cb, cr, stride_c, wp_ssd, wp_sum), refs = dicts(lowres[4], cb, cr, mvs, wp_ssd, wp_sum) (numpy planes in HOST
    memory, (array, org) pairs).  Returns (weights int32 [2, 3, 4], denoms int32 [2, 2])."""
    import numpy as np
    es = cur["lowres"][0].itemsize
    at = lambda pair: pair[0].ctypes.data + pair[1] * es
    p = WeightAnalyseHostParams()
    p.depth, p.lowres, p.lowres_stride = depth, at(cur["lowres"]), cur["lowres_stride"]
    p.lowres_width, p.lowres_lines, p.lowres_margin_x, p.lowres_margin_y = cur["lowres_width"], cur["lowres_lines"], lowres_margin[0], lowres_margin[1]
    p.cb, p.cr, p.stride_c, p.margin_xc, p.margin_yc = at(cur["cb"]), at(cur["cr"]), cur["stride_c"], chroma_margin[0], chroma_margin[1]
    p.pic_width, p.pic_height = pic_width, pic_height
    ic = np.ascontiguousarray(intra_cost, np.int32)
    p.intra_cost = ic.ctypes.data
    keep = []
    for k in range(3):
        p.wp_ssd[k], p.wp_sum[k] = int(cur["wp_ssd"][k]), int(cur["wp_sum"][k])
    p.plane_key = plane_keys[0] if plane_keys else 0
    p.nlists = len(refs)
    for i, r in enumerate(refs):
        for k in range(4):
            p.ref[i].lowres[k] = at(r["lowres"][k])
        p.ref[i].cb, p.ref[i].cr = at(r["cb"]), at(r["cr"])
        if r.get("mvs") is not None:
            m = np.ascontiguousarray(r["mvs"], np.int32)
            keep.append(m)
            p.ref[i].mvs = m.ctypes.data
        for k in range(3):
            p.ref[i].wp_ssd[k], p.ref[i].wp_sum[k] = int(r["wp_ssd"][k]), int(r["wp_sum"][k])
        p.ref[i].plane_key = plane_keys[1 + i] if plane_keys else 0
    out, den = np.zeros((2, 3, 4), np.int32), np.zeros((2, 2), np.int32)
    p.weights, p.denoms = out.ctypes.data, den.ctypes.data
    f = lib().x265hip_weight_analyse_host
    f.argtypes = [ctypes.POINTER(WeightAnalyseHostParams)]
    check(f(ctypes.byref(p)), "x265hip_weight_analyse_host")
    return out, den


class AqHevcParams(ctypes.Structure):
    """x265hip_aq_hevc_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("y", ctypes.c_void_p), ("stride", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("part", ctypes.c_int), ("sums", ctypes.c_void_p)]


class AqHevcOffsetsParams(ctypes.Structure):
    """x265hip_aq_hevc_offsets_params (include/x265hip.h)."""
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("part", ctypes.c_int), ("qp_adaptation_range", ctypes.c_double),
                ("sums", ctypes.c_void_p), ("activity", ctypes.c_void_p), ("qp_offset", ctypes.c_void_p), ("avg_activity", ctypes.c_void_p),
                ("inv_qscale", ctypes.c_void_p)]


def aq_hevc_quadrants(depth, y, stride, org, width, height, part, sums, stream=None):
    """x265hip_aq_hevc_quadrants: sums = device int64 tensor [partitions * 8] (uint64 bits)."""
    es = 1 if depth == 8 else 2
    p = AqHevcParams()
    p.depth, p.y, p.stride, p.width, p.height, p.part, p.sums = depth, y.data_ptr() + org * es, stride, width, height, part, sums.data_ptr()
    s = current_stream() if stream is None else stream
    f = lib().x265hip_aq_hevc_quadrants
    f.argtypes = [ctypes.POINTER(AqHevcParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_aq_hevc_quadrants")


def aq_hevc_offsets(width, height, part, qp_adaptation_range, sums):
    """Host side of --hevc-aq for one layer (x265hip_aq_hevc_offsets): sums = numpy uint64 [partitions, 4, 2]; returns (activity, qp_offset
    float64 [partitions], avg_activity float, inv_qscale int32 [partitions])."""
    import numpy as np
    sm = np.ascontiguousarray(sums, dtype=np.uint64)
    n = ((width + part - 1) // part) * ((height + part - 1) // part)
    act, qp, avg, inv = np.zeros(n, np.float64), np.zeros(n, np.float64), np.zeros(1, np.float64), np.zeros(n, np.int32)
    p = AqHevcOffsetsParams()
    p.width, p.height, p.part, p.qp_adaptation_range = width, height, part, float(qp_adaptation_range)
    p.sums, p.activity, p.qp_offset, p.avg_activity, p.inv_qscale = sm.ctypes.data, act.ctypes.data, qp.ctypes.data, avg.ctypes.data, inv.ctypes.data
    f = lib().x265hip_aq_hevc_offsets
    f.argtypes = [ctypes.POINTER(AqHevcOffsetsParams)]
    check(f(ctypes.byref(p)), "x265hip_aq_hevc_offsets")
    return act, qp, float(avg[0]), inv


class CuTreePropagateParams(ctypes.Structure):
    """x265hip_cutree_propagate_params (include/x265hip.h)."""
    _fields_ = [("width_in_cu", ctypes.c_int), ("height_in_cu", ctypes.c_int),
                ("propagate_in", ctypes.c_void_p), ("intra_cost", ctypes.c_void_p), ("lowres_costs", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p),
                ("mvs0", ctypes.c_void_p), ("mvs1", ctypes.c_void_p), ("fps_factor", ctypes.c_double), ("bipred_weight", ctypes.c_int),
                ("ref_cost0", ctypes.c_void_p), ("ref_cost1", ctypes.c_void_p)]


def cutree_propagate(width_in_cu, height_in_cu, propagate_in, intra_cost, lowres_costs, inv_qscale, mvs0, mvs1, fps_factor, bipred_weight,
                     ref_cost0, ref_cost1=None, stream=None):
    """One cuTree propagation step (x265hip_cutree_propagate); device tensors, ref_cost0 / ref_cost1 (int16 tensors holding uint16 bits)
    updated in place."""
    p = CuTreePropagateParams()
    p.width_in_cu, p.height_in_cu = width_in_cu, height_in_cu
    p.propagate_in, p.intra_cost, p.lowres_costs, p.inv_qscale = _p(propagate_in), intra_cost.data_ptr(), lowres_costs.data_ptr(), inv_qscale.data_ptr()
    p.mvs0, p.mvs1, p.fps_factor, p.bipred_weight = mvs0.data_ptr(), _p(mvs1), float(fps_factor), int(bipred_weight)
    p.ref_cost0, p.ref_cost1 = ref_cost0.data_ptr(), _p(ref_cost1)
    s = current_stream() if stream is None else stream
    f = lib().x265hip_cutree_propagate
    f.argtypes = [ctypes.POINTER(CuTreePropagateParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_cutree_propagate")


class CuTreeFinishParams(ctypes.Structure):
    """x265hip_cutree_finish_params (include/x265hip.h)."""
    _fields_ = [("nblocks", ctypes.c_int), ("intra_cost", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p), ("propagate_cost", ctypes.c_void_p),
                ("qp_aq_offset", ctypes.c_void_p), ("fps_factor_q8", ctypes.c_int), ("weight_delta", ctypes.c_double), ("strength", ctypes.c_double),
                ("qp_cutree_offset", ctypes.c_void_p)]


def cutree_finish(intra_cost, inv_qscale, propagate_cost, qp_aq_offset, fps_factor_q8, weight_delta, strength, qp_cutree_offset):
    """Host side of the cuTree step (x265hip_cutree_finish): numpy arrays; returns the updated copy of qp_cutree_offset."""
    import numpy as np
    ic, iq = np.ascontiguousarray(intra_cost, np.int32), np.ascontiguousarray(inv_qscale, np.int32)
    pc, qa = np.ascontiguousarray(propagate_cost, np.uint16), np.ascontiguousarray(qp_aq_offset, np.float64)
    out = np.ascontiguousarray(qp_cutree_offset, np.float64).copy()
    p = CuTreeFinishParams()
    p.nblocks, p.fps_factor_q8, p.weight_delta, p.strength = len(ic), int(fps_factor_q8), float(weight_delta), float(strength)
    p.intra_cost, p.inv_qscale, p.propagate_cost, p.qp_aq_offset = ic.ctypes.data, iq.ctypes.data, pc.ctypes.data, qa.ctypes.data
    p.qp_cutree_offset = out.ctypes.data
    f = lib().x265hip_cutree_finish
    f.argtypes = [ctypes.POINTER(CuTreeFinishParams)]
    check(f(ctypes.byref(p)), "x265hip_cutree_finish")
    return out


class CuTreeFinishHevcParams(ctypes.Structure):
    """x265hip_cutree_finish_hevc_params (include/x265hip.h)."""
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("part", ctypes.c_int), ("blocks_in_row", ctypes.c_int),
                ("intra_cost", ctypes.c_void_p), ("inv_qscale", ctypes.c_void_p), ("propagate_cost", ctypes.c_void_p),
                ("fps_factor_q8", ctypes.c_int), ("weight_delta", ctypes.c_double), ("strength", ctypes.c_double),
                ("qp_offset", ctypes.c_void_p), ("cutree_offset", ctypes.c_void_p)]


def cutree_finish_hevc_aq(width, height, part, blocks_in_row, intra_cost, inv_qscale, propagate_cost, fps_factor_q8, weight_delta, strength, qp_offset):
    """x265hip_cutree_finish_hevc_aq (host-side): one layer's dCuTreeOffset from its dQpOffset; numpy arrays in, float64 [partitions] out."""
    import numpy as np
    ic, iq, pc = np.ascontiguousarray(intra_cost, np.int32), np.ascontiguousarray(inv_qscale, np.int32), np.ascontiguousarray(propagate_cost, np.uint16)
    qo = np.ascontiguousarray(qp_offset, np.float64)
    out = np.zeros_like(qo)
    p = CuTreeFinishHevcParams()
    p.width, p.height, p.part, p.blocks_in_row = width, height, part, blocks_in_row
    p.intra_cost, p.inv_qscale, p.propagate_cost = ic.ctypes.data, iq.ctypes.data, pc.ctypes.data
    p.fps_factor_q8, p.weight_delta, p.strength = int(fps_factor_q8), float(weight_delta), float(strength)
    p.qp_offset, p.cutree_offset = qo.ctypes.data, out.ctypes.data
    f = lib().x265hip_cutree_finish_hevc_aq
    f.argtypes = [ctypes.POINTER(CuTreeFinishHevcParams)]
    check(f(ctypes.byref(p)), "x265hip_cutree_finish_hevc_aq")
    return out


def cutree_finish_qg8(width_in_cu, height_in_cu, intra_cost, inv_qscale8x8, propagate_cost, qp_aq_offset, fps_factor_q8, weight_delta, strength, qp_cutree_offset):
    """x265hip_cutree_finish_qg8: the --qg-size 8 branch (offsets on the full-resolution grid); returns the updated copy."""
    import numpy as np
    ic, iq = np.ascontiguousarray(intra_cost, np.int32), np.ascontiguousarray(inv_qscale8x8, np.int32)
    pc, qa = np.ascontiguousarray(propagate_cost, np.uint16), np.ascontiguousarray(qp_aq_offset, np.float64)
    out = np.ascontiguousarray(qp_cutree_offset, np.float64).copy()
    p = CuTreeFinishParams()
    p.nblocks, p.fps_factor_q8, p.weight_delta, p.strength = len(ic), int(fps_factor_q8), float(weight_delta), float(strength)
    p.intra_cost, p.inv_qscale, p.propagate_cost, p.qp_aq_offset = ic.ctypes.data, iq.ctypes.data, pc.ctypes.data, qa.ctypes.data
    p.qp_cutree_offset = out.ctypes.data
    f = lib().x265hip_cutree_finish_qg8
    f.argtypes = [ctypes.POINTER(CuTreeFinishParams), ctypes.c_int, ctypes.c_int]
    check(f(ctypes.byref(p), width_in_cu, height_in_cu), "x265hip_cutree_finish_qg8")
    return out


class FrameCostRecalculateParams(ctypes.Structure):
    """x265hip_frame_cost_recalculate_params (include/x265hip.h)."""
    _fields_ = [("width_in_cu", ctypes.c_int), ("height_in_cu", ctypes.c_int), ("lowres_costs", ctypes.c_void_p), ("qp_cutree_offset", ctypes.c_void_p),
                ("row_satds", ctypes.c_void_p), ("score", ctypes.c_void_p)]


def frame_cost_recalculate(width_in_cu, height_in_cu, lowres_costs, qp_cutree_offset):
    """Host side (x265hip_frame_cost_recalculate): numpy arrays in; returns (score, row_satds int32 [height_in_cu])."""
    import numpy as np
    lc, qp = np.ascontiguousarray(lowres_costs, np.uint16), np.ascontiguousarray(qp_cutree_offset, np.float64)
    rows, score = np.zeros(height_in_cu, np.int32), np.zeros(1, np.int64)
    p = FrameCostRecalculateParams()
    p.width_in_cu, p.height_in_cu = width_in_cu, height_in_cu
    p.lowres_costs, p.qp_cutree_offset, p.row_satds, p.score = lc.ctypes.data, qp.ctypes.data, rows.ctypes.data, score.ctypes.data
    f = lib().x265hip_frame_cost_recalculate
    f.argtypes = [ctypes.POINTER(FrameCostRecalculateParams)]
    check(f(ctypes.byref(p)), "x265hip_frame_cost_recalculate")
    return int(score[0]), rows


def frame_cost_recalculate_qg8(width_in_cu, height_in_cu, lowres_costs, qp_cutree_offset):
    """x265hip_frame_cost_recalculate_qg8: qp_cutree_offset on the full-resolution grid; returns (score, row_satds)."""
    import numpy as np
    lc, qp = np.ascontiguousarray(lowres_costs, np.uint16), np.ascontiguousarray(qp_cutree_offset, np.float64)
    rows, score = np.zeros(height_in_cu, np.int32), np.zeros(1, np.int64)
    p = FrameCostRecalculateParams()
    p.width_in_cu, p.height_in_cu = width_in_cu, height_in_cu
    p.lowres_costs, p.qp_cutree_offset, p.row_satds, p.score = lc.ctypes.data, qp.ctypes.data, rows.ctypes.data, score.ctypes.data
    f = lib().x265hip_frame_cost_recalculate_qg8
    f.argtypes = [ctypes.POINTER(FrameCostRecalculateParams)]
    check(f(ctypes.byref(p)), "x265hip_frame_cost_recalculate_qg8")
    return int(score[0]), rows


class LowresWeightCostParams(ctypes.Structure):
    """x265hip_lowres_weight_cost_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("fenc", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("stride", ctypes.c_ssize_t),
                ("width", ctypes.c_int), ("lines", ctypes.c_int), ("intra_cost", ctypes.c_void_p),
                ("ncand", ctypes.c_int), ("cand", (ctypes.c_int * 4) * 4), ("cost", ctypes.c_void_p)]


class LowresWeightApplyParams(ctypes.Structure):
    """x265hip_lowres_weight_apply_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("src", ctypes.c_void_p * 4), ("dst", ctypes.c_void_p * 4),
                ("stride", ctypes.c_ssize_t), ("rows", ctypes.c_int),
                ("scale", ctypes.c_int), ("denom", ctypes.c_int), ("offset", ctypes.c_int)]


def lowres_weight_cost(depth, fenc, ref, stride, org, width, lines, intra_cost, cands, cost, stream=None):
    """LookaheadTLD::weightCostLuma for up to four candidate weights: cands = [None | (scale, denom, offset)], cost = device int32
    tensor [len(cands)] (uint32 bits), overwritten."""
    es = 1 if depth == 8 else 2
    p = LowresWeightCostParams()
    p.depth, p.stride, p.width, p.lines = depth, stride, width, lines
    p.fenc, p.ref = fenc.data_ptr() + org * es, ref.data_ptr() + org * es
    p.intra_cost, p.cost, p.ncand = intra_cost.data_ptr(), cost.data_ptr(), len(cands)
    for i, c in enumerate(cands):
        vals = (0, 0, 0, 0) if c is None else (1, int(c[0]), int(c[1]), int(c[2]))
        for k in range(4):
            p.cand[i][k] = vals[k]
    s = current_stream() if stream is None else stream
    f = lib().x265hip_lowres_weight_cost
    f.argtypes = [ctypes.POINTER(LowresWeightCostParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_lowres_weight_cost")


def lowres_weight_apply(depth, src_planes, dst_planes, stride, rows, weight, stream=None):
    """weight_pp over the four whole lowres buffers (allocation start, rows x stride samples) with weight = (scale, denom, offset)."""
    p = LowresWeightApplyParams()
    p.depth, p.stride, p.rows = depth, stride, rows
    p.scale, p.denom, p.offset = int(weight[0]), int(weight[1]), int(weight[2])
    for i in range(4):
        p.src[i], p.dst[i] = src_planes[i].data_ptr(), dst_planes[i].data_ptr()
    s = current_stream() if stream is None else stream
    f = lib().x265hip_lowres_weight_apply
    f.argtypes = [ctypes.POINTER(LowresWeightApplyParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_lowres_weight_apply")


def sea_integral(depth, ref, stride, org, width, height, margin_x, margin_y, planes=None, stream=None):
    """The twelve block-sum planes of a padded reference picture (x265hip_sea_integral): returns (planes, org) with planes a
    uint32 device tensor [12][rows * stride] laid out like the picture (entry of sample (0,0) at element org)."""
    import torch
    es = 1 if depth == 8 else 2
    if planes is None:
        planes = torch.zeros((12, ref.numel() * ref.element_size() // es), dtype=torch.int32, device=ref.device)
    p = SeaIntegralParams()
    p.depth, p.ref, p.stride = depth, ref.data_ptr() + org * es, stride
    p.width, p.height, p.margin_x, p.margin_y = width, height, margin_x, margin_y
    for k in range(12):
        p.planes[k] = planes[k].data_ptr() + org * 4
    s = current_stream() if stream is None else stream
    f = lib().x265hip_sea_integral
    f.argtypes = [ctypes.POINTER(SeaIntegralParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_sea_integral")
    return planes, org


def deblock_bs_inter(width, height, level, mv, num_sig, bs_ver, bs_hor, stream=None, intra=None, slice_b=False, mv1=None, ref0=None,
                     ref1=None):
    """Boundary strengths of a picture of square blocks.  ref0 / ref1 (int8 picture ids per block), mv1 and slice_b describe pictures
    with several references / B pictures (deblock.cpp:217-247); left out = one list-0 reference."""
    p = DeblockBsParams()
    p.width, p.height, p.level = width, height, level
    p.mv, p.num_sig, p.bs_ver, p.bs_hor = mv.data_ptr(), num_sig.data_ptr(), bs_ver.data_ptr(), bs_hor.data_ptr()
    p.intra = _p(intra)
    p.slice_b, p.mv1, p.ref0, p.ref1 = int(bool(slice_b)), _p(mv1), _p(ref0), _p(ref1)
    s = current_stream() if stream is None else stream
    f = lib().x265hip_deblock_bs_inter
    f.argtypes = [ctypes.POINTER(DeblockBsParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_deblock_bs_inter")


def deblock_chroma(depth, cb, cr, stride, org, width, height, bs_ver, bs_hor, qp, qp_map=None, cb_qp_offset=0, cr_qp_offset=0,
                   tc_offset_div2=0, stream=None):
    """edgeFilterChroma over the Cb / Cr planes of a 4:2:0 picture in place (width / height = luma size; org = element offset of
    sample (0,0) in both plane tensors)."""
    es = 1 if depth == 8 else 2
    p = DeblockChromaParams()
    p.depth, p.cb, p.cr, p.stride = depth, cb.data_ptr() + org * es, cr.data_ptr() + org * es, stride
    p.width, p.height, p.bs_ver, p.bs_hor = width, height, bs_ver.data_ptr(), bs_hor.data_ptr()
    p.qp, p.qp_map = qp, _p(qp_map)
    p.cb_qp_offset, p.cr_qp_offset, p.tc_offset_div2 = cb_qp_offset, cr_qp_offset, tc_offset_div2
    s = current_stream() if stream is None else stream
    f = lib().x265hip_deblock_chroma
    f.argtypes = [ctypes.POINTER(DeblockChromaParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_deblock_chroma")


def deblock_luma(depth, rec, stride, org, width, height, bs_ver, bs_hor, qp, qp_map=None, beta_offset_div2=0, tc_offset_div2=0, stream=None):
    es = 1 if depth == 8 else 2
    p = DeblockParams()
    p.depth, p.rec, p.stride, p.width, p.height = depth, rec.data_ptr() + org * es, stride, width, height
    p.bs_ver, p.bs_hor, p.qp, p.qp_map = bs_ver.data_ptr(), bs_hor.data_ptr(), qp, _p(qp_map)
    p.beta_offset_div2, p.tc_offset_div2 = beta_offset_div2, tc_offset_div2
    s = current_stream() if stream is None else stream
    f = lib().x265hip_deblock_luma
    f.argtypes = [ctypes.POINTER(DeblockParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_deblock_luma")


class SaoStatsParams(ctypes.Structure):
    """x265hip_sao_stats_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("fenc", ctypes.c_void_p), ("fenc_stride", ctypes.c_ssize_t),
                ("rec", ctypes.c_void_p), ("rec_stride", ctypes.c_ssize_t), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("count", ctypes.c_void_p), ("offset_org", ctypes.c_void_p),
                ("ctu_width", ctypes.c_int), ("ctu_height", ctypes.c_int), ("plane_offset", ctypes.c_int)]


def sao_decide(depth, count, offset_org, nctu, params, init_offset=None, stream=None):
    """x265hip_sao_decide: saoStatsInitialOffset + distortion-only type choice per CTU; params int32 [nctu * 7] (device)."""
    s = current_stream() if stream is None else stream
    f = lib().x265hip_sao_decide
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    check(f(depth, count.data_ptr(), offset_org.data_ptr(), nctu, _p(init_offset), params.data_ptr(), s), "x265hip_sao_decide")


class SaoApplyParams(ctypes.Structure):
    """x265hip_sao_apply_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("src", ctypes.c_void_p), ("src_stride", ctypes.c_ssize_t),
                ("dst", ctypes.c_void_p), ("dst_stride", ctypes.c_ssize_t), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("ctu_params", ctypes.c_void_p), ("ctu_width", ctypes.c_int), ("ctu_height", ctypes.c_int)]


def sao_stats(depth, fenc, fenc_stride, fenc_org, rec, rec_stride, rec_org, width, height, count, offset_org, stream=None, ctu=(0, 0),
              plane_offset=0):
    """SAO::calcSaoStatsCTU for every CTU: count / offset_org int32 [numCtu * 5 * 32] device tensors."""
    es = 1 if depth == 8 else 2
    p = SaoStatsParams()
    p.depth, p.fenc, p.fenc_stride = depth, fenc.data_ptr() + fenc_org * es, fenc_stride
    p.rec, p.rec_stride, p.width, p.height = rec.data_ptr() + rec_org * es, rec_stride, width, height
    p.count, p.offset_org = count.data_ptr(), offset_org.data_ptr()
    p.ctu_width, p.ctu_height, p.plane_offset = ctu[0], ctu[1], plane_offset
    s = current_stream() if stream is None else stream
    f = lib().x265hip_sao_stats
    f.argtypes = [ctypes.POINTER(SaoStatsParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_sao_stats")


def sao_apply(depth, src, src_stride, src_org, dst, dst_stride, dst_org, width, height, ctu_params, stream=None, ctu=(0, 0)):
    """SAO::generateLumaOffsets / applyPixelOffsets for every CTU, out of place; ctu_params: int32 [numCtu * 7] device tensor."""
    es = 1 if depth == 8 else 2
    p = SaoApplyParams()
    p.depth, p.src, p.src_stride = depth, src.data_ptr() + src_org * es, src_stride
    p.dst, p.dst_stride, p.width, p.height = dst.data_ptr() + dst_org * es, dst_stride, width, height
    p.ctu_params = ctu_params.data_ptr()
    p.ctu_width, p.ctu_height = ctu
    s = current_stream() if stream is None else stream
    f = lib().x265hip_sao_apply
    f.argtypes = [ctypes.POINTER(SaoApplyParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_sao_apply")


def sao_planes(depth, planes, stream=None):
    """x265hip_sao_planes: 1..3 planes through SAO statistics -> on-device parameters -> application with one launch per step.
    planes: list of dicts with src / src_stride / src_org, rec / rec_stride / rec_org, out (None: statistics only - then for every
    plane), width, height, count, offset_org, params, ctu, plane_offset."""
    es = 1 if depth == 8 else 2
    n = len(planes)
    st = (SaoStatsParams * n)()
    ap = (SaoApplyParams * n)()
    with_apply = planes[0].get("out") is not None
    for i, q in enumerate(planes):
        st[i].depth, st[i].fenc, st[i].fenc_stride = depth, q["src"].data_ptr() + q["src_org"] * es, q["src_stride"]
        st[i].rec, st[i].rec_stride, st[i].width, st[i].height = q["rec"].data_ptr() + q["rec_org"] * es, q["rec_stride"], q["width"], q["height"]
        st[i].count, st[i].offset_org = q["count"].data_ptr(), q["offset_org"].data_ptr()
        st[i].ctu_width, st[i].ctu_height, st[i].plane_offset = q["ctu"][0], q["ctu"][1], q["plane_offset"]
        if with_apply:
            ap[i].depth, ap[i].src, ap[i].src_stride = depth, st[i].rec, q["rec_stride"]
            ap[i].dst, ap[i].dst_stride, ap[i].width, ap[i].height = q["out"].data_ptr() + q["rec_org"] * es, q["rec_stride"], q["width"], q["height"]
            ap[i].ctu_params = q["params"].data_ptr()
            ap[i].ctu_width, ap[i].ctu_height = q["ctu"]
    s = current_stream() if stream is None else stream
    f = lib().x265hip_sao_planes
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    check(f(n, ctypes.cast(st, ctypes.c_void_p), ctypes.cast(ap, ctypes.c_void_p) if with_apply else None, s), "x265hip_sao_planes")


def sao_apply_planes(depth, planes, stream=None):
    """x265hip_sao_apply_planes: the application of 1..3 planes as one launch; planes as in sao_planes (with out)."""
    es = 1 if depth == 8 else 2
    n = len(planes)
    ap = (SaoApplyParams * n)()
    for i, q in enumerate(planes):
        ap[i].depth, ap[i].src, ap[i].src_stride = depth, q["rec"].data_ptr() + q["rec_org"] * es, q["rec_stride"]
        ap[i].dst, ap[i].dst_stride, ap[i].width, ap[i].height = q["out"].data_ptr() + q["rec_org"] * es, q["rec_stride"], q["width"], q["height"]
        ap[i].ctu_params = q["params"].data_ptr()
        ap[i].ctu_width, ap[i].ctu_height = q["ctu"]
    s = current_stream() if stream is None else stream
    f = lib().x265hip_sao_apply_planes
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    check(f(n, ctypes.cast(ap, ctypes.c_void_p), s), "x265hip_sao_apply_planes")


class SaoRdoParams(ctypes.Structure):
    """x265hip_sao_rdo_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("planes", ctypes.c_int), ("ctus_w", ctypes.c_int), ("ctus_h", ctypes.c_int),
                ("count", ctypes.c_void_p * 3), ("offset_org", ctypes.c_void_p * 3), ("lambda_", ctypes.c_int64 * 2), ("lambda_ctu", ctypes.c_void_p),
                ("ctx_merge", ctypes.c_int), ("ctx_type", ctypes.c_int), ("frac_bits", ctypes.c_uint32), ("entropy_bits", ctypes.c_void_p),
                ("sao_flag", ctypes.c_int * 2), ("scratch", ctypes.c_void_p), ("ctu_params", ctypes.c_void_p * 3), ("num_no_sao", ctypes.c_void_p)]


def sao_rdo_scratch_bytes(ctus_w, ctus_h):
    f = lib().x265hip_sao_rdo_scratch_bytes
    f.restype, f.argtypes = ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]
    return int(f(ctus_w, ctus_h))


def sao_rdo(depth, counts, offset_orgs, ctus_w, ctus_h, lambdas, ctx_merge, ctx_type, entropy_bits, params, scratch, lambda_ctu=None, sao_flag=(1, 1),
            num_no_sao=None, frac_bits=0, stream=None):
    """x265hip_sao_rdo: SAO::rdoSaoUnitCu over a picture.  counts / offset_orgs / params: lists (1 or 3 planes) of device tensors;
    entropy_bits: HOST numpy uint32 [128] (the host's per-state bit costs); lambdas: (luma, chroma) ints; scratch: device tensor of
    sao_rdo_scratch_bytes(ctus_w, ctus_h) bytes."""
    import numpy as np
    p = SaoRdoParams()
    p.depth, p.planes, p.ctus_w, p.ctus_h = depth, len(counts), ctus_w, ctus_h
    for i in range(len(counts)):
        p.count[i], p.offset_org[i], p.ctu_params[i] = counts[i].data_ptr(), offset_orgs[i].data_ptr(), params[i].data_ptr()
    p.lambda_[0], p.lambda_[1] = int(lambdas[0]), int(lambdas[1])
    p.lambda_ctu = _p(lambda_ctu)
    p.ctx_merge, p.ctx_type, p.frac_bits = int(ctx_merge), int(ctx_type), int(frac_bits)
    bits = np.ascontiguousarray(entropy_bits, dtype=np.uint32)
    assert bits.size == 128
    p.entropy_bits = bits.ctypes.data
    p.sao_flag[0], p.sao_flag[1] = int(sao_flag[0]), int(sao_flag[1])
    p.scratch, p.num_no_sao = scratch.data_ptr(), _p(num_no_sao)
    s = current_stream() if stream is None else stream
    f = lib().x265hip_sao_rdo
    f.argtypes = [ctypes.POINTER(SaoRdoParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_sao_rdo")


class ReconPublishParams(ctypes.Structure):
    """x265hip_recon_publish_params (include/x265hip.h)."""
    _fields_ = [("comm", ctypes.c_void_p), ("rank", ctypes.c_int), ("root", ctypes.c_int), ("peer", ctypes.c_int), ("depth", ctypes.c_int),
                ("plane", ctypes.c_void_p * 3), ("stride", ctypes.c_ssize_t), ("stride_c", ctypes.c_ssize_t), ("margin_y", ctypes.c_int),
                ("margin_y_c", ctypes.c_int), ("height", ctypes.c_int), ("ctu_row0", ctypes.c_int), ("ctu_rows", ctypes.c_int)]


def me_best_reset(best, stream=None):
    s = current_stream() if stream is None else stream
    check(lib().x265hip_me_best_reset(best.data_ptr(), best.numel(), s), "x265hip_me_best_reset")


# ------------------------------------------------------------------ generic job-list entry points
class Plane(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("stride", ctypes.c_ssize_t)]


IP_HPP, IP_HPS, IP_VPP, IP_VPS, IP_VSP, IP_VSS, IP_HVPP, IP_P2S = range(8)
TR_DCT, TR_IDCT, TR_DST4, TR_IDST4, TR_LOWPASS_DCT = range(5)
Q_QUANT, Q_NQUANT, Q_DEQUANT_NORMAL, Q_DEQUANT_SCALING, Q_DENOISE, Q_COUNT_NONZERO, Q_COPY_CNT = range(7)
INTRA_PRED, INTRA_FILTER, INTRA_ALLANGS = range(3)
(OP_COPY_PP, OP_COPY_PS, OP_COPY_SP, OP_COPY_SS, OP_SUB_PS, OP_ADD_PS, OP_ADDAVG, OP_PIXELAVG, OP_BLOCKFILL,
 OP_CPY2DTO1D_SHL, OP_CPY2DTO1D_SHR, OP_CPY1DTO2D_SHL, OP_CPY1DTO2D_SHR, OP_TRANSPOSE, OP_WEIGHT_PP, OP_WEIGHT_SP,
 OP_SCALE1D_128TO64, OP_SCALE2D_64TO32, OP_SSE_SS, OP_SSD_S, OP_VAR) = range(21)
(LF_SIGN, LF_SAO_E0, LF_SAO_E1, LF_SAO_E1_2ROWS, LF_SAO_E2, LF_SAO_E3, LF_SAO_B0, LF_STATS_BO, LF_STATS_E0, LF_STATS_E1,
 LF_STATS_E2, LF_STATS_E3, LF_DEBLOCK_LUMA_STRONG, LF_DEBLOCK_CHROMA, LF_INTEGRAL_H, LF_INTEGRAL_V, LF_ADS) = range(17)


def job_dtype():
    import numpy as np
    return np.dtype([("off", "<i8", 4), ("arg", "<i4", 4)])


def make_jobs(entries, device):
    """entries: iterable of (offsets[<=4], args[<=4]) -> device tensor of x265hip_job records."""
    import numpy as np
    import torch
    entries = list(entries)
    arr = np.zeros(len(entries), dtype=job_dtype())
    for i, (offs, args) in enumerate(entries):
        arr["off"][i, :len(offs)] = offs
        arr["arg"][i, :len(args)] = args
    return torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(device)


def make_coeff_jobs(entries, device):
    """entries: iterable of (offsets[<=5], args[<=5]) -> device tensor of x265hip_coeff_job records (64 bytes each)."""
    import numpy as np
    import torch
    entries = list(entries)
    arr = np.zeros(len(entries), dtype=np.dtype([("off", "<i8", 5), ("arg", "<i4", 5), ("reserved", "<i4")]))
    for i, (offs, args) in enumerate(entries):
        arr["off"][i, :len(offs)] = offs
        arr["arg"][i, :len(args)] = args
    return torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(device)


def plane(t, stride=0, elem_off=0, elem_size=None):
    es = t.element_size() if elem_size is None else elem_size
    return Plane(t.data_ptr() + elem_off * es if t is not None else None, stride)


def _planes(ps, n):
    arr = (Plane * n)()
    for i in range(n):
        arr[i] = ps[i] if i < len(ps) and ps[i] is not None else Plane(None, 0)
    return arr


class TuTablesRec(ctypes.Structure):
    """x265hip_tu_tables (include/x265hip.h): device pointers, any may be NULL."""
    _fields_ = [("quant_coeff", ctypes.c_void_p), ("dequant_coeff", ctypes.c_void_p), ("nr_offset", ctypes.c_void_p), ("nr_residual_sum", ctypes.c_void_p),
                ("dct_coeff_out", ctypes.c_void_p), ("delta_u_out", ctypes.c_void_p),
                ("rdoq_cost_uncoded", ctypes.c_void_p), ("rdoq_cg_cost", ctypes.c_void_p), ("rdoq_levels", ctypes.c_void_p), ("rdoq_num_sig", ctypes.c_void_p),
                ("fenc_dct_out", ctypes.c_void_p), ("psy_scale", ctypes.c_int64)]


def tu_tables(quant_coeff=None, dequant_coeff=None, nr_offset=None, nr_residual_sum=None, dct_coeff_out=None, delta_u_out=None,
              rdoq_cost_uncoded=None, rdoq_cg_cost=None, rdoq_levels=None, rdoq_num_sig=None, fenc_dct_out=None, psy_scale=0):
    """Device tensors -> the table record a TU stage takes through its `tables` field (keep the returned object alive during the launch).
    dct_coeff_out (int16) / delta_u_out (int32), shaped like the stage's levels: capture for a host-side RDOQ pass.
    rdoq_*: the data-parallel half of Quant::rdoQuant - nquant's levels / count, costUncoded per coefficient (int64) and per 4x4 group;
    psy_scale != 0 = the psy pre-pass (fenc_dct_out: the source block's transform)."""
    r = TuTablesRec(_p(quant_coeff), _p(dequant_coeff), _p(nr_offset), _p(nr_residual_sum), _p(dct_coeff_out), _p(delta_u_out),
                    _p(rdoq_cost_uncoded), _p(rdoq_cg_cost), _p(rdoq_levels), _p(rdoq_num_sig), _p(fenc_dct_out), int(psy_scale))
    r._keep = (quant_coeff, dequant_coeff, nr_offset, nr_residual_sum, dct_coeff_out, delta_u_out, rdoq_cost_uncoded, rdoq_cg_cost, rdoq_levels, rdoq_num_sig,
               fenc_dct_out)
    return r


class IntraReconParams(ctypes.Structure):
    """x265hip_intra_recon_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("n", ctypes.c_int),
                ("fenc", ctypes.c_void_p), ("fenc_stride", ctypes.c_ssize_t),
                ("nb", ctypes.c_void_p),
                ("recon", ctypes.c_void_p), ("recon_stride", ctypes.c_ssize_t),
                ("qp", ctypes.c_int), ("intra_slice", ctypes.c_int),
                ("jobs", ctypes.c_void_p), ("njobs", ctypes.c_int),
                ("levels", ctypes.c_void_p), ("num_sig", ctypes.c_void_p), ("dist", ctypes.c_void_p), ("chroma", ctypes.c_int),
                ("tables", ctypes.c_void_p)]


def intra_recon_batch(depth, n, fenc, fenc_stride, nb, recon, recon_stride, qp, intra_slice, jobs, njobs,
                      levels, num_sig, dist, stream=None, chroma=False, tables=None):
    """Intra TU candidate set (search.cpp:335-373): one (TU, mode) candidate per job, see include/x265hip.h.
    chroma=True: the 4:2:0 chroma flavour (unfiltered neighbours, no edge smoothing, DCT for 4x4).  tables: hipabi.tu_tables(...)"""
    p = IntraReconParams()
    p.tables = ctypes.addressof(tables) if tables is not None else None
    p.depth, p.n, p.chroma = depth, n, int(bool(chroma))
    p.fenc, p.fenc_stride, p.nb = fenc.data_ptr(), fenc_stride, nb.data_ptr()
    p.recon, p.recon_stride = recon.data_ptr(), recon_stride
    p.qp, p.intra_slice, p.jobs, p.njobs = qp, intra_slice, jobs.data_ptr(), njobs
    p.levels, p.num_sig, p.dist = levels.data_ptr(), num_sig.data_ptr(), dist.data_ptr()
    s = current_stream() if stream is None else stream
    f = lib().x265hip_intra_recon_batch
    f.argtypes = [ctypes.POINTER(IntraReconParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_intra_recon_batch")


def interp_batch(kind, depth, taps, w, h, src, dst, jobs, njobs, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_interp_batch
    f.argtypes = [ctypes.c_int] * 5 + [Plane, Plane, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    check(f(kind, depth, taps, w, h, src, dst, jobs.data_ptr(), njobs, s), "x265hip_interp_batch")


def transform_batch(kind, depth, n, src, dst, jobs, njobs, use_mfma=0, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_transform_batch
    f.argtypes = [ctypes.c_int] * 3 + [Plane, Plane, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    check(f(kind, depth, n, src, dst, jobs.data_ptr(), njobs, use_mfma, s), "x265hip_transform_batch")


def quant_batch(kind, planes, jobs, njobs, result=None, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_quant_batch
    f.argtypes = [ctypes.c_int, ctypes.POINTER(Plane), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    check(f(kind, _planes(planes, 4), jobs.data_ptr(), njobs, _p(result), s), "x265hip_quant_batch")


def intra_batch(kind, depth, n, src, dst, jobs, njobs, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_intra_batch
    f.argtypes = [ctypes.c_int] * 3 + [Plane, Plane, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    check(f(kind, depth, n, src, dst, jobs.data_ptr(), njobs, s), "x265hip_intra_batch")


def blockop_batch(op, depth, w, h, planes, jobs, njobs, result=None, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_blockop_batch
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(Plane), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    check(f(op, depth, w, h, _planes(planes, 3), jobs.data_ptr(), njobs, _p(result), s), "x265hip_blockop_batch")


def loopfilter_batch(kind, depth, planes, jobs, njobs, result=None, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_loopfilter_batch
    f.argtypes = [ctypes.c_int] * 2 + [ctypes.POINTER(Plane), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    check(f(kind, depth, _planes(planes, 4), jobs.data_ptr(), njobs, _p(result), s), "x265hip_loopfilter_batch")


# x265hip_frame_kind / x265hip_coeff_kind (include/x265hip.h)
FR_PLANECOPY_CP, FR_PLANECOPY_SP, FR_PLANECOPY_SP_SHL, FR_PLANECOPY_PP_SHR, FR_PLANE_CLIP_MAX, FR_SSIM_CORE, FR_SSIM_END4, FR_FIX8_PACK, FR_FIX8_UNPACK = range(9)
CF_SCAN_POS_LAST, CF_FIND_POS_FIRST_LAST, CF_COST_COEFF_NXN, CF_COST_COEFF_REMAIN, CF_COST_C1C2, CF_RDOQ_NONPSY, CF_RDOQ_PSY, CF_RDOQ_PSY_1P, CF_RDOQ_PSY_2P = range(9)


def frame_batch(kind, depth, w, h, planes, jobs, njobs, out=None, stream=None):
    """x265hip_frame_batch: the frame-level helpers (row a16); planes = two Plane records, jobs = device x265hip_job array."""
    s = current_stream() if stream is None else stream
    f = lib().x265hip_frame_batch
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(Plane), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    check(f(kind, depth, w, h, _planes(planes, 2), jobs.data_ptr(), njobs, _p(out), s), "x265hip_frame_batch")


def frame_init_lowres(depth, src, src_off_bytes, src_stride, dsts, dst_stride, width, height, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_frame_init_lowres
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    d = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in dsts])
    check(f(depth, src.data_ptr() + src_off_bytes, src_stride, d, dst_stride, width, height, s), "x265hip_frame_init_lowres")


def propagate_cost(dst, propagate_in, intra_costs, inter_costs, inv_qscales, fps_factor, length, stream=None):
    s = current_stream() if stream is None else stream
    f = lib().x265hip_propagate_cost
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    check(f(dst.data_ptr(), propagate_in.data_ptr(), intra_costs.data_ptr(), inter_costs.data_ptr(), inv_qscales.data_ptr(), float(fps_factor), length, s),
          "x265hip_propagate_cost")


def set_entropy_bits(bits128):
    """x265hip_set_entropy_bits: the host's 128 per-state CABAC bit costs (g_entropyBits / x265_entropyStateBits)."""
    import numpy as np
    b = np.ascontiguousarray(bits128, dtype=np.uint32)
    assert b.size == 128
    f = lib().x265hip_set_entropy_bits
    f.argtypes = [ctypes.c_void_p]
    check(f(b.ctypes.data), "x265hip_set_entropy_bits")


def coeff_batch(kind, depth, bufs, jobs, njobs, result=None, stream=None):
    """x265hip_coeff_batch: the RDOQ helpers (row a9); bufs = five device tensors (or None), jobs = device array of
    x265hip_coeff_job (int64 off[5], int32 arg[5], int32 reserved = 64 bytes)."""
    s = current_stream() if stream is None else stream
    f = lib().x265hip_coeff_batch
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    b = (ctypes.c_void_p * 5)(*[_p(t) for t in bufs])
    check(f(kind, depth, b, jobs.data_ptr(), njobs, _p(result), s), "x265hip_coeff_batch")


def pixelcmp_batch(kind, depth, w, h, a, a_stride, b, b_stride, njobs, out,
                   a_off=None, a_step=0, b_off=None, b_step=0, a_base=0, b_base=0, stream=None):
    es = 1 if depth == 8 else 2
    s = current_stream() if stream is None else stream
    check(lib().x265hip_pixelcmp_batch(kind, depth, w, h,
                                       a.data_ptr() + a_base * es, a_stride, _p(a_off), a_step,
                                       b.data_ptr() + b_base * es, b_stride, _p(b_off), b_step,
                                       njobs, out.data_ptr(), s), "x265hip_pixelcmp_batch")


class PhasePlanesParams(ctypes.Structure):
    """x265hip_phase_planes_params (include/x265hip.h)."""
    _fields_ = [("depth", ctypes.c_int), ("chroma", ctypes.c_int), ("src", ctypes.c_void_p), ("dst", ctypes.c_void_p),
                ("stride", ctypes.c_ssize_t), ("rows", ctypes.c_int)]


def phase_planes(depth, src, src_off_bytes, dst, stride, rows, chroma=False, stream=None):
    """x265hip_phase_planes: every fractional phase of one padded plane.  src: device tensor that holds the plane at byte offset
    src_off_bytes; dst: 15 (luma) / 63 (chroma) planes (their first 4 and last 8 rows are not produced)."""
    p = PhasePlanesParams(depth, int(bool(chroma)), src.data_ptr() + src_off_bytes, dst.data_ptr(), stride, rows)
    s = current_stream() if stream is None else stream
    f = lib().x265hip_phase_planes
    f.argtypes = [ctypes.POINTER(PhasePlanesParams), ctypes.c_void_p]
    check(f(ctypes.byref(p), s), "x265hip_phase_planes")


# ---- round 6: sub-sample cost tables (csrc/cost_kernels.hip, csrc/cost_stream.hip) ----------------------------------------------------
class CostCandidatesParams(ctypes.Structure):
    """x265hip_cost_candidates_params"""
    _fields_ = [("nctu", ctypes.c_int), ("window", ctypes.c_int), ("surf", ctypes.c_void_p), ("centres", ctypes.c_void_p),
                ("shapes", ctypes.c_int), ("candidates", ctypes.c_int), ("cand", ctypes.c_void_p), ("mv_cost", ctypes.c_void_p)]


class CostTablesParams(ctypes.Structure):
    """x265hip_cost_tables_params"""
    _fields_ = [("depth", ctypes.c_int), ("width", ctypes.c_int),
                ("stride", ctypes.c_ssize_t), ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int),
                ("stride_c", ctypes.c_ssize_t), ("margin_y_c", ctypes.c_int),
                ("ctu_row0", ctypes.c_int), ("ctu_rows", ctypes.c_int),
                ("fenc", ctypes.c_void_p * 3), ("ref", ctypes.c_void_p * 3), ("phases", ctypes.c_void_p * 3),
                ("plane_bytes", ctypes.c_size_t), ("plane_bytes_c", ctypes.c_size_t),
                ("shapes", ctypes.c_int), ("candidates", ctypes.c_int), ("subme", ctypes.c_int), ("chroma", ctypes.c_int),
                ("cand", ctypes.c_void_p), ("tables", ctypes.c_void_p), ("sad_costs", ctypes.c_int)]


class CostStreamParams(ctypes.Structure):
    """x265hip_cost_stream_params"""
    _fields_ = [("depth", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("stride", ctypes.c_ssize_t), ("margin_x", ctypes.c_int), ("margin_y", ctypes.c_int),
                ("stride_c", ctypes.c_ssize_t), ("margin_y_c", ctypes.c_int),
                ("centre_range", ctypes.c_int), ("window", ctypes.c_int),
                ("candidates", ctypes.c_int), ("shapes", ctypes.c_int), ("subme", ctypes.c_int), ("chroma", ctypes.c_int), ("sad_costs", ctypes.c_int),
                ("slots", ctypes.c_int), ("pictures", ctypes.c_int), ("views", ctypes.c_int), ("band_rows", ctypes.c_int), ("device_plus_1", ctypes.c_int)]


class CostStreamStats(ctypes.Structure):
    """x265hip_cost_stream_stats_t"""
    _fields_ = [(n, ctypes.c_uint64) for n in ("pairs_opened", "pairs_completed", "bands", "rows_served", "rows_uploaded", "failed", "stale_pairs",
                                               "views_opened", "views_shared", "lines_weighted", "us_busy", "bytes_downloaded", "bytes_uploaded", "table_bytes")]


def cost_pu_list(shapes):
    """[n, 4] x, y, w, h of the PU list (host-only: needs no device)."""
    import numpy as np
    L = lib()
    L.x265hip_cost_pu_rect.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int * 4)]
    n = L.x265hip_cost_pu_count(shapes)
    out = np.zeros((n, 4), np.int32)
    r = (ctypes.c_int * 4)()
    for i in range(n):
        check(L.x265hip_cost_pu_rect(shapes, i, ctypes.byref(r)), "x265hip_cost_pu_rect")
        out[i] = list(r)
    return out


def cost_positions(subme):
    """[n, 2] quarter-sample offsets of the position set (host-only)."""
    import numpy as np
    L = lib()
    L.x265hip_cost_positions.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    out = np.zeros((169, 2), np.int8)
    n = check(L.x265hip_cost_positions(subme, out.ctypes.data, 169), "x265hip_cost_positions")
    return out[:n].copy()


def cost_record_bytes(subme, sad_costs=0):
    return lib().x265hip_cost_record_bytes(subme, int(bool(sad_costs)))


def cost_ctu_bytes(subme, shapes, candidates, sad_costs=0):
    L = lib()
    L.x265hip_cost_ctu_bytes.restype = ctypes.c_size_t
    return L.x265hip_cost_ctu_bytes(subme, shapes, candidates, int(bool(sad_costs)))


def cost_candidates(surf, centres, nctu, window, shapes, candidates, cand, stream=None, mv_cost=None):
    L = lib()
    L.x265hip_cost_candidates.argtypes = [ctypes.POINTER(CostCandidatesParams), ctypes.c_void_p]
    p = CostCandidatesParams(nctu, window, _p(surf), _p(centres), shapes, candidates, _p(cand), _p(mv_cost))
    check(L.x265hip_cost_candidates(ctypes.byref(p), current_stream() if stream is None else stream), "x265hip_cost_candidates")


def cost_tables(depth, width, stride, margin_x, margin_y, stride_c, margin_y_c, ctu_row0, ctu_rows, fenc, ref, phases, plane_bytes, plane_bytes_c,
                shapes, candidates, subme, chroma, cand, tables, stream=None, sad_costs=0):
    """fenc / ref / phases: three device tensors each (allocation starts; chroma entries None when chroma = 0)."""
    L = lib()
    L.x265hip_cost_tables.argtypes = [ctypes.POINTER(CostTablesParams), ctypes.c_void_p]
    p = CostTablesParams()
    p.depth, p.width, p.stride, p.margin_x, p.margin_y, p.stride_c, p.margin_y_c = depth, width, stride, margin_x, margin_y, stride_c, margin_y_c
    p.ctu_row0, p.ctu_rows = ctu_row0, ctu_rows
    for i in range(3):
        p.fenc[i], p.ref[i], p.phases[i] = _p(fenc[i]), _p(ref[i]), _p(phases[i])
    p.plane_bytes, p.plane_bytes_c = plane_bytes, plane_bytes_c
    p.shapes, p.candidates, p.subme, p.chroma = shapes, candidates, subme, int(bool(chroma))
    p.cand, p.tables, p.sad_costs = _p(cand), _p(tables), int(bool(sad_costs))
    check(L.x265hip_cost_tables(ctypes.byref(p), current_stream() if stream is None else stream), "x265hip_cost_tables")
