"""Later stages of the frame pipeline (after motion search and sub-pel refinement); see pipeline.py.

  InterRecon  - fused prediction + residual coding round trip (x265hip_inter_recon; reference callers
                predict.cpp:245-265, quant.cpp:397-480,543-605, search.cpp:357-375)
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import hipabi
from .pipeline import PUS_PER_CTU, DevicePicture


class ReconParams(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int), ("level", ctypes.c_int),
        ("qp", ctypes.c_int), ("intra_slice", ctypes.c_int),
        ("fenc", ctypes.c_void_p), ("fenc_stride", ctypes.c_ssize_t),
        ("fref", ctypes.c_void_p), ("fref_stride", ctypes.c_ssize_t),
        ("recon", ctypes.c_void_p), ("recon_stride", ctypes.c_ssize_t),
        ("mv", ctypes.c_void_p), ("levels", ctypes.c_void_p), ("num_sig", ctypes.c_void_p), ("dist", ctypes.c_void_p),
        ("tables", ctypes.c_void_p),
    ]


def _tu_part(stage, ctu_row0, ctu_rows):
    import copy
    o = copy.copy(stage)
    cw = stage.w64 // 64
    c0, o.nctu, o.h64 = cw * ctu_row0, cw * ctu_rows, ctu_rows * 64
    nb, nn = stage.nblk, stage.n * stage.n
    o.levels = stage.levels[c0 * nb * nn:(c0 + o.nctu) * nb * nn]
    o.num_sig = stage.num_sig[c0 * nb:(c0 + o.nctu) * nb]
    o.dist = stage.dist[c0 * nb:(c0 + o.nctu) * nb]
    return o


class InterRecon:
    """Stage 3: for every NxN block (N = 8 << level) predict with the refined mv, transform / quantise the
    residual, reconstruct, measure SSE.  Outputs: recon plane, levels, num_sig, dist."""

    def __init__(self, nctu, w64, h64, depth, level, qp, device, intra_slice=0):
        import torch
        self.nctu, self.w64, self.h64, self.depth, self.level, self.qp, self.intra = nctu, w64, h64, depth, level, qp, intra_slice
        self.n = 8 << level
        self.nblk = (64 // self.n) ** 2
        self.levels = torch.zeros(nctu * self.nblk * self.n * self.n, dtype=torch.int16, device=device)
        self.num_sig = torch.zeros(nctu * self.nblk, dtype=torch.int32, device=device)
        self.dist = torch.zeros(nctu * self.nblk, dtype=torch.int64, device=device)
        self.tables = None          # hipabi.tu_tables(...): scaling-list coefficients / denoiser tables of this block size, or None

    def part(self, ctu_row0, ctu_rows):
        """The stage for `ctu_rows` CTU rows from `ctu_row0` (outputs = the matching slices; see pipeline.MotionSearch.part)."""
        return _tu_part(self, ctu_row0, ctu_rows)

    def algorithmic_bytes(self, bpp=1):
        """per block: source N^2 + reference patch (N+7)^2 + recon N^2 pixels, levels 2*N^2, 16 B of results"""
        n = self.n
        return self.nctu * self.nblk * ((2 * n * n + (n + 7) * (n + 7)) * bpp + 2 * n * n + 16)

    def run(self, cur: DevicePicture, ref: DevicePicture, recon_plane, mv, stream=None):
        es = 1 if self.depth == 8 else 2
        p = ReconParams()
        p.depth, p.width, p.height, p.level, p.qp, p.intra_slice = self.depth, self.w64, self.h64, self.level, self.qp, self.intra
        p.fenc, p.fenc_stride = cur.t.data_ptr() + cur.org * es, cur.stride
        p.fref, p.fref_stride = ref.t.data_ptr() + ref.org * es, ref.stride
        p.recon, p.recon_stride = recon_plane.data_ptr() + cur.org * es, cur.stride
        p.mv, p.levels, p.num_sig, p.dist = mv.data_ptr(), self.levels.data_ptr(), self.num_sig.data_ptr(), self.dist.data_ptr()
        p.tables = ctypes.addressof(self.tables) if self.tables is not None else None
        s = hipabi.current_stream() if stream is None else stream
        f = hipabi.lib().x265hip_inter_recon
        f.argtypes = [ctypes.POINTER(ReconParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(p), s), "x265hip_inter_recon")

    def checksum(self):
        import torch
        return {"levels": int(self.levels.to(torch.int64).sum().item()), "num_sig": int(self.num_sig.sum().item()),
                "dist": int(self.dist.sum().item())}


class PredWeight(ctypes.Structure):
    """x265hip_pred_weight (include/x265hip.h)."""
    _fields_ = [("present", ctypes.c_int), ("weight", ctypes.c_int), ("offset", ctypes.c_int), ("log2_denom", ctypes.c_int)]


class ReconBiParams(ctypes.Structure):
    """x265hip_recon_bi_params (include/x265hip.h)."""
    _fields_ = [("base", ReconParams), ("fref1", ctypes.c_void_p), ("mv1", ctypes.c_void_p), ("dir", ctypes.c_void_p),
                ("weight0", ctypes.POINTER(PredWeight)), ("weight1", ctypes.POINTER(PredWeight))]


class InterReconBi(InterRecon):
    """The inter TU stage for B pictures (x265hip_inter_recon_bi): per block list 0, list 1 or the average of both
    (predInterLumaShort + addAvg, predict.cpp:168-304)."""

    def run(self, cur: DevicePicture, ref0: DevicePicture, ref1: DevicePicture, recon_plane, mv0, mv1, dir_flags=None, stream=None, weights=None):
        """weights: (list 0, list 1), each None (the list has no weight table) or (present, weight, offset, log2_denom) - explicit
        weighted prediction (addWeightUni / addWeightBi); a P picture with weights is `dir` all 1 with ref1 = ref0."""
        es = 1 if self.depth == 8 else 2
        q = ReconBiParams()
        keep = []
        for name, w in zip(("weight0", "weight1"), weights or (None, None)):
            if w is not None:
                keep.append(PredWeight(*[int(v) for v in w]))
                setattr(q, name, ctypes.pointer(keep[-1]))
        p = q.base
        p.depth, p.width, p.height, p.level, p.qp, p.intra_slice = self.depth, self.w64, self.h64, self.level, self.qp, self.intra
        p.fenc, p.fenc_stride = cur.t.data_ptr() + cur.org * es, cur.stride
        p.fref, p.fref_stride = ref0.t.data_ptr() + ref0.org * es, ref0.stride
        p.recon, p.recon_stride = recon_plane.data_ptr() + cur.org * es, cur.stride
        p.mv, p.levels, p.num_sig, p.dist = mv0.data_ptr(), self.levels.data_ptr(), self.num_sig.data_ptr(), self.dist.data_ptr()
        p.tables = ctypes.addressof(self.tables) if self.tables is not None else None
        q.fref1, q.mv1 = ref1.t.data_ptr() + ref1.org * es, mv1.data_ptr()
        q.dir = None if dir_flags is None else dir_flags.data_ptr()
        s = hipabi.current_stream() if stream is None else stream
        f = hipabi.lib().x265hip_inter_recon_bi
        f.argtypes = [ctypes.POINTER(ReconBiParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(q), s), "x265hip_inter_recon_bi")


class InterReconChroma:
    """The same stage for one chroma plane of a 4:2:0 picture (x265hip_inter_recon_chroma; reference predInterChromaPixel,
    predict.cpp:304-351, + the residual round trip on half-size blocks).  Planes are flat device tensors with `stride` samples per
    row and sample (0,0) at element `org`; `qp` is the plane's quantiser QP (chroma mapping / offsets applied by the caller)."""

    def __init__(self, nctu, w64, h64, depth, level, qp, device, intra_slice=0):
        import torch
        self.nctu, self.w64, self.h64, self.depth, self.level, self.qp, self.intra = nctu, w64, h64, depth, level, qp, intra_slice
        self.n = 4 << level
        self.nblk = (32 // self.n) ** 2
        self.levels = torch.zeros(nctu * self.nblk * self.n * self.n, dtype=torch.int16, device=device)
        self.num_sig = torch.zeros(nctu * self.nblk, dtype=torch.int32, device=device)
        self.dist = torch.zeros(nctu * self.nblk, dtype=torch.int64, device=device)
        self.tables = None

    def part(self, ctu_row0, ctu_rows):
        return _tu_part(self, ctu_row0, ctu_rows)

    def params(self, fenc, fref, recon, stride, org, mv):
        es = 1 if self.depth == 8 else 2
        p = ReconParams()
        p.depth, p.width, p.height, p.level, p.qp, p.intra_slice = self.depth, self.w64, self.h64, self.level, self.qp, self.intra
        p.fenc, p.fenc_stride = fenc.data_ptr() + org * es, stride
        p.fref, p.fref_stride = fref.data_ptr() + org * es, stride
        p.recon, p.recon_stride = recon.data_ptr() + org * es, stride
        p.mv, p.levels, p.num_sig, p.dist = mv.data_ptr(), self.levels.data_ptr(), self.num_sig.data_ptr(), self.dist.data_ptr()
        p.tables = ctypes.addressof(self.tables) if self.tables is not None else None
        return p

    def run(self, fenc, fref, recon, stride, org, mv, stream=None):
        p = self.params(fenc, fref, recon, stride, org, mv)
        s = hipabi.current_stream() if stream is None else stream
        f = hipabi.lib().x265hip_inter_recon_chroma
        f.argtypes = [ctypes.POINTER(ReconParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(p), s), "x265hip_inter_recon_chroma")

    @staticmethod
    def run_pair(stages, fencs, frefs, recons, stride, org, mv, stream=None):
        """Cb and Cr (stages = the two planes' InterReconChroma objects) in one launch: x265hip_inter_recon_chroma_pair."""
        ps = [st.params(fenc, fref, recon, stride, org, mv) for st, fenc, fref, recon in zip(stages, fencs, frefs, recons)]
        s = hipabi.current_stream() if stream is None else stream
        f = hipabi.lib().x265hip_inter_recon_chroma_pair
        f.argtypes = [ctypes.POINTER(ReconParams), ctypes.POINTER(ReconParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(ps[0]), ctypes.byref(ps[1]), s), "x265hip_inter_recon_chroma_pair")


class InterReconChromaBi(InterReconChroma):
    """One chroma plane of a B picture (or of a P picture with explicit weights): x265hip_inter_recon_chroma_bi - per block list 0,
    list 1 or both, with the plane's own weights (Predict::motionCompensation's chroma half, predict.cpp:77-243, 304-409)."""

    def run(self, fenc, fref0, fref1, recon, stride, org, mv0, mv1, dir_flags=None, stream=None, weights=None):
        q = ReconBiParams()
        q.base = self.params(fenc, fref0, recon, stride, org, mv0)          # copied into the record
        es = 1 if self.depth == 8 else 2
        q.fref1, q.mv1 = fref1.data_ptr() + org * es, mv1.data_ptr()
        q.dir = None if dir_flags is None else dir_flags.data_ptr()
        keep = []
        for name, w in zip(("weight0", "weight1"), weights or (None, None)):
            if w is not None:
                keep.append(PredWeight(*[int(v) for v in w]))
                setattr(q, name, ctypes.pointer(keep[-1]))
        s = hipabi.current_stream() if stream is None else stream
        f = hipabi.lib().x265hip_inter_recon_chroma_bi
        f.argtypes = [ctypes.POINTER(ReconBiParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(q), s), "x265hip_inter_recon_chroma_bi")


def extend_border_rows(plane, pic: DevicePicture, top: bool, bottom: bool, stream=None, chroma=False):
    """Row-wise border extension of the band `pic` is a view of (x265hip_extend_border_rows): left / right margins of the band's rows, the
    picture's top margin when the band is its first, the bottom margin when it is its last."""
    from . import frames as F
    es = 1 if pic.depth == 8 else 2
    s = hipabi.current_stream() if stream is None else stream
    f = hipabi.lib().x265hip_extend_border_rows
    f.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    if chroma:
        my = F.CHROMA_MARGIN_Y
        hipabi.check(f(plane.data_ptr() + pic.org_c * es, pic.stride_c, pic.w64 // 2, pic.h64 // 2, F.CHROMA_MARGIN_X, my if top else 0, my if bottom else 0,
                       pic.depth, s), "x265hip_extend_border_rows")
        return
    my = F.MARGIN_Y
    hipabi.check(f(plane.data_ptr() + pic.org * es, pic.stride, pic.w64, pic.h64, F.MARGIN_X, my if top else 0, my if bottom else 0, pic.depth, s),
                 "x265hip_extend_border_rows")


def extend_border(plane, pic: DevicePicture, stream=None, chroma=False):
    """Replicate the picture edges into the margins of `plane` (same geometry as `pic`; chroma: one of its 4:2:0 planes), on device."""
    from . import frames as F
    es = 1 if pic.depth == 8 else 2
    s = hipabi.current_stream() if stream is None else stream
    f = hipabi.lib().x265hip_extend_border
    f.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    if chroma:
        hipabi.check(f(plane.data_ptr() + pic.org_c * es, pic.stride_c, pic.w64 // 2, pic.h64 // 2, F.CHROMA_MARGIN_X, F.CHROMA_MARGIN_Y, pic.depth, s),
                     "x265hip_extend_border")
        return
    hipabi.check(f(plane.data_ptr() + pic.org * es, pic.stride, pic.w64, pic.h64, F.MARGIN_X, F.MARGIN_Y, pic.depth, s),
                 "x265hip_extend_border")


class BorderPlane(ctypes.Structure):
    _fields_ = [("pic", ctypes.c_void_p), ("stride", ctypes.c_ssize_t), ("width", ctypes.c_int), ("height", ctypes.c_int), ("margin_x", ctypes.c_int),
                ("margin_top", ctypes.c_int), ("margin_bottom", ctypes.c_int)]


def extend_border_picture(planes, pic: DevicePicture, stream=None, top=True, bottom=True):
    """extend_border of [Y, Cb, Cr] (or [Y]) of one picture as ONE launch (x265hip_extend_border_planes); `pic` a band view with top / bottom =
    whether it is the picture's first / last band: the row-wise form (extend_border_rows) of all planes at once."""
    from . import frames as F
    es = 1 if pic.depth == 8 else 2
    s = hipabi.current_stream() if stream is None else stream
    arr = (BorderPlane * len(planes))()
    arr[0] = BorderPlane(planes[0].data_ptr() + pic.org * es, pic.stride, pic.w64, pic.h64, F.MARGIN_X, F.MARGIN_Y if top else 0, F.MARGIN_Y if bottom else 0)
    for i in range(1, len(planes)):
        arr[i] = BorderPlane(planes[i].data_ptr() + pic.org_c * es, pic.stride_c, pic.w64 // 2, pic.h64 // 2, F.CHROMA_MARGIN_X,
                             F.CHROMA_MARGIN_Y if top else 0, F.CHROMA_MARGIN_Y if bottom else 0)
    f = hipabi.lib().x265hip_extend_border_planes
    f.argtypes = [ctypes.POINTER(BorderPlane), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    hipabi.check(f(arr, len(planes), pic.depth, s), "x265hip_extend_border_planes")


# g_chromaScale (constants.cpp:346-350) as Quant::setChromaQP applies it to 4:2:0 pictures (quant.cpp:233-244)
_CHROMA_SCALE = list(range(30)) + [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37] + list(range(38, 52)) + [51] * 12


def chroma_quant_qp(qp, depth, offset=0):
    """The chroma quantiser's QP for a CU whose luma quantiser runs at `qp` (= CU QP + QP_BD_OFFSET)."""
    bd = 6 * (depth - 8)
    q = min(max(qp - bd + offset, -bd), 57)
    if q >= 30:
        q = _CHROMA_SCALE[q]
    return q + bd


class Deblock:
    """In-loop deblocking of the luma reconstruction on device (x265hip_deblock_bs_inter + x265hip_deblock_luma; reference
    Deblock::getBoundaryStrength / edgeFilterLuma, deblock.cpp:191-215, 317-415) for the reconstruction stage's block grid."""

    def __init__(self, w64, h64, depth, level, qp, device):
        import torch
        self.w64, self.h64, self.depth, self.level, self.qp = w64, h64, depth, level, qp
        self.bs_ver = torch.zeros((h64 // 4) * (w64 // 8), dtype=torch.uint8, device=device)
        self.bs_hor = torch.zeros((h64 // 8) * (w64 // 4), dtype=torch.uint8, device=device)

    def run(self, plane, pic: DevicePicture, mv, num_sig):
        hipabi.deblock_bs_inter(self.w64, self.h64, self.level, mv, num_sig, self.bs_ver, self.bs_hor)
        hipabi.deblock_luma(self.depth, plane, pic.stride, pic.org, self.w64, self.h64, self.bs_ver, self.bs_hor, self.qp)


class Sao:
    """Sample adaptive offset of the deblocked luma picture: the two pixel passes on device (x265hip_sao_stats /
    x265hip_sao_apply; reference SAO::calcSaoStatsCTU sao.cpp:735-917 and generateLumaOffsets / applyPixelOffsets :572-630,
    :274-570).  The parameter choice between them (rdoSaoUnitCu, entropy-coder bit counts) is host work: `stats()` fills
    count / offset_org [numCtu, 5, 32]; `apply(params)` takes int32 [numCtu, 7] = typeIdx, bandPos, offset[4], mergeLeft."""

    def __init__(self, width, height, depth, device, ctu=(64, 64), plane_offset=0):
        """width / height: the PLANE's size; ctu: the CTU's footprint in it (4:2:0 chroma: (32, 32) with plane_offset 2)."""
        import torch
        self.width, self.height, self.depth, self.ctu, self.plane_offset = width, height, depth, ctu, plane_offset
        self.nctu = ((width + ctu[0] - 1) // ctu[0]) * ((height + ctu[1] - 1) // ctu[1])
        self.count = torch.zeros(self.nctu * 160, dtype=torch.int32, device=device)
        self.offset_org = torch.zeros(self.nctu * 160, dtype=torch.int32, device=device)
        self.params = torch.zeros(self.nctu * 7, dtype=torch.int32, device=device)

    def stats(self, src, rec, rec_stride, rec_org, src_plane=None):
        """src: a DevicePicture (its luma plane) or, with src_plane, any plane of the same stride / origin as the reconstruction."""
        if src_plane is not None:
            hipabi.sao_stats(self.depth, src_plane, rec_stride, rec_org, rec, rec_stride, rec_org, self.width, self.height, self.count, self.offset_org,
                             ctu=self.ctu, plane_offset=self.plane_offset)
            return
        hipabi.sao_stats(self.depth, src.t, src.stride, src.org, rec, rec_stride, rec_org, self.width, self.height, self.count, self.offset_org,
                         ctu=self.ctu, plane_offset=self.plane_offset)

    def plane(self, src, src_stride, src_org, rec, rec_stride, rec_org, out=None):
        """This plane's record for hipabi.sao_planes (several planes, one launch per SAO step)."""
        return dict(src=src, src_stride=src_stride, src_org=src_org, rec=rec, rec_stride=rec_stride, rec_org=rec_org, out=out, width=self.width,
                    height=self.height, count=self.count, offset_org=self.offset_org, params=self.params, ctu=self.ctu, plane_offset=self.plane_offset)

    def decide(self):
        """saoStatsInitialOffset + the distortion-only type choice of x265hip_sao_decide -> self.params (stays on the device)."""
        hipabi.sao_decide(self.depth, self.count, self.offset_org, self.nctu, self.params)

    def apply(self, rec, rec_stride, rec_org, out, params=None):
        hipabi.sao_apply(self.depth, rec, rec_stride, rec_org, out, rec_stride, rec_org, self.width, self.height,
                         self.params if params is None else params, ctu=self.ctu)


class Lookahead:
    """Lookahead picture preparation + intra cost estimate on device (x265hip_lowres_init / x265hip_lowres_intra; reference
    Lowres::init lowres.cpp:294-306 and LookaheadTLD::lowresIntraEstimate slicetype.cpp:696-772).  Geometry follows
    Lowres::create (lowres.cpp:50-72): half resolution rounded up to whole 8x8 blocks, the full-resolution margins; the
    row stride is sized for the ROUNDED width so the right margin never runs into the next row (the reference sizes it
    from the unrounded width - a layout choice, not a result)."""

    def __init__(self, width, height, depth, device, intra_penalty=5):
        import torch
        from . import frames as F
        self.depth, self.penalty = depth, intra_penalty
        self.wcu, self.hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
        self.width, self.lines = self.wcu * 8, self.hcu * 8
        self.mx, self.my = F.MARGIN_X, F.MARGIN_Y
        self.stride = (self.width + 2 * self.mx + 31) & ~31
        self.org = self.stride * self.my + self.mx
        dt = torch.uint8 if depth == 8 else torch.int16
        self.planes = [torch.zeros(self.stride * (self.lines + 2 * self.my), dtype=dt, device=device) for _ in range(4)]
        n = self.wcu * self.hcu
        self.intra_cost = torch.zeros(n, dtype=torch.int32, device=device)
        self.intra_mode = torch.zeros(n, dtype=torch.uint8, device=device)
        self.lowres_costs = torch.zeros(n, dtype=torch.int16, device=device)

    def run(self, pic: DevicePicture):
        hipabi.lowres_init(self.depth, pic.t, pic.stride, pic.org, self.planes, self.stride, self.org,
                           self.width, self.lines, self.mx, self.my)
        hipabi.lowres_intra(self.depth, self.planes[0], self.stride, self.org, self.wcu, self.hcu, self.penalty,
                            self.intra_cost, self.intra_mode, self.lowres_costs)

    def checksum(self):
        import torch
        return {"intra_cost": int(self.intra_cost.sum(dtype=torch.int64).item()), "intra_mode": int(self.intra_mode.sum(dtype=torch.int64).item())}


class AdaptiveQuant:
    """The adaptive-quantisation pass of the lookahead - LookaheadTLD::calcAdaptiveQuantFrame (slicetype.cpp:439-694; AQ modes 0-3,
    no hevcAq / edge mode / HDR10 / per-block quant offsets).  The pixel work (every block's AC energy through the var primitive, the
    picture's wp_sum / wp_ssd totals) is one device launch (x265hip_aq_energy); the QP offsets are the reference's double-precision
    expressions, evaluated by the library's host-side x265hip_aq_offsets through the C library's pow / log2 like the reference."""

    def __init__(self, width, height, depth, device, qg_size=16, aq_mode=2, aq_strength=1.0, weightp=True):
        import torch
        self.width, self.height, self.depth, self.qg = width, height, depth, qg_size
        self.mode, self.strength, self.weightp = aq_mode, float(aq_strength), bool(weightp)
        self.bw, self.bh = (width + qg_size - 1) // qg_size, (height + qg_size - 1) // qg_size
        self.energy = torch.zeros(self.bw * self.bh, dtype=torch.int32, device=device)
        self.wp = torch.zeros(6, dtype=torch.int64, device=device)

    def run(self, y: DevicePicture, cb=None, cr=None, stride_c=0, org_c=0):
        """y: the source luma picture; cb / cr: device chroma planes (4:2:0) or None.  Returns (qp_aq_offset float64 [blocks],
        inv_qscale int32 [blocks], wp_sum [3], wp_ssd [3]) as Lowres holds them afterwards."""
        import numpy as np
        hipabi.aq_energy(self.depth, y.t, y.stride, y.org, self.width, self.height, self.qg, self.energy, self.wp, cb, cr, stride_c, org_c)
        energy = self.energy.cpu().numpy().view(np.uint32)
        wp = self.wp.cpu().numpy().view(np.uint64)
        qp, inv = hipabi.aq_offsets(self.depth, self.qg, self.mode, self.strength, energy)
        wp_sum = [int(v) for v in wp[:3]]
        wp_ssd = [int(v) for v in wp[3:]]
        if self.weightp:                                             # the final normalisation, slicetype.cpp:662-675
            col, row = ((self.width + 8) >> 4) << 4, ((self.height + 8) >> 4) << 4
            dims = [col * row, (col >> 1) * (row >> 1), (col >> 1) * (row >> 1)]
            m64 = 0xffffffffffffffff                                 # `sum * sum` wraps in uint64_t in the reference (bright 4K 10/12-bit)
            wp_ssd = [(wp_ssd[i] - ((wp_sum[i] * wp_sum[i] + dims[i] // 2) & m64) // dims[i]) & m64 for i in range(3)]
        return qp, inv, wp_sum, wp_ssd


class HevcAq:
    """--hevc-aq: the pass calcAdaptiveQuantFrame runs instead of the AQ modes - LookaheadTLD::xPreanalyze / xPreanalyzeQp
    (slicetype.cpp:293-441).  Per enabled layer (lowres.h:123-129, 64x64 CTUs) one launch produces the quadrant sums of every partition
    (x265hip_aq_hevc_quadrants); activities, the layer's average and the QP offsets are the reference's double-precision expressions in
    the library's host-side x265hip_aq_hevc_offsets; the wp_sum / wp_ssd statistics of the same loop come from x265hip_aq_energy."""
    LAYERS = {64: (1, 0, 1, 0), 32: (1, 1, 1, 0), 16: (1, 1, 1, 0), 8: (1, 1, 1, 1)}

    def __init__(self, width, height, depth, device, qg_size=16, qp_adaptation_range=1.0, weightp=True):
        import torch
        self.width, self.height, self.depth, self.qg, self.range = width, height, depth, qg_size, float(qp_adaptation_range)
        self.parts = [64 >> d for d in range(4) if self.LAYERS[qg_size][d]]
        self.sums = [torch.zeros(8 * ((width + p - 1) // p) * ((height + p - 1) // p), dtype=torch.int64, device=device) for p in self.parts]
        self.wp_pass = AdaptiveQuant(width, height, depth, device, qg_size=8 if qg_size == 8 else 16, aq_mode=0, weightp=weightp)

    def run(self, y: DevicePicture, cb=None, cr=None, stride_c=0, org_c=0):
        """Returns ({partition size: (activity, qp_offset, avg_activity)}, inv_qscale of the deepest layer, wp_sum [3], wp_ssd [3])."""
        import numpy as np
        for p, sm in zip(self.parts, self.sums):
            hipabi.aq_hevc_quadrants(self.depth, y.t, y.stride, y.org, self.width, self.height, p, sm)
        _, _, wp_sum, wp_ssd = self.wp_pass.run(y, cb, cr, stride_c, org_c)
        layers, inv = {}, None
        for p, sm in zip(self.parts, self.sums):
            act, qp, avg, inv = hipabi.aq_hevc_offsets(self.width, self.height, p, self.range, sm.cpu().numpy().view(np.uint64).reshape(-1, 4, 2))
            layers[p] = (act, qp, avg)
        return layers, inv, wp_sum, wp_ssd


class WeightAnalysis:
    """Weighted-reference analysis of a prepared picture against a prepared reference - LookaheadTLD::weightsAnalyse
    (slicetype.cpp:860-957).  The pixel work runs on the device (x265hip_lowres_weight_cost scores the unweighted reference and the
    offset candidate in ONE launch - the second candidate does not depend on the first score -, x265hip_lowres_weight_apply weights
    the four planes); the float guess in between is evaluated here in single precision, operation by operation as the reference
    writes it.  wp_ssd / wp_sum are the pictures' luma statistics (Lowres::wp_ssd[0] / wp_sum[0], left there by the reference's
    adaptive-quantisation pass)."""

    def __init__(self, la: Lookahead, device):
        import torch
        self.la, self.depth = la, la.depth
        self.cost = torch.zeros(4, dtype=torch.int32, device=device)
        self.weighted = [torch.zeros_like(p) for p in la.planes]          # the reference's wbuffer[0..3]

    @property
    def weighted_ref(self):
        """The weighted planes in the shape LookaheadCost takes a reference in: estimateFrameCost searches them instead of the
        reference's own planes once a weight is accepted (wfref0, slicetype.cpp:3222,3267; P pictures - the bi-directional
        candidates of B pictures keep the unweighted planes, :3328)."""
        from types import SimpleNamespace
        return SimpleNamespace(planes=self.weighted)

    def analyse(self, cur: Lookahead, ref: Lookahead, wp_ssd, wp_sum):
        """wp_ssd / wp_sum: (current, reference).  Returns (weight or None, minscore, origscore); with a weight, self.weighted holds
        the weighted reference planes (weightedRef.isWeighted)."""
        import numpy as np
        f32 = np.float32
        depth = self.depth
        guess = np.sqrt(f32(int(wp_ssd[0])) / f32(int(wp_ssd[1]))) if (wp_ssd[0] and wp_ssd[1]) else f32(1.0)
        npx = f32(cur.lines * cur.width)
        fenc_mean = f32(int(wp_sum[0])) / npx / f32(1 << (depth - 8))
        ref_mean = f32(int(wp_sum[1])) / npx / f32(1 << (depth - 8))
        if abs(ref_mean - fenc_mean) < f32(0.5) and abs(f32(1.0) - guess) < f32(1.0 / 128.0):
            return None, 0, 0
        # WeightParam::setFromWeightAndOffset(w, 0, 7, true), slice.h:304-316
        w, mindenom = int(guess * f32(128) + f32(0.5)), 7
        while mindenom > 0 and w > 127:
            mindenom -= 1
            w >>= 1
        minscale, minoff, found = min(w, 127), 0, False
        cur_scale = minscale
        cur_off = int(fenc_mean - ref_mean * f32(cur_scale) / f32(1 << mindenom) + f32(0.5))
        if cur_off < -128 or cur_off > 127:
            cur_off = max(-128, min(127, cur_off))
            cur_scale = int(f32(1 << mindenom) * (fenc_mean - f32(cur_off)) / ref_mean + f32(0.5))
            cur_scale = max(0, min(127, cur_scale))
        # origscore: wp.wtPresent is still 0 at :907 - the UNWEIGHTED cost; then the offset candidate (:925)
        hipabi.lowres_weight_cost(depth, cur.planes[0], ref.planes[0], cur.stride, cur.org, cur.width, cur.lines, cur.intra_cost,
                                  [None, (cur_scale, mindenom, cur_off)], self.cost)
        scores = self.cost[:2].cpu().numpy().view(np.uint32)
        origscore = minscore = int(scores[0])
        if not minscore:
            return None, minscore, origscore
        s = int(scores[1])
        if s < minscore:
            minscore, minscale, minoff, found = s, cur_scale, cur_off, True
        if mindenom > 0 and not (minscale & 1):
            idx = 32 if not minscale else (minscale & -minscale).bit_length() - 1        # CTZ
            shift = min(idx, mindenom)
            mindenom -= shift
            minscale >>= shift
        if (not found) or (minscale == 1 << mindenom and minoff == 0) or f32(minscore) / f32(origscore) > f32(0.998):
            return None, minscore, origscore
        weight = (minscale, mindenom, minoff)
        rows = cur.lines + 2 * cur.my
        hipabi.lowres_weight_apply(depth, ref.planes, self.weighted, cur.stride, rows, weight)
        return weight, minscore, origscore


class LookaheadCost:
    """The lookahead's frame cost estimate of a prepared picture against one (P) or two (B) prepared references
    (x265hip_lowres_cost; reference CostEstimateGroup::estimateFrameCost / estimateCUCost, slicetype.cpp:3115-3388).  `lam` is
    x265_lambda_tab[X265_LOOKAHEAD_QP] (1.0 for 8-bit, 16.0 for 10-bit); the mv cost table is built on the host like BitCost's
    (bitcost.cpp:51-55,103-118)."""

    def __init__(self, la: Lookahead, device, lam=None, bidir=False):
        import numpy as np
        import torch
        from . import frames as F
        self.depth, self.bidir = la.depth, bidir
        lam = (1.0 if la.depth == 8 else (16.0 if la.depth == 10 else 64.0)) if lam is None else lam
        cq, self.qoff = F.qpel_cost_table(16, lam=lam, qmax=4 * (max(la.width, la.lines) + 64))
        self.cost_q = torch.from_numpy(cq.view(np.int16)).to(device)
        n = la.wcu * la.hcu
        self.mvs = torch.zeros(n * 2, dtype=torch.int32, device=device)
        self.mv_costs = torch.zeros(n, dtype=torch.int32, device=device)
        self.mvs1 = torch.zeros(n * 2, dtype=torch.int32, device=device) if bidir else None
        self.mv_costs1 = torch.zeros(n, dtype=torch.int32, device=device) if bidir else None
        self.lowres_costs = torch.zeros(n, dtype=torch.int16, device=device)
        self.row_satds = torch.zeros(la.hcu, dtype=torch.int32, device=device)
        self.frame = torch.zeros(4, dtype=torch.int64, device=device)

    def pair(self, cur: Lookahead, ref: Lookahead, ref1: Lookahead = None, do_search=(1, 1), ref_bi: Lookahead = None):
        """ref_bi: --weightp on a B picture - `ref` then carries the weighted list-0 planes (WeightAnalysis.weighted_ref) and ref_bi the
        reference's own planes, which the bi-directional candidates keep (slicetype.cpp:3328)."""
        return hipabi.lowres_cost_pair(self.depth, cur.org, cur.planes[0], ref.planes, cur.intra_cost, self.mvs, self.mv_costs,
                                       self.lowres_costs, self.row_satds, self.frame,
                                       ref1_planes=None if ref1 is None else ref1.planes, mvs1=self.mvs1, mv_costs1=self.mv_costs1,
                                       do_search=do_search, ref_bi_planes=None if ref_bi is None else ref_bi.planes)

    def run(self, cur: Lookahead, ref: Lookahead, ref1: Lookahead = None, do_search=(1, 1), bframe_bias=0, stream=None, ref_bi: Lookahead = None):
        hipabi.lowres_cost(self.depth, cur.stride, cur.wcu, cur.hcu, self.cost_q, self.qoff, [self.pair(cur, ref, ref1, do_search, ref_bi)],
                           bframe_bias=bframe_bias, stream=stream)

    @staticmethod
    def run_batch(stages, curs, refs, refs1=None, bframe_bias=0, stream=None, device_table=None):
        """One launch for many independent pictures of one geometry and kind (one workgroup each).  device_table: a table made by
        `table()` for exactly these operands - the launch then involves no host-to-device copy."""
        s0, c0 = stages[0], curs[0]
        refs1 = [None] * len(stages) if refs1 is None else refs1
        pairs = None if device_table is not None else [s.pair(c, r, r1) for s, c, r, r1 in zip(stages, curs, refs, refs1)]
        hipabi.lowres_cost(s0.depth, c0.stride, c0.wcu, c0.hcu, s0.cost_q, s0.qoff, pairs, bframe_bias=bframe_bias, stream=stream,
                           device_table=device_table)

    @staticmethod
    def table(stages, curs, refs, device, refs1=None, stream=None):
        refs1 = [None] * len(stages) if refs1 is None else refs1
        return hipabi.lowres_cost_table([s.pair(c, r, r1) for s, c, r, r1 in zip(stages, curs, refs, refs1)], device, stream=stream)


class PatternSearch:
    """Motion search drivers for every 8x8..64x64 PU of every CTU (x265hip_me_search; reference
    MotionEstimate::motionEstimate, motion.cpp:739-1561) with predictor (0,0): integer pattern + sub-pel refinement in
    one launch.  Output has SubpelRefine's layout: int32 [ctu*85][2] = {cost, qmvx | qmvy << 16}."""

    def __init__(self, w64, h64, depth, method, subme, merange, device, lam=4.0):
        import numpy as np
        import torch
        from . import frames as F
        self.depth, self.method, self.subme, self.merange = depth, method, subme, merange
        jobs = []
        for cy in range(0, h64, 64):
            for cx in range(0, w64, 64):
                for n in (8, 16, 32, 64):
                    npu = (64 // n) ** 2
                    for z in range(npu):
                        bx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4)
                        by = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)
                        jobs.append((cx + bx * n, cy + by * n, n, n, 0, 0, 0, 0, 0))
        self.njobs = len(jobs)
        arr = np.array(jobs, dtype=hipabi.me_search_job_dtype())
        self.jobs = torch.from_numpy(arr.view(np.int32).reshape(-1, 9).copy()).to(device)
        cq, self.qoff = F.qpel_cost_table(merange, lam, qmax=8 * (merange + 8) + 64)
        self.cost_q = torch.from_numpy(cq.view(np.int16)).to(device)
        self.out = torch.zeros(self.njobs * 2, dtype=torch.int32, device=device)
        self.w64, self.h64, self.planes = w64, h64, None

    def run(self, cur: DevicePicture, ref: DevicePicture):
        from . import frames as F
        r = self.merange
        integral = None
        if self.method == hipabi.ME_SEA:
            # the reference picture's block-sum planes, rebuilt per reference like FrameFilter does after reconstruction
            integral = hipabi.sea_integral(self.depth, ref.t, ref.stride, ref.org, self.w64, self.h64, F.MARGIN_X, F.MARGIN_Y, planes=self.planes)
            self.planes = integral[0]
        hipabi.me_search(self.depth, cur.t, cur.stride, cur.org, ref.t, ref.stride, ref.org, self.method, self.subme, r,
                         self.cost_q, self.qoff, (-r, -r), (r, r), self.jobs, self.njobs, integral=integral)
        o = self.out.view(-1, 2)
        o[:, 0] = self.jobs[:, 8]
        o[:, 1] = (self.jobs[:, 6] & 0xffff) | (self.jobs[:, 7] << 16)

    def checksum(self):
        import torch
        return {"subpel": int(self.out.to(dtype=torch.int64).sum().item())}


class FramePipeline:
    """Closed-loop frame pipeline: every frame is searched in, predicted from and reconstructed against the
    RECONSTRUCTION of the previous frame (like the reference's P-frame chain), all on device:
        ME (exhaustive, SAD surfaces and/or best mv) -> sub-pel refinement -> prediction + residual round trip
        (NxN blocks) -> border extension -> the reconstruction becomes the next reference.
    With lookahead=(width, height) the source picture also goes through the lookahead stage (half-resolution planes + intra
    cost estimate per 8x8 block), which only depends on the source."""

    def __init__(self, w64, h64, depth, device, rng=57, subme=2, level=2, qp=27, want_surf=True, packed=False, lookahead=None,
                 search="full", deblock=False, sao=False, lookahead_cost_batch=0, chroma=False, sao_apply=False, sign_hide=False,
                 subpel_planes=False, parallel_planes=False, split=1, sao_rdo=None):
        """split (with parallel_planes): the picture goes through search -> sub-pel refinement -> reconstruction in `split` parts of whole
        CTU rows; while the search of part k + 1 runs on the caller's stream, part k is refined and reconstructed on a side stream.  No
        CTU's result depends on another CTU before the loop filters, which still run on the whole picture: same outputs, and only the last
        part's refinement + reconstruction stay behind the search on the critical path.  Measured at 4K 8-bit (profiles/r02_split.txt):
        2.94 ms per picture with 2 parts against 2.29 ms in one piece - the record-per-lane search kernel wants every CU's LDS and both of
        its workgroup slots, and workgroups of other kernels resident next to it cost it more than the overlap hides.  Off by default."""
        import torch
        from .pipeline import MotionSearch, SubpelRefine
        self.depth = depth
        # parallel_planes: after the sub-pel stage the three planes are independent chains (reconstruction -> deblocking -> SAO ->
        # border) of small, latency-bound launches: Cb and Cr run on their own HIP streams next to Y, the lookahead (source picture
        # only) next to the search; everything joins the caller's stream before run() returns
        self.parallel = bool(parallel_planes)
        self.pstreams = None
        self._spare_ready = None
        self.schedule2 = os.environ.get("X265HIP_PAR_SCHEDULE", "2") != "1"          # 1: the round-2 schedule (A/B runs)
        self.split = max(1, min(int(split), h64 // 64))
        self.parts = None
        # the three planes' SAO passes as one launch per step: on by default where every launch is on one stream (0.144 vs 0.187 ms at 4K);
        # with the planes' chains on side streams the per-plane passes already overlap and fusing them costs 0.07 ms ("2" forces it there too)
        self.fuse_sao = os.environ.get("X265HIP_FUSE_SAO", "1") != "0"
        self.fuse_sao_parallel = os.environ.get("X265HIP_FUSE_SAO", "1") == "2"
        # experiment switches of the parallel-planes launch structure (read once): where the lookahead and the phase planes run
        self.la_after_search = os.environ.get("X265HIP_LA_AFTER_ME", "1") == "1"
        self.prep_next_to_search = os.environ.get("X265HIP_PREP_OVERLAP", "0") == "1"
        self.ms = MotionSearch(w64, h64, rng, depth, device, want_surf=want_surf and search == "full", want_best=True, packed=packed)
        # subpel_planes: sub-pel candidates read from the reference picture's phase planes (one x265hip_phase_planes launch per frame)
        self.sp = SubpelRefine(self.ms, subme, device, phase_planes=subpel_planes)
        # search != "full": the pattern-search drivers replace the exhaustive search + sub-pel pair
        self.ps = None
        if search != "full":
            method = {"dia": hipabi.ME_DIA, "hex": hipabi.ME_HEX, "umh": hipabi.ME_UMH, "star": hipabi.ME_STAR, "sea": hipabi.ME_SEA}[search]
            self.ps = PatternSearch(w64, h64, depth, method, subme, rng, device)
        # sign_hide: Quant::signBitHidingHDQ after the quantiser (pps.bSignHideEnabled, the x265 default)
        self.tu_flags = hipabi.TU_SIGN_HIDE if sign_hide else 0
        self.rc = InterRecon(self.ms.nctu, w64, h64, depth, level, qp, device, intra_slice=self.tu_flags)
        self.la = Lookahead(lookahead[0], lookahead[1], depth, device) if lookahead else None
        # The lookahead's P-frame cost estimate runs AHEAD of the encode like the reference's lookahead thread: every
        # `lookahead_cost_batch` frames one launch on a side stream scores that many (picture, previous picture) pairs.  The
        # prepared pictures live in a ring so that a slot is only rewritten after the launches that read it.
        self.lcb = lookahead_cost_batch if lookahead else 0
        if self.lcb:
            self.ring = [self.la] + [Lookahead(lookahead[0], lookahead[1], depth, device) for _ in range(2 * self.lcb)]
            self.lc = [LookaheadCost(self.la, device) for _ in range(self.lcb)]
            self.side = torch.cuda.Stream(device=device)
            self.frame_no, self.pending, self.read_done = 0, [], {}
            # the operand tables of the launches repeat with the ring: upload each once, up front for the full batches
            self.device, self.tables = device, {}
            nring = len(self.ring)
            for j in range(nring):
                self._table(tuple(((j * self.lcb + 1 + i) % nring, (j * self.lcb + i) % nring) for i in range(self.lcb)))
            torch.cuda.synchronize()
        # `qp` is the quantiser's QP (qp + QP_BD_OFFSET, what transformNxN works with); the deblocking tables are indexed with the
        # CU's own QP (m_qp, 0..51)
        self.db = Deblock(w64, h64, depth, level, max(qp - 6 * (depth - 8), 0), device) if deblock else None
        # SAO statistics of the deblocked reconstruction (what rdoSaoUnitCu reads); the offsets themselves are the host's decision
        self.sao = Sao(w64, h64, depth, device) if sao else None
        self.recon = None
        # 4:2:0 chroma through the same closed loop (predInterChromaPixel + residual round trip, edgeFilterChroma, SAO on 32x32 footprints)
        self.chroma = chroma
        self.qp = qp
        self.recon_c, self.out, self.out_c = None, None, None
        if chroma:
            qpc = chroma_quant_qp(qp, depth)
            self.rc_c = [InterReconChroma(self.ms.nctu, w64, h64, depth, level, qpc, device, intra_slice=self.tu_flags) for _ in range(2)]
            self.sao_c = [Sao(w64 // 2, h64 // 2, depth, device, ctu=(32, 32), plane_offset=2) for _ in range(2)] if sao else None
        # SAO applied in the loop: the parameters come from x265hip_sao_decide (initial offsets + distortion-only choice), so the
        # picture handed to the next frame is deblocked AND offset like a decoder's
        self.sao_apply = bool(sao and sao_apply)
        # sao_rdo: dict(lambdas=(luma, chroma), ctx_merge, ctx_type, entropy_bits) - the parameters come from x265hip_sao_rdo, the
        # reference's own rate-distortion decision (SAO::rdoSaoUnitCu) on all planes' statistics, instead of the distortion-only stand-in
        self.sao_rdo = sao_rdo if (sao and sao_apply) else None
        if self.sao_rdo is not None:
            self.sao_scratch = torch.zeros(hipabi.sao_rdo_scratch_bytes(w64 // 64, h64 // 64), dtype=torch.uint8, device=device)
            self.sao_no = torch.zeros(2, dtype=torch.int32, device=device)
        # band mode (BandedFramePipeline): (is_first_band, is_last_band) -> row-wise border extension instead of the whole-picture one
        self.band_border = None

    def _sao_rdo(self):
        """x265hip_sao_rdo on the statistics of all planes -> every plane's Sao.params."""
        st = [self.sao] + (list(self.sao_c) if self.chroma else [])
        q = self.sao_rdo
        hipabi.sao_rdo(self.depth, [x.count for x in st], [x.offset_org for x in st], self.ms.w64 // 64, self.ms.h64 // 64, q["lambdas"], q["ctx_merge"], q["ctx_type"],
                       q["entropy_bits"], [x.params for x in st], self.sao_scratch, sao_flag=(1, 1 if self.chroma else 0), num_no_sao=self.sao_no)

    def _sao_planes(self, cur):
        return [self.sao.plane(cur.t, cur.stride, cur.org, self.recon, cur.stride, cur.org, self.out)] + \
               ([self.sao_c[i].plane(cur.c[i], cur.stride_c, cur.org_c, self.recon_c[i], cur.stride_c, cur.org_c, self.out_c[i]) for i in range(2)] if self.chroma else [])

    def run(self, cur: DevicePicture, ref: DevicePicture, mark=None):
        """ref: the current reference picture (extended borders; with chroma=True also its Cb / Cr planes); returns the luma plane of
        the picture for the next frame's reference list (final_planes() has all three).  mark(name), if given, is called after every
        stage (bench.py records an event there)."""
        import torch
        par = self.parallel and mark is None and self.chroma and not self.lcb and self.ps is None and self.db is not None and self.sao is not None \
            and self.sao_apply
        if par:
            return self._run_parallel(cur, ref)
        mark = mark or (lambda name: None)
        if self.recon is None:
            self.recon = torch.zeros_like(cur.t)
        if self.lcb:
            slot = self.frame_no % len(self.ring)
            if slot in self.read_done:                       # the cost launches that read this slot must be through
                torch.cuda.current_stream().wait_event(self.read_done.pop(slot))
            self.la = self.ring[slot]
            self.la.run(cur)
            if self.frame_no:
                self.pending.append((slot, (self.frame_no - 1) % len(self.ring)))
            self.frame_no += 1
            if len(self.pending) == self.lcb:
                self.launch_lookahead_costs()
        elif self.la is not None:
            self.la.run(cur)
        if self.ps is None:
            self.ms.reset()                                  # 4 us fill of best[]: kept out of the search kernel's interval
        mark("lookahead")
        if self.ps is not None:
            self.ps.run(cur, ref)
            mv = self.ps.out
            mark("me")
        else:
            self.ms.search(cur, ref)                         # ONE fused launch: SAD surfaces + best mv
            mark("me")
            self.sp.run(cur, ref)
            mv = self.sp.out
        mark("subpel")
        self.rc.run(cur, ref, self.recon, mv)
        mark("recon")
        if self.chroma:
            if self.recon_c is None:
                self.recon_c = [torch.zeros_like(p) for p in cur.c]
            if self.rc_c[0].tables is None and self.rc_c[1].tables is None:
                InterReconChroma.run_pair(self.rc_c, cur.c, ref.c, self.recon_c, cur.stride_c, cur.org_c, mv)      # Cb + Cr: one launch
            else:
                for i in range(2):
                    self.rc_c[i].run(cur.c[i], ref.c[i], self.recon_c[i], cur.stride_c, cur.org_c, mv)
            mark("recon_chroma")
        if self.db is not None:
            self.db.run(self.recon, cur, mv, self.rc.num_sig)
            if self.chroma:                  # Bs 2 edges only (intra CUs): none in an all-inter picture, the pass still runs like the reference's
                hipabi.deblock_chroma(self.depth, self.recon_c[0], self.recon_c[1], cur.stride_c, cur.org_c, cur.w64, cur.h64,
                                      self.db.bs_ver, self.db.bs_hor, self.db.qp)
            mark("deblock")
        final, final_c = self.recon, self.recon_c
        if self.sao_rdo is not None:
            # statistics of all planes (one launch) -> the reference's rate-distortion decision (x265hip_sao_rdo: it needs every plane's
            # statistics, Cb / Cr share a type) -> application of all planes (one launch)
            if self.out is None:
                self.out = torch.zeros_like(cur.t)
                self.out_c = [torch.zeros_like(p) for p in cur.c] if self.chroma else None
            planes = self._sao_planes(cur)
            hipabi.sao_planes(self.depth, [dict(q, out=None) for q in planes])
            mark("sao_stats")
            self._sao_rdo()
            mark("sao_rdo")
            hipabi.sao_apply_planes(self.depth, planes)
            final, final_c = self.out, self.out_c
            mark("sao_apply")
        elif self.sao is not None and self.sao_apply and self.chroma and self.fuse_sao:
            # Y, Cb, Cr through statistics -> parameters -> application with ONE launch per step (x265hip_sao_planes)
            if self.out is None:
                self.out = torch.zeros_like(cur.t)
                self.out_c = [torch.zeros_like(p) for p in cur.c]
            hipabi.sao_planes(self.depth, [self.sao.plane(cur.t, cur.stride, cur.org, self.recon, cur.stride, cur.org, self.out)] +
                              [self.sao_c[i].plane(cur.c[i], cur.stride_c, cur.org_c, self.recon_c[i], cur.stride_c, cur.org_c, self.out_c[i]) for i in range(2)])
            final, final_c = self.out, self.out_c
            mark("sao")                      # statistics + parameters + application of the three planes: three launches
        elif self.sao is not None:
            self.sao.stats(cur, self.recon, cur.stride, cur.org)
            if self.chroma:
                for i in range(2):
                    self.sao_c[i].stats(None, self.recon_c[i], cur.stride_c, cur.org_c, src_plane=cur.c[i])
            mark("sao_stats")
            if self.sao_apply:
                if self.out is None:
                    self.out = torch.zeros_like(cur.t)
                    self.out_c = [torch.zeros_like(p) for p in cur.c] if self.chroma else None
                self.sao.decide()
                self.sao.apply(self.recon, cur.stride, cur.org, self.out)
                final = self.out
                if self.chroma:
                    for i in range(2):
                        self.sao_c[i].decide()
                        self.sao_c[i].apply(self.recon_c[i], cur.stride_c, cur.org_c, self.out_c[i])
                    final_c = self.out_c
                mark("sao_apply")
        one_launch = os.environ.get("X265HIP_BORDER_PLANES", "1") != "0"          # round 6: the picture's (band's) planes in one launch
        if self.band_border is not None:
            top, bottom = self.band_border
            if one_launch:
                extend_border_picture([final] + (list(final_c) if self.chroma else []), cur, top=top, bottom=bottom)
            else:
                extend_border_rows(final, cur, top, bottom)
                if self.chroma:
                    for i in range(2):
                        extend_border_rows(final_c[i], cur, top, bottom, chroma=True)
        elif one_launch:
            extend_border_picture([final] + (list(final_c) if self.chroma else []), cur)
        else:
            extend_border(final, cur)
            if self.chroma:
                for i in range(2):
                    extend_border(final_c[i], cur, chroma=True)
        mark("border")
        self.final, self.final_c = final, final_c
        return final

    def _run_parallel(self, cur: DevicePicture, ref: DevicePicture):
        """The same launches as run(), Y on the caller's stream, Cb / Cr on two side streams from the reconstruction on, the lookahead on
        a third; all joined before returning (stage outputs are identical: no stage shares an output buffer with another plane)."""
        import torch
        main = torch.cuda.current_stream()
        if self.pstreams is None:
            # (a higher HIP priority for the stream that carries the chroma chain - it ends after the luma chain - moved nothing: 2.189 ms
            # against 2.163 without, profiles/r03_step_timeline.txt)
            self.pstreams = [torch.cuda.Stream() for _ in range(4)]
        sCb, sCr, sLa = self.pstreams[:3]
        if self.recon is None:
            self.recon = torch.zeros_like(cur.t)
        if self.recon_c is None:
            self.recon_c = [torch.zeros_like(p) for p in cur.c]
        if self.out is None:
            self.out = torch.zeros_like(cur.t)
            self.out_c = [torch.zeros_like(p) for p in cur.c]
        if self.sao_rdo is not None and self.split == 1 and self.la_after_search and self.schedule2 and self.band_border is None:
            return self._run_parallel2(cur, ref, main, sCb)
        start = torch.cuda.Event(); start.record(main)
        # The lookahead of the source picture only depends on the source, but it does not run next to the search: the record-per-lane search
        # kernel loses more to any co-resident kernel than that kernel takes (see split below); next to the latency-bound stages behind the
        # search it is free - 2.20 against 2.25 ms per 4K picture (X265HIP_LA_AFTER_ME=0: the old placement)
        la_after_me = self.la is not None and self.split == 1 and self.la_after_search
        if self.la is not None and not la_after_me:
            sLa.wait_event(start)
            with torch.cuda.stream(sLa):
                self.la.run(cur)
        # the reference's phase planes need only the reference and could run next to the search (X265HIP_PREP_OVERLAP=1), but their 141 MB
        # of plane writes compete with the search's record stream: measured 2.34 ms per step against 2.30 with the planes after the search
        overlap_prep = self.prep_next_to_search
        if overlap_prep:
            sCb.wait_event(start)
            with torch.cuda.stream(sCb):
                self.sp.prepare(ref)
                ev_pl = torch.cuda.Event(); ev_pl.record(sCb)
        self.ms.reset()
        mv = self.sp.out
        if self.split > 1:
            ev_rec = self._search_to_recon_in_parts(cur, ref, main, sCb, sCr, start)
        else:
            self.ms.search(cur, ref)
            if la_after_me:                  # the lookahead next to the stages behind the search instead of next to the search
                ev_me = torch.cuda.Event(); ev_me.record(main)
                sLa.wait_event(ev_me)
                with torch.cuda.stream(sLa):
                    self.la.run(cur)
            if overlap_prep:
                main.wait_event(ev_pl)
            self.sp.run(cur, ref, prepared=overlap_prep)
            ev_mv = torch.cuda.Event(); ev_mv.record(main)
            # reconstruction: one plane per stream
            self.rc.run(cur, ref, self.recon, mv)
            ev_rec = []
            for i, st in enumerate((sCb, sCr)):
                st.wait_event(ev_mv)
                with torch.cuda.stream(st):
                    self.rc_c[i].run(cur.c[i], ref.c[i], self.recon_c[i], cur.stride_c, cur.org_c, mv)
                    e = torch.cuda.Event(); e.record(st); ev_rec.append(e)
        # deblocking: boundary strengths + luma on the caller's stream; the chroma pass (both planes, one entry point) on Cb's stream
        self.db.run(self.recon, cur, mv, self.rc.num_sig)
        ev_bs = torch.cuda.Event(); ev_bs.record(main)
        sCb.wait_event(ev_bs); sCb.wait_event(ev_rec[1])
        with torch.cuda.stream(sCb):
            hipabi.deblock_chroma(self.depth, self.recon_c[0], self.recon_c[1], cur.stride_c, cur.org_c, cur.w64, cur.h64,
                                  self.db.bs_ver, self.db.bs_hor, self.db.qp)
            ev_dbc = torch.cuda.Event(); ev_dbc.record(sCb)
        sCr.wait_event(ev_dbc)
        if self.sao_rdo is not None:
            # the decision couples the planes: statistics of Y on the caller's stream and of Cb / Cr on theirs, the decision + the application
            # of all three on the caller's stream, the border extensions back on the side streams
            self.sao.stats(cur, self.recon, cur.stride, cur.org)
            evs = []
            for i, st in enumerate((sCb, sCr)):
                with torch.cuda.stream(st):
                    self.sao_c[i].stats(None, self.recon_c[i], cur.stride_c, cur.org_c, src_plane=cur.c[i])
                    e = torch.cuda.Event(); e.record(st); evs.append(e)
            for e in evs:
                main.wait_event(e)
            self._sao_rdo()
            hipabi.sao_apply_planes(self.depth, self._sao_planes(cur))
            ev_sao = torch.cuda.Event(); ev_sao.record(main)
        elif self.fuse_sao_parallel:
            # SAO of the three planes: one launch per step on the caller's stream; only the border extensions go back to the side streams
            main.wait_event(ev_dbc)
            hipabi.sao_planes(self.depth, [self.sao.plane(cur.t, cur.stride, cur.org, self.recon, cur.stride, cur.org, self.out)] +
                              [self.sao_c[i].plane(cur.c[i], cur.stride_c, cur.org_c, self.recon_c[i], cur.stride_c, cur.org_c, self.out_c[i]) for i in range(2)])
            ev_sao = torch.cuda.Event(); ev_sao.record(main)
        else:
            # SAO + border extension per plane
            self.sao.stats(cur, self.recon, cur.stride, cur.org)
            self.sao.decide()
            self.sao.apply(self.recon, cur.stride, cur.org, self.out)

        def border(plane, chroma):
            if self.band_border is not None:              # a band of a picture: side margins, top / bottom margin only at the picture's edge
                extend_border_rows(plane, cur, self.band_border[0], self.band_border[1], chroma=chroma)
            else:
                extend_border(plane, cur, chroma=chroma)
        border(self.out, False)
        done = []
        for i, st in enumerate((sCb, sCr)):
            if self.fuse_sao_parallel or self.sao_rdo is not None:
                st.wait_event(ev_sao)
            with torch.cuda.stream(st):
                if not self.fuse_sao_parallel and self.sao_rdo is None:
                    self.sao_c[i].stats(None, self.recon_c[i], cur.stride_c, cur.org_c, src_plane=cur.c[i])
                    self.sao_c[i].decide()
                    self.sao_c[i].apply(self.recon_c[i], cur.stride_c, cur.org_c, self.out_c[i])
                border(self.out_c[i], True)
                e = torch.cuda.Event(); e.record(st); done.append(e)
        if self.la is not None:
            e = torch.cuda.Event(); e.record(sLa); done.append(e)
        for e in done:
            main.wait_event(e)
        self.final, self.final_c = self.out, self.out_c
        return self.final

    def _run_parallel2(self, cur, ref, main, sC):
        """The launch schedule of the default bench step since round 3, read off the dispatch timeline of one step
        (tools/gpu_visit.sh timeline, profiles/r03_step_timeline.txt).  Same launches and outputs as run(); what moved:
          * Cb + Cr are ONE chain of pair launches on one side stream (reconstruction pair, deblocking of both planes, statistics of both
            planes): two chains of half-size launches next to the luma chain each ran 1.8x their stand-alone time and ended 50 us after it;
          * the lookahead of the source picture runs on that stream BEHIND the chroma statistics, i.e. next to the SAO decision - a serial
            pass on ONE compute unit (115 us with the other 255 idle) - instead of next to the phase planes / sub-pel refinement, which it
            slowed; the minima of the NEXT picture's search are cleared there too (MotionSearch.reset_spare);
          * ONE side stream: every cross-stream dependency costs the waiting stream ~8 us on this runtime, and a step on three side streams
            had eight of them on the caller's stream - now three records and two waits."""
        import torch
        ms = self.ms
        if self._spare_ready:                             # cleared next to the previous picture's stages, on the stream the caller joined last
            ms.swap_best()
        else:
            ms.reset()
        if self.prep_next_to_search:                      # X265HIP_PREP_OVERLAP=1 (experiment): the reference's phase planes next to the search, late in it
            ev0 = torch.cuda.Event(); ev0.record(main)
            sC.wait_event(ev0)
            with torch.cuda.stream(sC):
                self.sp.prepare(ref)
                ev_pl = torch.cuda.Event(); ev_pl.record(sC)
        ms.search(cur, ref)
        if self.prep_next_to_search:
            main.wait_event(ev_pl)
        self.sp.run(cur, ref, prepared=self.prep_next_to_search)
        mv = self.sp.out
        ev_mv = torch.cuda.Event(); ev_mv.record(main)
        self.rc.run(cur, ref, self.recon, mv)
        sC.wait_event(ev_mv)
        with torch.cuda.stream(sC):
            InterReconChroma.run_pair(self.rc_c, cur.c, ref.c, self.recon_c, cur.stride_c, cur.org_c, mv)
        self.db.run(self.recon, cur, mv, self.rc.num_sig)
        ev_bs = torch.cuda.Event(); ev_bs.record(main)
        planes = self._sao_planes(cur)
        sC.wait_event(ev_bs)
        with torch.cuda.stream(sC):
            hipabi.deblock_chroma(self.depth, self.recon_c[0], self.recon_c[1], cur.stride_c, cur.org_c, cur.w64, cur.h64,
                                  self.db.bs_ver, self.db.bs_hor, self.db.qp)
            hipabi.sao_planes(self.depth, [dict(q, out=None) for q in planes[1:3]])
            ev_cstats = torch.cuda.Event(); ev_cstats.record(sC)
            ms.reset_spare()                              # the sub-pel stage has long read this picture's minima: the other buffer may be cleared
            self._spare_ready = True
            if self.la is not None:
                self.la.run(cur)
            ev_side = torch.cuda.Event(); ev_side.record(sC)
        hipabi.sao_planes(self.depth, [dict(planes[0], out=None)])
        main.wait_event(ev_cstats)
        self._sao_rdo()
        hipabi.sao_apply_planes(self.depth, planes)
        if os.environ.get("X265HIP_BORDER_PLANES", "1") == "0":       # A/B: three launches back to back (15 us) instead of one
            extend_border(self.out, cur)
            for i in range(2):
                extend_border(self.out_c[i], cur, chroma=True)
        else:
            extend_border_picture([self.out] + list(self.out_c), cur)
        main.wait_event(ev_side)                          # the lookahead and the cleared minima (long done: next to the SAO decision)
        self.final, self.final_c = self.out, self.out_c
        return self.final

    def swap_output(self, spare):
        """Ping-pong of the decoded picture: hands out the planes the last run() wrote (the next reference) and takes `spare` - [Y, Cb, Cr]
        planes the host no longer needs, e.g. the reference that has just been replaced - as the destination of the next run().  What a
        decoded-picture buffer does instead of copying a picture per frame.  None when the last output is not a plane set of its own."""
        if self.final is not self.out or self.out is None or (self.chroma and self.final_c is not self.out_c):
            return None
        outs = self.final_planes()
        self.out = spare[0]
        if self.chroma:
            self.out_c = list(spare[1:3])
        return outs

    def _search_to_recon_in_parts(self, cur, ref, main, sCb, sCr, start):
        """Search on `main`, part after part; a part's sub-pel refinement + luma reconstruction follow on a fourth stream, its chroma
        reconstructions on the Cb / Cr streams - except the last part's luma chain, which continues on `main` (the loop filters wait for it
        there anyway).  Returns the events after which Cb / Cr are reconstructed; `main` has waited for every luma part."""
        import torch
        if self.parts is None:
            rows = self.ms.h64 // 64
            cuts = [rows * k // self.split for k in range(self.split + 1)]
            self.parts = []
            for r0, r1 in zip(cuts[:-1], cuts[1:]):
                msp = self.ms.part(r0, r1 - r0)
                self.parts.append((r0, r1 - r0, msp, self.sp.part(msp, r0), self.rc.part(r0, r1 - r0), [q.part(r0, r1 - r0) for q in self.rc_c]))
        sY = self.pstreams[3]
        sY.wait_event(start)
        with torch.cuda.stream(sY):                       # the reference's phase planes: next to the first part's search
            self.sp.prepare(ref)
        last = len(self.parts) - 1
        for k, (r0, n, msp, spp, rcp, rcc) in enumerate(self.parts):
            c, r = cur.band_view(r0, n), ref.band_view(r0, n)
            msp.search(c, r)
            ev_me = torch.cuda.Event(); ev_me.record(main)
            st = main if k == last else sY
            if k == last:
                main.wait_stream(sY)                      # the phase planes (and the earlier parts' luma chains)
            else:
                sY.wait_event(ev_me)
            with torch.cuda.stream(st):
                spp.run(c, r, prepared=True)
                ev_mv = torch.cuda.Event(); ev_mv.record(st)
                rcp.run(c, r, self.recon, spp.out)
            for i, sc in enumerate((sCb, sCr)):
                sc.wait_event(ev_mv)
                with torch.cuda.stream(sc):
                    rcc[i].run(cur.c[i], ref.c[i], self.recon_c[i], cur.stride_c, c.org_c, spp.out)
        ev_rec = []
        for sc in (sCb, sCr):
            e = torch.cuda.Event(); e.record(sc); ev_rec.append(e)
        return ev_rec

    def final_planes(self):
        """[luma, cb, cr] of the picture the last run() produced for the next frame's reference list."""
        return [self.final] + (list(self.final_c) if self.chroma else [])

    def _table(self, key):
        if key not in self.tables:
            n = len(key)
            self.tables[key] = LookaheadCost.table(self.lc[:n], [self.ring[c] for c, _ in key], [self.ring[r] for _, r in key], self.device)
        return self.tables[key]

    def launch_lookahead_costs(self):
        """Score the pending (picture, previous picture) pairs in one launch on the side stream."""
        import torch
        if not self.lcb or not self.pending:
            return
        ready = torch.cuda.Event()
        ready.record()                                       # the pictures were prepared on the current stream
        self.side.wait_event(ready)
        n = len(self.pending)
        LookaheadCost.run_batch(self.lc[:n], [self.ring[c] for c, _ in self.pending], [self.ring[r] for _, r in self.pending],
                                stream=self.side.cuda_stream, device_table=self._table(tuple(self.pending)))
        done = torch.cuda.Event()
        done.record(self.side)
        for c, r in self.pending:
            self.read_done[c] = done
            self.read_done[r] = done
        self.pending = []

    def checksum(self):
        import torch
        out = {}
        if self.lcb:
            out["lookahead_cost"] = int(sum(int(s.frame[0].item()) for s in self.lc))
        out.update(self.ps.checksum() if self.ps is not None else self.sp.checksum())
        out.update(self.rc.checksum())
        if self.la is not None:
            out.update(self.la.checksum())
        if self.sao is not None:
            out["sao_count"] = int(self.sao.count.sum(dtype=torch.int64).item())
            out["sao_offset_org"] = int(self.sao.offset_org.sum(dtype=torch.int64).item())
        out["recon"] = int(self.final.view(torch.uint8).to(torch.int64).sum().item())
        if self.chroma:
            out["recon_cb"] = int(self.final_c[0].view(torch.uint8).to(torch.int64).sum().item())
            out["recon_cr"] = int(self.final_c[1].view(torch.uint8).to(torch.int64).sum().item())
            out["levels_c"] = int(sum(r.levels.to(torch.int64).sum().item() for r in self.rc_c))
        if self.sao_apply:
            out["sao_types"] = int((self.sao.params.view(-1, 7)[:, 0] >= 0).sum().item())
        return out


class BandedFramePipeline:
    """The frame pipeline band by band - the reference's `--slices` picture: bands of `band_rows` CTU rows, each searched, reconstructed
    and loop-filtered as a slice of its own (deblocking and SAO stop at the band boundary like the reference's slices: m_cuAbove is NULL
    for the first row of a slice, cudata.cpp:319; SAO's firstRowInSlice / lastRowInSlice, sao.cpp:286-287), while the exhaustive search,
    sub-pel refinement and prediction read the whole reference picture around the band.  A finished band (filtered, side margins
    extended) can be handed to the rank that searches it next while the following bands are still in flight - the granularity of the
    reference's m_reconRowFlag (framefilter.cpp:664).  band_ready(b, row0, rows), if given, is called after band b's launches."""

    def __init__(self, w64, h64, depth, device, band_rows=4, lookahead=None, graphs=False, streams=1, **kw):
        """graphs: a band's ~25 launches are captured once per (band, source picture, reference picture) into a HIP graph and replayed with
        one launch - a band is 240 workgroups per kernel, so enqueueing its launches one by one from the host costs more than running them
        (3.9 ms per 4K frame of 9 bands against 2.2 ms for the frame in one piece).  The buffers must then stay where they are between frames."""
        self.w64, self.h64, self.depth, self.device = w64, h64, depth, device
        self.use_graphs, self.graphs, self.warm = bool(graphs), {}, set()
        rows = h64 // 64
        self.bands = [(r, min(band_rows, rows - r)) for r in range(0, rows, band_rows)]
        kw.pop("lookahead_cost_batch", None)
        # options that make a stage read or write WHOLE pictures per call have no banded meaning: every band would recompute the whole
        # reference's phase planes, and in ring mode read reference rows that have not arrived yet (round-2 advisor finding)
        # (parallel_planes - the Cb / Cr chains of a band on their own streams - IS a per-band option: tests/test_gpu_banded.py)
        for opt in ("subpel_planes", "split"):
            if kw.get(opt):
                raise ValueError(f"BandedFramePipeline: {opt} is a whole-picture option and cannot be forwarded to the band pipelines")
        # streams > 1: band b runs on HIP stream b % streams with its own set of stage buffers.  The bands of one picture do not depend on
        # each other (each is a slice), so the exhaustive search of band b + 1 - the one launch that fills the chip - overlaps the
        # reconstruction / deblocking / SAO launches of band b, which at band size are a few dozen workgroups each and bound by their own
        # latency.  begin_frame() forks the band streams from the caller's stream, end_frame() joins them.
        self.nstreams = max(1, int(streams))
        self.pipe_sets = [{n: FramePipeline(w64, n * 64, depth, device, lookahead=None, **kw) for n in sorted({n for _, n in self.bands})}
                          for _ in range(self.nstreams)]
        self.pipes = self.pipe_sets[0]
        self.streams = None
        self.la = Lookahead(lookahead[0], lookahead[1], depth, device) if lookahead else None
        self.planes = None          # [Y, Cb, Cr] the bands are written into: reconstruction and filtered picture
        self.chroma = bool(kw.get("chroma"))
        self.sao_apply = bool(kw.get("sao") and kw.get("sao_apply"))

    def _alloc(self, cur):
        import torch
        if self.planes is None:
            self.recon = [torch.zeros_like(p) for p in cur.planes()]
            self.final = [torch.zeros_like(p) for p in cur.planes()] if self.sao_apply else self.recon
        for pipe in [q for ps in self.pipe_sets for q in ps.values()]:
            pipe.recon = self.recon[0]
            pipe.recon_c = self.recon[1:3] if self.chroma else None
            if self.sao_apply:
                pipe.out = self.final[0]
                pipe.out_c = self.final[1:3] if self.chroma else None
        self.planes = self.final

    def begin_frame(self, cur):
        """Per-frame work that does not depend on the reference: output planes, the lookahead stage of the source picture."""
        self._alloc(cur)
        if self.nstreams > 1:
            import torch
            if self.streams is None:
                self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.nstreams)]
            ev = torch.cuda.Event()
            ev.record()
            for st in self.streams:          # whatever the caller queued before (the hand-off of the previous picture) comes first
                st.wait_event(ev)
        if self.la is not None:
            self.la.run(cur)

    def band_context(self, b):
        """Context manager under which band b's launches (and the transfers that wait for / follow them) are issued."""
        import contextlib
        if self.nstreams == 1 or self.streams is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.streams[b % self.nstreams])

    def end_frame(self):
        """The caller's stream continues after every band of the picture."""
        if self.nstreams > 1 and self.streams is not None:
            import torch
            main = torch.cuda.current_stream()
            for st in self.streams:
                main.wait_stream(st)

    def _launch_band(self, b, cur, ref):
        row0, n = self.bands[b]
        pipe = self.pipe_sets[b % self.nstreams][n]
        pipe.band_border = (b == 0, b == len(self.bands) - 1)
        pipe.run(cur.band_view(row0, n), ref.band_view(row0, n))

    def run_band(self, b, cur, ref):
        if not self.use_graphs:
            return self._launch_band(b, cur, ref)
        import torch
        n = (self.bands[b][1], b % self.nstreams)
        if n not in self.warm:                     # the first band of this height runs launch by launch: lazy allocations, kernel attributes
            self.warm.add(n)
            return self._launch_band(b, cur, ref)
        key = (b,) + tuple(p.data_ptr() for p in cur.planes()) + tuple(p.data_ptr() for p in ref.planes())
        g = self.graphs.get(key)
        if g is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):       # other threads (RCCL's watchdog) keep working during the capture
                self._launch_band(b, cur, ref)
            self.graphs[key] = g
        g.replay()

    def capture(self, cur, ref):
        """Record every band of (cur, ref) ahead of time (set-up, outside any timed region).  Executes the bands once."""
        self.begin_frame(cur)
        for _ in range(2):
            for b in range(len(self.bands)):
                with self.band_context(b):
                    self.run_band(b, cur, ref)
        self.end_frame()

    def run(self, cur, ref, band_ready=None, before_band=None):
        self.begin_frame(cur)
        for b, (row0, n) in enumerate(self.bands):
            with self.band_context(b):
                if before_band is not None:
                    before_band(b, row0, n)                 # e.g. wait until the reference rows this band reads have arrived
                self.run_band(b, cur, ref)
                if band_ready is not None:
                    band_ready(b, row0, n)
        self.end_frame()
        return self.planes[0]

    def final_planes(self):
        return list(self.planes)
