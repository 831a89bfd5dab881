"""Later stages of the frame pipeline (after motion search and sub-pel refinement); see pipeline.py.

  InterRecon  - fused prediction + residual coding round trip (x265hip_inter_recon; reference callers
                predict.cpp:245-265, quant.cpp:397-480,543-605, search.cpp:357-375)
"""
from __future__ import annotations

import ctypes

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
    ]


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
        s = hipabi.current_stream() if stream is None else stream
        f = hipabi.lib().x265hip_inter_recon
        f.argtypes = [ctypes.POINTER(ReconParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(p), s), "x265hip_inter_recon")

    def checksum(self):
        import torch
        return {"levels": int(self.levels.to(torch.int64).sum().item()), "num_sig": int(self.num_sig.sum().item()),
                "dist": int(self.dist.sum().item())}


def extend_border(plane, pic: DevicePicture, stream=None):
    """Replicate the picture edges into the margins of `plane` (same geometry as `pic`), on device."""
    from . import frames as F
    es = 1 if pic.depth == 8 else 2
    s = hipabi.current_stream() if stream is None else stream
    f = hipabi.lib().x265hip_extend_border
    f.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    hipabi.check(f(plane.data_ptr() + pic.org * es, pic.stride, pic.w64, pic.h64, F.MARGIN_X, F.MARGIN_Y, pic.depth, s),
                 "x265hip_extend_border")


class FramePipeline:
    """Closed-loop frame pipeline: every frame is searched in, predicted from and reconstructed against the
    RECONSTRUCTION of the previous frame (like the reference's P-frame chain), all on device:
        ME (exhaustive, SAD surfaces and/or best mv) -> sub-pel refinement -> prediction + residual round trip
        (NxN blocks) -> border extension -> the reconstruction becomes the next reference."""

    def __init__(self, w64, h64, depth, device, rng=57, subme=2, level=2, qp=27, want_surf=True, packed=False):
        import torch
        from .pipeline import MotionSearch, SubpelRefine
        self.depth = depth
        self.ms = MotionSearch(w64, h64, rng, depth, device, want_surf=want_surf, want_best=True, packed=packed)
        self.sp = SubpelRefine(self.ms, subme, device)
        self.rc = InterRecon(self.ms.nctu, w64, h64, depth, level, qp, device)
        self.recon = None

    def run(self, cur: DevicePicture, ref: DevicePicture):
        """ref.t is the current reference plane (extended borders); returns the plane holding the new reconstruction."""
        import torch
        if self.recon is None:
            self.recon = torch.zeros_like(cur.t)
        self.ms.run(cur, ref)
        self.sp.run(cur, ref)
        self.rc.run(cur, ref, self.recon, self.sp.out)
        extend_border(self.recon, cur)
        return self.recon

    def checksum(self):
        import torch
        out = {}
        out.update(self.sp.checksum())
        out.update(self.rc.checksum())
        out["recon"] = int(self.recon.view(torch.uint8).to(torch.int64).sum().item())
        return out
