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
