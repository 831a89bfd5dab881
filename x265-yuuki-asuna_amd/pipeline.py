"""Frame pipeline driver (tier T2 of SURVEY.md section 7): issues the encoder's block-primitive
work for whole frames as batched HIP launches, everything resident in HBM.

It is NOT an HEVC encoder: it runs the data-parallel stages the reference's callers
(motion.cpp / search.cpp / quant.cpp / framefilter.cpp) spend their time in, in the same
dependency order, through the C ABI of libx265hip.so, and produces integer results (SAD surfaces,
motion vectors, ...) that the tests compare bit-for-bit with the oracle.

Stages implemented so far
  ME   exhaustive integer motion search for every 8x8..64x64 PU of every CTU
       (x265hip_me_fullsearch; reference caller motion.cpp:1397-1445, primitives pu[].sad/sad_x4)
"""
from __future__ import annotations

import numpy as np

from . import frames as F
from . import hipabi

LEVEL_SIZES = (8, 16, 32, 64)
LEVEL_PUS = (64, 16, 4, 1)
LEVEL_BASE = (0, 64, 80, 84)
PUS_PER_CTU = 85


class DevicePicture:
    """A padded picture plane in HBM (layout: frames.padded_dims)."""

    def __init__(self, img: np.ndarray, device, cb: np.ndarray = None, cr: np.ndarray = None):
        import torch
        buf, self.stride, self.org, self.w64, self.h64 = F.pad_plane(img)
        self.depth = 8 if img.dtype == np.uint8 else 10

        def up(b):
            return torch.from_numpy(b if b.dtype == np.uint8 else b.view(np.int16)).to(device)   # torch has no uint16 arithmetic; raw bits only
        self.t = up(buf)
        self.host = buf
        # optional 4:2:0 chroma planes (flat tensors, PicYuv chroma geometry)
        self.c, self.c_host, self.stride_c, self.org_c = None, None, 0, 0
        if cb is not None:
            pads = [F.pad_chroma(p, self.w64, self.h64) for p in (cb, cr)]
            self.stride_c, self.org_c = pads[0][1], pads[0][2]
            self.c_host = [p[0] for p in pads]
            self.c = [up(p[0]).reshape(-1) for p in pads]

    def planes(self):
        """[luma, cb, cr] device tensors (luma only without chroma)."""
        return [self.t] + (self.c or [])

    def like(self, planes):
        """A picture of the same geometry over other device planes (e.g. a reconstruction that becomes a reference)."""
        o = DevicePicture.__new__(DevicePicture)
        o.__dict__.update(self.__dict__)
        o.t = planes[0]
        o.c = list(planes[1:3]) if len(planes) >= 3 else None
        o.host, o.c_host = None, None
        return o


class MotionSearch:
    """Owns the output buffers of the ME stage for one picture size.

    surf : int32 [ctu][mvy][mvx/4][85][4]  (85 = 64 8x8 + 16 16x16 + 4 32x32 + 1 64x64 PUs, z-order;
           mv columns in groups of 4, last group padded - the pad column holds unspecified values);
           packed=True (8-bit): 720-byte groups, uint16 for the 8x8 / 16x16 levels (X265HIP_SURF_PACKED)
    best : int64 [ctu][85]            cost << 32 | raster mv index
    """

    def __init__(self, w64, h64, rng, depth, device, want_surf=True, want_best=True, lam=4.0, packed=False):
        import torch
        self.w64, self.h64, self.range, self.depth = w64, h64, rng, depth
        self.nctu = (w64 // 64) * (h64 // 64)
        self.nc = 2 * rng + 1
        self.ng = (self.nc + 3) // 4
        self.packed = bool(packed and want_surf)
        self.group_bytes = hipabi.SURF_GROUP_BYTES_PACKED if self.packed else hipabi.SURF_GROUP_BYTES_I32
        self.surf = torch.zeros(self.nctu * self.nc * self.ng * self.group_bytes // 4, dtype=torch.int32, device=device) if want_surf else None
        self.best = torch.empty(self.nctu * PUS_PER_CTU, dtype=torch.int64, device=device) if want_best else None
        cost = F.mv_cost_table(rng, lam)
        self.cost_host = cost
        self.cost_x = torch.from_numpy(cost.view(np.int16)).to(device)
        self.cost_y = self.cost_x.clone()

    def algorithmic_bytes(self, bpp=1):
        """SURVEY.md section 8(d), batched full-window SAD: per PU (W*H + (W+2R)(H+2R))*bpp read +
        4*(2R+1)^2 written when the surface is produced (8 bytes per PU when only the minimum is)."""
        r = self.range
        total = 0
        for l in range(4):
            n = LEVEL_SIZES[l]
            npu = self.nctu * LEVEL_PUS[l]
            rd = (n * n + (n + 2 * r) * (n + 2 * r)) * bpp
            wr = 4 * self.nc * self.nc if self.surf is not None else 8
            total += npu * (rd + wr)
        return total

    def hbm_floor_bytes(self, bpp=1):
        """What one launch must move through HBM at minimum: both pictures once + the outputs."""
        pix = self.w64 * self.h64 * bpp * 2
        out = self.surf.numel() * 4 if self.surf is not None else 0
        out += self.best.numel() * 8 if self.best is not None else 0
        return pix + out

    def reset(self):
        if self.best is not None:
            hipabi.me_best_reset(self.best)

    def run(self, cur: DevicePicture, ref: DevicePicture):
        self.reset()
        self.search(cur, ref)

    def search(self, cur: DevicePicture, ref: DevicePicture):
        """The one exhaustive-search launch (best[] must have been reset)."""
        hipabi.me_fullsearch(self.depth, self.w64, self.h64, self.range,
                             cur.t, cur.stride, ref.t, ref.stride,
                             surf=self.surf, best=self.best, cost_x=self.cost_x, cost_y=self.cost_y,
                             fenc_off=cur.org, fref_off=ref.org,
                             surf_format=hipabi.SURF_PACKED if self.packed else hipabi.SURF_I32)

    def level_view(self, level):
        """(surface view [nmv, npu], best view [nctu, npu]) of one PU level."""
        b, n = LEVEL_BASE[level], LEVEL_PUS[level]
        sv = None
        if self.surf is not None:      # [ctu*mvy, group, pu, col] -> [ctu*mvy, mvx, pu] with the pad column dropped
            if self.packed:
                import torch
                raw = self.surf.view(torch.uint8).view(self.nctu * self.nc, self.ng, self.group_bytes)
                if level < 2:          # uint16 records at byte 0 (8x8) / 512 (16x16)
                    o = 0 if level == 0 else 512
                    g = raw[:, :, o:o + n * 8].contiguous().view(torch.int16).to(torch.int32) & 0xffff
                else:                  # int32 records at byte 640 (32x32) / 704 (64x64)
                    o = 640 if level == 2 else 704
                    g = raw[:, :, o:o + n * 16].contiguous().view(torch.int32)
                g = g.view(self.nctu * self.nc, self.ng, n, 4)
            else:
                g = self.surf.view(self.nctu * self.nc, self.ng, PUS_PER_CTU, 4)[:, :, b:b + n, :]
            sv = g.permute(0, 1, 3, 2).reshape(self.nctu * self.nc, self.ng * 4, n)[:, :self.nc, :].reshape(-1, n)
        bv = self.best.view(-1, PUS_PER_CTU)[:, b:b + n] if self.best is not None else None
        return sv, bv

    def checksum(self):
        """Order-independent digest of the stage outputs."""
        import torch
        out = {}
        if self.best is not None:
            out["best"] = int(self.best.sum().item())
        if self.surf is not None:
            out["surf"] = int(sum(self.level_view(l)[0].sum(dtype=torch.int64).item() for l in range(4)))
        return out


class SubpelRefine:
    """Stage 2: sub-pel refinement of the ME stage's best integer mv for all 85 PUs of every CTU
    (x265hip_subpel_refine; reference caller motion.cpp:1448-1664).  Output int32 [ctu*85][2] =
    {cost, qmvx | qmvy << 16}."""

    def __init__(self, ms: MotionSearch, subme: int, device, lam=4.0):
        import torch
        self.ms, self.subme = ms, subme
        cq, self.qoff = F.qpel_cost_table(ms.range, lam)
        self.cost_q_host = cq
        self.cost_q = torch.from_numpy(cq.view(np.int16)).to(device)
        self.out = torch.zeros(ms.nctu * PUS_PER_CTU * 2, dtype=torch.int32, device=device)

    def run(self, cur: DevicePicture, ref: DevicePicture):
        ms = self.ms
        hipabi.subpel_refine(ms.depth, ms.w64, ms.h64, ms.range, self.subme, cur.t, cur.stride, ref.t, ref.stride,
                             ms.best, self.cost_q, self.qoff, self.out, fenc_off=cur.org, fref_off=ref.org)

    def checksum(self):
        return {"subpel": int(self.out.to(dtype=__import__("torch").int64).sum().item())}


class FrameParallel:
    """Frame-parallel sharding across GPUs: rank r encodes frame step*world + r.
    All frames of a step search in the same reference picture - the newest picture of the previous step,
    owned by the last rank - so the only data-path exchange is a one-to-many broadcast of that picture,
    issued where the reference raises m_reconRowFlag (encoder/framefilter.cpp:664).  Works on any
    torch.distributed backend (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world = rank, world

    def frame_index(self, step: int) -> int:
        return step * self.world + self.rank

    def reference_owner(self) -> int:
        return self.world - 1

    def exchange(self, ref_plane, newest_plane):
        """ref_plane <- the reference owner's newest picture (in place on every rank).  Either one plane each or equally long
        lists of planes (luma, Cb, Cr)."""
        if isinstance(ref_plane, (list, tuple)):
            for r, n in zip(ref_plane, newest_plane):
                self.exchange(r, n)
            return
        if self.world == 1:
            ref_plane.copy_(newest_plane)
            return
        import torch.distributed as dist
        if self.rank == self.reference_owner():
            ref_plane.copy_(newest_plane)
        dist.broadcast(ref_plane, src=self.reference_owner())
