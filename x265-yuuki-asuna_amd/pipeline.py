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

    def band_view(self, ctu_row0, ctu_rows):
        """The same planes seen as a band of `ctu_rows` CTU rows starting at CTU row `ctu_row0`: sample (0,0) moves to the band's first
        row (luma and chroma), h64 shrinks to the band.  Stages called with a band view treat the band like a picture of its own -
        exactly what the reference does with a slice (--slices: loop filters stop at the slice boundary, cudata.cpp:319, sao.cpp:286-287)
        - while search windows and prediction keep reading the whole reference plane around it."""
        o = DevicePicture.__new__(DevicePicture)
        o.__dict__.update(self.__dict__)
        o.org = self.org + ctu_row0 * 64 * self.stride
        o.org_c = self.org_c + ctu_row0 * 32 * self.stride_c
        o.h64 = ctu_rows * 64
        return o

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
           packed=True (8-bit): 720-byte groups, uint16 for the 8x8 / 16x16 levels (X265HIP_SURF_PACKED);
           packed="b": the same records in blocks of 64, chunk-major inside a block (X265HIP_SURF_PACKED_B: the record-per-lane kernel's native
           layout since round 3, one contiguous block per wavefront step);
           packed="t": the same records stored chunk-major inside a motion-vector row (X265HIP_SURF_PACKED_T, the layout of the
           record-per-lane kernel)
    best : int64 [ctu][85]            cost << 32 | raster mv index
    """

    def __init__(self, w64, h64, rng, depth, device, want_surf=True, want_best=True, lam=4.0, packed=False):
        import torch
        self.w64, self.h64, self.range, self.depth = w64, h64, rng, depth
        self.nctu = (w64 // 64) * (h64 // 64)
        self.nc = 2 * rng + 1
        self.ng = (self.nc + 3) // 4
        self.packed = bool(packed and want_surf)
        self.tiled = self.packed and packed == "t"
        self.blocked = self.packed and packed == "b"
        self.group_bytes = hipabi.SURF_GROUP_BYTES_PACKED if self.packed else hipabi.SURF_GROUP_BYTES_I32
        self.surf_format = hipabi.SURF_PACKED_B if self.blocked else (hipabi.SURF_PACKED_T if self.tiled else (hipabi.SURF_PACKED if self.packed else hipabi.SURF_I32))
        self.nblk = (self.nc * self.ng + 63) // 64                 # X265HIP_SURF_PACKED_B: blocks of 64 records per CTU (the last one allocated whole)
        ctu_bytes = self.nblk * 64 * self.group_bytes if self.blocked else self.nc * self.ng * self.group_bytes
        self.surf = torch.zeros(self.nctu * ctu_bytes // 4, dtype=torch.int32, device=device) if want_surf else None
        self.best = torch.empty(self.nctu * PUS_PER_CTU, dtype=torch.int64, device=device) if want_best else None
        self.best_spare = None          # second buffer of the minima (reset_spare / swap_best), allocated on first use
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

    def part(self, ctu_row0, ctu_rows):
        """The same stage restricted to `ctu_rows` CTU rows from `ctu_row0`: a shallow copy whose outputs are the matching slices of this
        object's tensors (both are CTU-major), to be run with DevicePicture.band_view(ctu_row0, ctu_rows) - every CTU's search is
        independent of the others, so the parts together leave exactly what one whole-picture launch leaves."""
        import copy
        o = copy.copy(self)
        cw = self.w64 // 64
        c0, o.nctu, o.h64 = cw * ctu_row0, cw * ctu_rows, ctu_rows * 64
        if self.best is not None:
            o.best = self.best[c0 * PUS_PER_CTU:(c0 + o.nctu) * PUS_PER_CTU]
        if self.surf is not None:
            per = (self.nblk * 64 if self.blocked else self.nc * self.ng) * self.group_bytes // 4
            o.surf = self.surf[c0 * per:(c0 + o.nctu) * per]
        return o

    def reset(self):
        if self.best is not None:
            hipabi.me_best_reset(self.best)

    def reset_spare(self, stream=None):
        """Double-buffered minima: clears the buffer the NEXT picture's search will merge into (on `stream`, next to this picture's
        stages) - the 8 us fill and the gap in front of it leave the chain between two searches.  swap_best() makes it current."""
        import torch
        if self.best is None:
            return
        if self.best_spare is None:
            self.best_spare = torch.empty_like(self.best)
        hipabi.me_best_reset(self.best_spare, stream=stream)

    def swap_best(self):
        if self.best_spare is not None:
            self.best, self.best_spare = self.best_spare, self.best

    def run(self, cur: DevicePicture, ref: DevicePicture, centres=None):
        self.reset()
        self.search(cur, ref, centres=centres)

    def search(self, cur: DevicePicture, ref: DevicePicture, centres=None):
        """The one exhaustive-search launch (best[] must have been reset).  centres: optional int16 [ctu][2] device tensor - every CTU's window is
        centred on its own displacement (x265hip_me_params.centres)."""
        hipabi.me_fullsearch(self.depth, self.w64, self.h64, self.range,
                             cur.t, cur.stride, ref.t, ref.stride,
                             surf=self.surf, best=self.best, cost_x=self.cost_x, cost_y=self.cost_y,
                             fenc_off=cur.org, fref_off=ref.org,
                             surf_format=self.surf_format, centres=centres)

    def level_view(self, level):
        """(surface view [nmv, npu], best view [nctu, npu]) of one PU level."""
        b, n = LEVEL_BASE[level], LEVEL_PUS[level]
        sv = None
        if self.surf is not None:      # [ctu*mvy, group, pu, col] -> [ctu*mvy, mvx, pu] with the pad column dropped
            if self.packed:
                import torch
                o, nb = ((0, n * 8), (512, n * 8), (640, n * 16), (704, n * 16))[level]      # byte range of the level inside the 720-byte record
                if self.blocked:       # [ctu][block][chunk 45][slot 64][16 B]: the level's chunks -> [ctu][record][level bytes] -> the first nc * ng records
                    lv = self.surf.view(torch.uint8).view(self.nctu, self.nblk, 45, 64, 16)[:, :, o >> 4:(o + nb) >> 4].permute(0, 1, 3, 2, 4) \
                             .reshape(self.nctu, self.nblk * 64, nb)[:, :self.nc * self.ng].reshape(self.nctu * self.nc, self.ng, nb)
                elif self.tiled:       # [row][chunk 45][group][16 B]: the level's chunks -> [row][group][level bytes]
                    lv = self.surf.view(torch.uint8).view(self.nctu * self.nc, 45, self.ng, 16)[:, o >> 4:(o + nb) >> 4].permute(0, 2, 1, 3) \
                             .reshape(self.nctu * self.nc, self.ng, nb)
                else:
                    lv = self.surf.view(torch.uint8).view(self.nctu * self.nc, self.ng, self.group_bytes)[:, :, o:o + nb].contiguous()
                g = (lv.view(torch.int16).to(torch.int32) & 0xffff) if level < 2 else lv.view(torch.int32)      # uint16 / int32 records
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

    def __init__(self, ms: MotionSearch, subme: int, device, lam=4.0, phase_planes=False):
        """phase_planes=True: the reference picture's 15 fractional-phase planes are computed first (x265hip_phase_planes, one launch)
        and every candidate is READ from them instead of being interpolated per candidate tile - same samples, same result."""
        import torch
        self.ms, self.subme = ms, subme
        self.use_planes, self.planes = bool(phase_planes), None
        cq, self.qoff = F.qpel_cost_table(ms.range, lam)
        self.cost_q_host = cq
        self.cost_q = torch.from_numpy(cq.view(np.int16)).to(device)
        self.out = torch.zeros(ms.nctu * PUS_PER_CTU * 2, dtype=torch.int32, device=device)

    def part(self, ms_part, ctu_row0):
        """This stage for the CTU rows of `ms_part` (MotionSearch.part): a shallow copy writing the matching slice of `out`.  The phase
        planes stay the parent's: prepare() them there, run the part with prepared=True."""
        import copy
        o = copy.copy(self)
        o.ms, o.parent = ms_part, self
        c0 = (ms_part.w64 // 64) * ctu_row0
        o.out = self.out[c0 * PUS_PER_CTU * 2:(c0 + ms_part.nctu) * PUS_PER_CTU * 2]
        return o

    def prepare(self, ref: DevicePicture):
        """The reference picture's phase planes (needs only the reference: may run on another stream next to the integer search)."""
        import torch
        if not self.use_planes:
            return
        nb = ref.t.numel() * ref.t.element_size()
        if self.planes is None or self.planes.numel() != 15 * nb:
            self.planes = torch.empty(15 * nb, dtype=torch.uint8, device=ref.t.device)
        hipabi.phase_planes(self.ms.depth, ref.t, 0, self.planes, ref.stride, nb // (ref.stride * (1 if self.ms.depth == 8 else 2)))

    def run(self, cur: DevicePicture, ref: DevicePicture, prepared=False):
        ms = self.ms
        if not prepared:
            self.prepare(ref)
        planes = getattr(self, "parent", self).planes if self.use_planes else None
        hipabi.subpel_refine(ms.depth, ms.w64, ms.h64, ms.range, self.subme, cur.t, cur.stride, ref.t, ref.stride,
                             ms.best, self.cost_q, self.qoff, self.out, fenc_off=cur.org, fref_off=ref.org, phase_planes=planes)

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


class DistTransport:
    """The ring's transfers over torch.distributed point-to-point operations (gloo in the CPU tests, RCCL through torch on request): the
    three plane slices of a band travel as ONE grouped transfer.  The hand-off of producer rank s uses process group s % 2, so that the
    traffic a rank sends never queues behind the traffic it receives (a backend that serialises one communicator's operations, world 2)."""

    def __init__(self, rank, world, stage_through_host=False):
        self.rank, self.world, self.stage = rank, world, stage_through_host
        self.groups = [None, None]

    def setup(self, device=None):
        import datetime
        import torch
        import torch.distributed as dist
        if self.world > 1:
            self.groups = [dist.new_group(list(range(self.world)), timeout=datetime.timedelta(seconds=240)) for _ in range(2)]
            # one collective per group, made by every rank: the communicators exist before the first point-to-point transfer (RCCL creates
            # them lazily, and a first use by only two of the ranks must not turn into a collective the others never join)
            for g in self.groups:
                t = torch.zeros(1, device=device if device is not None else "cpu")
                dist.all_reduce(t, group=g)
        return self.groups

    def _p2p(self, op, planes, ranges, peer, group):
        import torch
        import torch.distributed as dist
        flat = lambda t: t.reshape(-1)
        if not self.stage:
            return dist.batch_isend_irecv([dist.P2POp(op, flat(p)[a:z], peer, group) for p, (a, z) in zip(planes, ranges)])
        views = [flat(p)[a:z] for p, (a, z) in zip(planes, ranges)]
        hosts = [v.cpu() if op is dist.isend else torch.empty(v.shape, dtype=v.dtype) for v in views]
        works = dist.batch_isend_irecv([dist.P2POp(op, h, peer, group) for h in hosts])

        class Staged:
            def __init__(self, w, pairs): self.w, self.pairs = w, pairs
            def wait(self):
                self.w.wait()
                for v, h in self.pairs:
                    v.copy_(h)
        return [Staged(w, ([(v, h)] if op is dist.irecv else [])) for w, v, h in zip(works, views, hosts)]

    def send(self, planes, ranges, band, peers):
        import torch.distributed as dist
        out = []
        for peer in peers:
            out += self._p2p(dist.isend, planes, ranges, peer, self.groups[self.rank % 2])
        return out

    def recv(self, planes, ranges, band, src):
        import torch.distributed as dist
        return self._p2p(dist.irecv, planes, ranges, src, self.groups[src % 2])

    def close(self):
        pass


class DistBcastTransport(DistTransport):
    """Round 6 (round-5 verdict, next 6): the ONE-communicator broadcast `north_star` names, as an A/B switch next to the point-to-point flows - torch.distributed's
    broadcast over one group of all ranks (gloo in the CPU tests).  A finished band of an anchor is one broadcast per plane slice, rooted at the anchor's rank; EVERY
    rank takes part in every broadcast, in one global order (FrameParallelRing: anchor by anchor, band by band) - the ranks that do not read the anchor receive it
    into a scratch picture.  `collective` tells the ring so."""
    collective = True

    def setup(self, device=None):
        import datetime
        import torch
        import torch.distributed as dist
        self.group = None
        if self.world > 1:
            self.group = dist.new_group(list(range(self.world)), timeout=datetime.timedelta(seconds=240))
            dist.all_reduce(torch.zeros(1, device=device if device is not None else "cpu"), group=self.group)
        self.groups = [self.group, self.group]
        return self.groups

    def _bcast(self, planes, ranges, root):
        import torch
        import torch.distributed as dist
        views = [p.reshape(-1)[a:z] for p, (a, z) in zip(planes, ranges)]
        if not self.stage:
            return [dist.broadcast(v, src=root, group=self.group, async_op=True) for v in views]
        hosts = [v.cpu() if root == self.rank else torch.empty(v.shape, dtype=v.dtype) for v in views]
        works = [dist.broadcast(h, src=root, group=self.group, async_op=True) for h in hosts]

        class Staged:
            def __init__(self, w, v, h, copy): self.w, self.v, self.h, self.copy = w, v, h, copy
            def wait(self):
                self.w.wait()
                if self.copy:
                    self.v.copy_(self.h)
        return [Staged(w, v, h, root != self.rank) for w, v, h in zip(works, views, hosts)]

    def send(self, planes, ranges, band, peers):
        return self._bcast(planes, ranges, self.rank)

    def recv(self, planes, ranges, band, src):
        return self._bcast(planes, ranges, src)


class AbiBcastTransport:
    """The broadcast transport through the library's C ABI: ONE RCCL communicator over all ranks (x265hip_comm_init with nranks = world), a band's three plane slices as
    one group of ncclBroadcast calls rooted at the producer (x265hip_recon_publish_rows with peer = -1), all on ONE copy stream per rank - the collectives of one
    communicator are ordered anyway.  `X265HIP_RING_TRANSPORT=bcast` in bench.py; the point-to-point flows of AbiTransport stay the default (xGMI is point to point: a
    broadcast is a ring or a tree of the same links, and it makes every rank wait for the slowest one to join)."""
    collective = True

    def __init__(self, rank, world, device, depth, geom, height):
        self.rank, self.world, self.device, self.depth, self.geom, self.height = rank, world, device, depth, geom, height
        self.comm, self.stream = None, None

    def setup(self, device=None):
        import ctypes
        import torch
        import torch.distributed as dist
        L = hipabi.lib()
        L.x265hip_comm_unique_id.argtypes = [ctypes.c_void_p]
        L.x265hip_comm_init.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_uint8 * 128)()
            hipabi.check(L.x265hip_comm_unique_id(buf), "x265hip_comm_unique_id")
            uid = torch.tensor(list(buf), dtype=torch.uint8)
        uid = uid.to(self.device)
        if self.world > 1:
            dist.broadcast(uid, src=0)
        torch.cuda.synchronize()
        comm = ctypes.c_void_p()
        hipabi.check(L.x265hip_comm_init(ctypes.byref(comm), self.world, uid.cpu().numpy().tobytes(), self.rank), "x265hip_comm_init")
        self.comm = comm
        self.stream = torch.cuda.Stream(device=self.device)
        return None

    def _publish(self, planes, band, root):
        import ctypes
        import torch
        st, my, sc, myc = self.geom
        p = hipabi.ReconPublishParams()
        p.comm, p.rank, p.root, p.peer, p.depth = self.comm, self.rank, root, -1, self.depth
        for i in range(3):
            p.plane[i] = planes[i].data_ptr()
        p.stride, p.stride_c, p.margin_y, p.margin_y_c, p.height = st, sc, my, myc, self.height
        p.ctu_row0, p.ctu_rows = band
        ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)                      # the band's kernels (root) / the readers of the old picture (everyone else) come first
        f = hipabi.lib().x265hip_recon_publish_rows
        f.argtypes = [ctypes.POINTER(hipabi.ReconPublishParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(p), self.stream.cuda_stream), "x265hip_recon_publish_rows")
        done = torch.cuda.Event(); done.record(self.stream)

        class Done:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(done)
        return [Done()]

    def send(self, planes, ranges, band, peers):
        return self._publish(planes, band, self.rank)

    def recv(self, planes, ranges, band, src):
        return self._publish(planes, band, src)

    def close(self):
        if self.comm is not None:
            hipabi.lib().x265hip_comm_destroy(self.comm)
            self.comm = None


class AbiTransport:
    """The ring's transfers through the library's C ABI (csrc/recon_publish.hip): x265hip_comm_* build the communicators,
    x265hip_recon_publish_rows issues a band's three plane slices as one RCCL group on a copy stream - the calls a C++ host makes, so the
    code RCCL executes in `bench.py --gpus N` is the library's own (round-2 verdict, next 7a).
    One 2-rank communicator per DIRECTED flow (producer s -> consumer s + d, d = 1 .. refs): a rank uses a communicator either for sending
    or for receiving, never both, so no transfer queues behind one that waits for the other side (the cycle a single communicator closes
    when every rank sits in a send the peer has queued behind its own send).  The ids travel by torch.distributed (bootstrap only); the
    communicators are joined in one global order of the flows, which keeps the blocking joins acyclic."""

    def __init__(self, rank, world, device, depth, geom, height, refs=1):
        self.rank, self.world, self.device, self.depth, self.geom, self.height, self.refs = rank, world, device, depth, geom, height, refs
        self.send_comm, self.recv_comm = {}, {}          # distance d -> communicator (this rank is rank 0 when it sends, 1 when it receives)
        self.streams = {}

    def setup(self, device=None):
        import ctypes
        import torch
        import torch.distributed as dist
        L = hipabi.lib()
        L.x265hip_comm_unique_id.argtypes = [ctypes.c_void_p]
        L.x265hip_comm_init.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        dists = [d for d in range(1, self.refs + 1) if d % self.world]           # d a multiple of world: the consumer is this rank itself
        mine = torch.zeros((max(1, len(dists)), 128), dtype=torch.uint8)
        for k, d in enumerate(dists):                                            # the producer of a flow makes its id
            buf = (ctypes.c_uint8 * 128)()
            hipabi.check(L.x265hip_comm_unique_id(buf), "x265hip_comm_unique_id")
            mine[k] = torch.tensor(list(buf), dtype=torch.uint8)
        mine = mine.to(self.device)
        gathered = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(gathered, mine)
        torch.cuda.synchronize()
        for k, d, s, role in self.joins(self.rank, self.world, self.refs):         # one global order: (distance, producer)
            raw = gathered[s][k].cpu().numpy().tobytes()
            comm = ctypes.c_void_p()
            hipabi.check(L.x265hip_comm_init(ctypes.byref(comm), 2, raw, role), "x265hip_comm_init")
            (self.recv_comm if role else self.send_comm)[d] = comm
        self.streams = {("s", d): torch.cuda.Stream(device=self.device) for d in dists}
        self.streams.update({("r", d): torch.cuda.Stream(device=self.device) for d in dists})
        return None

    @staticmethod
    def joins(rank, world, refs):
        """The blocking 2-rank communicator joins of `rank`, in the order it makes them: (index of the flow's id in the producer's id list,
        distance d, producer s, role) with role 0 = this rank produces the flow (rank 0 of the communicator), 1 = it consumes it.  Every rank
        walks the SAME global order of flows (distance, then producer) and skips the flows it is not part of, so two ranks always meet in
        the flow that is first for both of them: no cycle of waiting joins can form (tests/test_dist_cpu.py simulates it for worlds 2 .. 8)."""
        out = []
        dists = [d for d in range(1, refs + 1) if d % world]
        for k, d in enumerate(dists):
            for s in range(world):
                c = (s + d) % world
                if rank == s:
                    out.append((k, d, s, 0))
                elif rank == c:
                    out.append((k, d, s, 1))
        return out

    def _publish(self, comm, planes, band, sending, stream):
        import ctypes
        import torch
        st, my, sc, myc = self.geom
        p = hipabi.ReconPublishParams()
        p.comm, p.rank, p.root, p.peer, p.depth = comm, 0 if sending else 1, 0, 1, self.depth
        for i in range(3):
            p.plane[i] = planes[i].data_ptr()
        p.stride, p.stride_c, p.margin_y, p.margin_y_c, p.height = st, sc, my, myc, self.height
        p.ctu_row0, p.ctu_rows = band
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event(); ready.record(cur)
        stream.wait_event(ready)                           # the band's kernels (send) / the readers of the old picture (receive) come first
        f = hipabi.lib().x265hip_recon_publish_rows
        f.argtypes = [ctypes.POINTER(hipabi.ReconPublishParams), ctypes.c_void_p]
        hipabi.check(f(ctypes.byref(p), stream.cuda_stream), "x265hip_recon_publish_rows")
        done = torch.cuda.Event(); done.record(stream)

        class Done:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(done)
        return [Done()]

    def send(self, planes, ranges, band, peers):
        out = []
        for peer in peers:
            d = (peer - self.rank) % self.world
            out += self._publish(self.send_comm[d], planes, band, True, self.streams[("s", d)])
        return out

    def recv(self, planes, ranges, band, src):
        d = (self.rank - src) % self.world
        return self._publish(self.recv_comm[d], planes, band, False, self.streams[("r", d)])

    def close(self):
        L = hipabi.lib()
        for c in list(self.send_comm.values()) + list(self.recv_comm.values()):
            L.x265hip_comm_destroy(c)
        self.send_comm, self.recv_comm = {}, {}


class FrameParallelRing:
    """Frame-parallel encoding with the reference's REAL dependency: frame f is encoded by rank f % world and
    searches / predicts from frames f - 1 .. f - refs, which other ranks are producing at the same time - band by band.  A band of CTU
    rows is handed on as soon as it is final (filtered, side margins extended): the point where the reference raises
    m_reconRowFlag (encoder/framefilter.cpp:664), and the consumer starts a band once the reference rows its search window and
    interpolation taps can touch have arrived - the wait of encoder/frameencoder.cpp:852-868 with m_refLagRows.  Every plane's rows
    (Y, Cb, Cr incl. the side margins, plus the top / bottom margin with the first / last band) travel as one contiguous slice of the
    padded plane, point to point to EVERY rank that needs them: with refs = k reference pictures a finished band goes to the ranks of
    frames f + 1 .. f + k - one-to-many, each over its own link (xGMI is point to point) - the broadcast of SURVEY 8(e).
    The transfers go through a transport: AbiTransport (the library's C ABI, RCCL) in bench.py, DistTransport (torch.distributed:
    gloo in the CPU tests, host-staged for ranks that share a GPU).

    bands: [(first CTU row, CTU rows)]; slices are computed from the plane geometry handed to run_frame."""

    def __init__(self, rank: int, world: int, bands, lag_rows_luma: int, stage_through_host: bool = False, refs: int = 1, transport=None, gop: int = 0):
        """stage_through_host: device slices travel through host copies (a backend without device point-to-point transfers - the gloo
        dry run of bench.py's N > 1 path on a box with fewer GPUs than ranks); never used with RCCL.
        gop (round 6; round-5 verdict, next 6): 0 = the P-only chain above.  G > 0 = mini-GOPs of G pictures: frames that are multiples of G are
        ANCHORS (they read the anchor before them and are the only pictures anybody reads), the G - 1 pictures between two anchors are
        NON-REFERENCED and read the anchor before them - x265's default structure (bframes 4, common/param.cpp:166-168: four non-referenced
        pictures between two anchors, G = 5) with the dependents predicted from ONE side (the step has no bi-predictive search).  The ranks of
        the pictures between two anchors are independent of each other, the chain that bounds the ring is the anchors' (one hop per G pictures).
        An anchor's finished bands go ONCE to every rank that encodes one of the next G pictures; a rank keeps the anchor for all its pictures
        that read it (and its own anchors in the caller's reference planes), so one (producer, consumer) pair never has two copies of a band
        in flight."""
        self.rank, self.world, self.bands = rank, world, list(bands)
        self.gop = max(0, int(gop))
        self.refs = max(1, int(refs)) if not self.gop else 1
        self.lag = lag_rows_luma            # luma rows below a band's last row that its search / interpolation may read
        self.prev, self.next = (rank - 1) % world, (rank + 1) % world
        self.transport = transport if transport is not None else DistTransport(rank, world, stage_through_host)
        self._sends = []
        self.wait_events = None             # time_waits(): (before, after) device events around every band's wait for its reference rows
        # a COLLECTIVE transport (one communicator, broadcasts): every rank joins every anchor's broadcasts, in one global order
        self.collective = bool(getattr(self.transport, "collective", False))
        self._bc_cursor, self._bc_scratch, self._bc_works = 0, None, []
        if self.collective and world > 1 and not self.gop:
            raise ValueError("FrameParallelRing: a broadcast transport needs mini-GOPs (gop > 0): in the P-only chain every band has ONE consumer")

    @staticmethod
    def broadcast_anchors(gop, total_frames):
        """The anchors whose bands travel (every anchor somebody reads), in the global order every rank joins their broadcasts in."""
        return [a for a in range(0, total_frames, gop) if a + 1 <= total_frames - 1]

    def _join_as_bystander(self, a, geom, like):
        """Anchor a's broadcasts, received into a scratch picture: this rank encodes no picture that reads a, but a collective needs everyone."""
        import torch
        if self._bc_scratch is None:
            self._bc_scratch = [torch.empty_like(p) for p in like]
        nb = len(self.bands)
        for b, (r0, rn) in enumerate(self.bands):
            self._bc_works += self.transport.recv(self._bc_scratch, self._rows(geom, r0, rn, b == 0, b == nb - 1), (r0, rn), a % self.world)

    def drain(self, geom, like, total_frames):
        """End of the job (collective transports; a no-op otherwise): the broadcasts of the anchors this rank has not joined yet."""
        if not (self.collective and self.world > 1):
            return
        anchors = self.broadcast_anchors(self.gop, total_frames)
        while self._bc_cursor < len(anchors):
            self._join_as_bystander(anchors[self._bc_cursor], geom, like)
            self._bc_cursor += 1
        self.finish()

    def time_waits(self, on=True):
        """Diagnostics for the first hardware runs (round-3 verdict, next 8): how long does a band's stream sit waiting for the reference
        rows it needs?  Two device events per band on the band's own stream; wait_ms() sums them."""
        self.wait_events = [] if on else None

    def wait_ms(self):
        """(total ms the bands' streams waited for reference rows since time_waits(), number of bands timed)"""
        import torch
        if not self.wait_events:
            return 0.0, 0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in self.wait_events)), len(self.wait_events)

    def frame_index(self, step: int) -> int:
        return step * self.world + self.rank

    @staticmethod
    def _rows(geom, row0, nrows, first, last):
        """(start, stop) element ranges of the band's rows in the three flat planes; geom = (stride, margin_y, stride_c, margin_y_c)."""
        st, my, sc, myc = geom
        y0 = my + row0 * 64 - (my if first else 0)
        y1 = my + (row0 + nrows) * 64 + (my if last else 0)
        c0 = myc + row0 * 32 - (myc if first else 0)
        c1 = myc + (row0 + nrows) * 32 + (myc if last else 0)
        return [(y0 * st, y1 * st), (c0 * sc, c1 * sc), (c0 * sc, c1 * sc)]

    def bands_needed(self, b):
        """Index of the last reference band that must have arrived before band b may start."""
        row_end = (self.bands[b][0] + self.bands[b][1]) * 64 + self.lag
        need = b
        while need + 1 < len(self.bands) and self.bands[need + 1][0] * 64 < row_end:
            need += 1
        return need

    def make_groups(self, device=None):
        """Collective call: every rank makes it once, after init_process_group - the transport builds its communicators."""
        self.groups = self.transport.setup(device)
        return self.groups

    def run_frame(self, step, geom, ref_planes, out_planes, process_band, total_frames=None, first_frame_is_local=True, band_context=None):
        """One frame of this rank.  ref_planes: the flat Y / Cb / Cr tensors the previous frame's bands are received into - with refs > 1
        a LIST of such plane sets, [d - 1] for frame f - d (for frames before the start of the job they already hold the start picture);
        out_planes: where process_band(b, row0, nrows) leaves this frame's finished bands (not reused before the next call returns them:
        the sends of a frame are only waited for at the start of this rank's next frame, or by finish()).  total_frames: frames of the whole
        job - a frame past the end has no consumer and nothing is sent to it.
        band_context(b): context manager under which band b is waited for, processed and sent (stages.BandedFramePipeline.band_context:
        bands alternate between HIP streams, and a transfer orders itself against the stream that is current when it is issued / waited
        for - so a band's arrival and departure only hold up that band's stream); the receives are posted outside it."""
        import contextlib
        f = self.frame_index(step)
        nb = len(self.bands)
        T = self.transport
        keep_own_anchor = False
        if self.gop:
            G = self.gop
            anchor = ((f - 1) // G) * G if f > 0 else -1   # the picture frame f reads (-1: the start picture, local on every rank)
            d0 = f - anchor
            ref_sets = {d0: ref_planes}
            # the anchor travels to this rank once: with its FIRST picture after the anchor (a later one, f - world > anchor, finds it in
            # ref_planes); an anchor this rank made itself was copied into ref_planes when it was finished
            srcs = [d0] if (self.world > 1 and anchor >= 0 and d0 % self.world and f - self.world <= anchor) else []
            peers = []
            if f % G == 0 and self.world > 1:
                last = f + G if total_frames is None else min(f + G, total_frames - 1)
                ranks = {(self.rank + k) % self.world for k in range(1, last - f + 1)}
                peers = sorted(ranks - {self.rank}, key=lambda r: (r - self.rank) % self.world)
                keep_own_anchor = any(k % self.world == 0 for k in range(1, last - f + 1))
            if self.world == 1:
                keep_own_anchor = f % G == 0
            if self.collective and self.world > 1:
                # the broadcasts of every anchor older than the one this picture reads, which this rank has not joined yet: as a bystander.  (The anchor it reads
                # is either joined below as a consumer, or was joined with an earlier picture of this rank, or is this rank's own.)
                if total_frames is None:
                    raise ValueError("FrameParallelRing: a broadcast transport needs total_frames (which anchors travel is a global fact)")
                anchors = self.broadcast_anchors(G, total_frames)
                while self._bc_cursor < len(anchors) and anchors[self._bc_cursor] < f and not (srcs and anchors[self._bc_cursor] == anchor):
                    assert anchors[self._bc_cursor] < anchor or not srcs, (f, anchor, anchors[self._bc_cursor])
                    self._join_as_bystander(anchors[self._bc_cursor], geom, ref_planes)
                    self._bc_cursor += 1
                if srcs:
                    assert anchors[self._bc_cursor] == anchor, (f, anchor, self._bc_cursor)
                    self._bc_cursor += 1
        else:
            ref_sets = dict(enumerate(ref_planes if self.refs > 1 else [ref_planes], 1))
            assert len(ref_sets) == self.refs
            # which references travel: frame f - d exists (first_frame_is_local: the job's first frames read the start picture) and was made by
            # another rank (d a multiple of world: this rank made it itself - the caller hands its own planes in)
            srcs = [d for d in range(1, self.refs + 1)
                    if self.world > 1 and d % self.world and not (first_frame_is_local and f - d < 0)]
            peers = [(self.rank + d) % self.world for d in range(1, self.refs + 1)
                     if self.world > 1 and d % self.world and (total_frames is None or f + d < total_frames)]
        self.finish()                                       # the previous frame's sends must be through before out_planes is rewritten
        posted = {d: -1 for d in srcs}
        arrived = {d: -1 for d in srcs}
        pending = {d: {} for d in srcs}

        def post(d, upto):
            while posted[d] < upto:                         # receives are posted in band order per source
                posted[d] += 1
                r0, rn = self.bands[posted[d]]
                pending[d][posted[d]] = T.recv(ref_sets[d], self._rows(geom, r0, rn, posted[d] == 0, posted[d] == nb - 1), (r0, rn), (self.rank - d) % self.world)
        for d in srcs:
            if d > 1 or self.collective:
                post(d, nb - 1)                             # older references were finished long ago: everything at once (a collective transport: always -
                                                            # this rank's own broadcasts must not be queued between two bands of an older anchor)
        for b, (row0, n) in enumerate(self.bands):
            need = self.bands_needed(b)
            if 1 in posted:
                post(1, need)                               # the newest reference: just ahead of the band that needs it
            with (band_context(b) if band_context is not None else contextlib.nullcontext()):
                timed = self.wait_events is not None and srcs
                if timed:
                    import torch
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                for d in srcs:
                    while arrived[d] < need:
                        arrived[d] += 1
                        for w in pending[d].pop(arrived[d]):
                            w.wait()
                if timed:
                    e1.record()
                    self.wait_events.append((e0, e1))
                process_band(b, row0, n)
                if peers:
                    self._sends += T.send(out_planes, self._rows(geom, row0, n, b == 0, b == nb - 1), (row0, n), peers)
        for d in srcs:                                      # bands below the last search window still belong to the reference picture
            post(d, nb - 1)
            for bb in sorted(pending[d]):
                for w in pending[d][bb]:
                    w.wait()
        if self.gop and self.collective and self.world > 1 and peers:
            anchors = self.broadcast_anchors(self.gop, total_frames)
            assert anchors[self._bc_cursor] == f, (f, self._bc_cursor)      # this rank's own anchor: its broadcasts were queued band by band above
            self._bc_cursor += 1
        if keep_own_anchor:                                 # this rank reads its own anchor later: the reference planes are where it looks for it
            for r, o in zip(ref_planes, out_planes):
                r.copy_(o)

    def finish(self):
        for w in self._sends + self._bc_works:
            w.wait()
        self._sends, self._bc_works = [], []
