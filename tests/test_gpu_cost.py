"""GPU parity of the SUB-SAMPLE COST TABLES (round 6; include/x265hip.h): x265hip_cost_candidates, x265hip_cost_tables and the
row-granular service x265hip_cost_stream against the oracle (oracle/x265_oracle_pipeline8.c: every value produced the way
MotionEstimate::subpelCompare, motion.cpp:1571-1664, produces it - the PU's own interpolation and satd entries of the pinned
primitive table).  Bit-exact: every byte of every record."""
import ctypes
import importlib
import time

import numpy as np
import pytest

import cost_oracle as C

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
F = importlib.import_module("x265-yuuki-asuna_amd.frames")


def _hier_surfaces(rng, nctu, window, hi):
    """Random SAD rasters of the 64 8x8 blocks with the 16x16 / 32x32 / 64x64 levels summed from them, as I32 records."""
    nc, ng = 2 * window + 1, (2 * window + 4) // 4
    lv = [rng.integers(0, hi, (nctu, nc, ng * 4, 64)).astype(np.int64)]
    for n in (16, 4, 1):
        lv.append(lv[-1].reshape(nctu, nc, ng * 4, n, 4).sum(axis=4))
    return np.ascontiguousarray(np.concatenate(lv, axis=3).reshape(nctu, nc, ng, 4, 85).transpose(0, 1, 2, 4, 3).astype(np.int32))


@pytest.mark.parametrize("window,shapes,k,hi", [(3, 0, 1, 4), (8, 1, 2, 3), (8, 2, 2, 4000), (12, 2, 1, 60000), (0, 2, 2, 9)])
def test_candidates_match_oracle(window, shapes, k, hi):
    import torch
    dev = torch.device("cuda:0")
    O = C.oracle()
    rng = np.random.default_rng(window * 10 + shapes)
    nctu = 7
    surf = _hier_surfaces(rng, nctu, window, hi)                              # hi small: ties everywhere - the scan-order rule decides
    centres = rng.integers(-40, 41, (nctu, 2)).astype(np.int16)
    npu = len(A.cost_pu_list(shapes))
    mvc = (np.abs(np.arange(-window, window + 1)) * max(1, hi // 8)).astype(np.uint16)      # a vector cost that really reorders the minima
    for cen, cost in ((centres, None), (None, None), (centres, mvc)):
        want = O.cost_candidates(surf, cen, nctu, window, shapes, k, mv_cost=cost)
        d_cand = torch.full((nctu, npu, k, 2), 77, dtype=torch.int16, device=dev)
        A.cost_candidates(torch.from_numpy(surf).to(dev), None if cen is None else torch.from_numpy(cen).to(dev), nctu, window, shapes, k, d_cand,
                          mv_cost=None if cost is None else torch.from_numpy(cost.view(np.int16)).to(dev))
        torch.cuda.synchronize()
        got = d_cand.cpu().numpy()
        assert np.array_equal(got, want), f"window {window} shapes {shapes} k {k}: {np.count_nonzero(got != want)} candidate components differ"
        if cost is not None and window:
            assert not np.array_equal(want, O.cost_candidates(surf, cen, nctu, window, shapes, k)), "the vector cost changed no candidate"
    if window == 0:
        assert (want[:, :, 1, 0] == -32768).all()                             # one displacement only: there is no second candidate


def _device_picture(pic, dev):
    import torch
    return [torch.from_numpy(pic[n].view(np.uint8)).to(dev) for n in ("y", "cb", "cr")]


def _device_phases(depth, d_planes, g, dev):
    import torch
    out = []
    for i, (st, rw) in enumerate(((g["stride"], g["rows"]), (g["stride_c"], g["rows_c"]), (g["stride_c"], g["rows_c"]))):
        nph = 63 if i else 15
        d = torch.zeros(nph * d_planes[i].numel(), dtype=torch.uint8, device=dev)
        # the kernel reads a few bytes around the plane's first / last rows: give the source guard space
        es = 1 if depth == 8 else 2
        lo, hi = 4 * st * es + 64, 8 * st * es
        src = torch.zeros(lo + d_planes[i].numel() + hi, dtype=torch.uint8, device=dev)
        src[lo:lo + d_planes[i].numel()] = d_planes[i]
        A.phase_planes(depth, src, lo, d, st, rw, chroma=bool(i))
        out.append(d)
    return out


@pytest.mark.parametrize("few", [0, 5], ids=["scattered", "few_vectors"])
@pytest.mark.parametrize("depth,chroma,subme,shapes,k,sad", [(8, 1, 3, 1, 1, 0), (8, 1, 4, 2, 2, 1), (10, 1, 4, 2, 2, 1), (12, 1, 3, 2, 1, 0), (8, 0, 2, 2, 2, 1), (10, 0, 7, 0, 1, 1), (8, 1, 5, 1, 2, 0),
                                                             (8, 1, 7, 1, 1, 1)])
def test_tables_match_oracle(depth, chroma, subme, shapes, k, sad, few):
    """few_vectors: the PUs of a CTU sit on a handful of distinct vectors - the shared-tile kernel (a map of 8x8-block costs per vector, summed per PU) writes the
    records; scattered: (nearly) every PU-candidate has a vector of its own - those CTUs are flagged and left to the per-PU kernel.  Both against the same oracle."""
    import torch
    dev = torch.device("cuda:0")
    O = C.oracle()
    clip = F.synth_clip(192, 128, 2, depth=depth, seed=700 + depth + subme)
    pmax = (1 << depth) - 1
    y1 = clip[1][0].copy(); y1[::9, ::7] = pmax; y1[4::13, 3::5] = 0        # extremes: the clipping paths of the filters and large tile sums
    fenc, ref = C.picture((y1, clip[1][1], clip[1][2])), C.picture(clip[0])
    g = fenc
    nctu, npu = (g["width"] // 64) * (g["height"] // 64), len(A.cost_pu_list(shapes))
    rng = np.random.default_rng(subme)
    cand = rng.integers(-9, 10, (nctu, npu, k, 2)).astype(np.int16)
    if few:
        pool = np.array([[3, 2], [2, 2], [3, 1], [-7, 5], [0, 0]], np.int16)
        cand = pool[rng.integers(0, few, (nctu, npu, k))]
        cand[0] = rng.integers(-9, 10, (npu, k, 2)).astype(np.int16)           # ... and one CTU that is scattered all the same: both kernels in one launch pair
    cand[:, :, 0] = np.array([3, 2], np.int16)                                 # the clip's own motion: the small costs a real search ends on
    cand[1, 5, k - 1, 0] = -32768
    want = O.cost_tables(depth, [fenc["y"], fenc["cb"], fenc["cr"]], [ref["y"], ref["cb"], ref["cr"]], g["stride"], g["stride_c"], g["margin_x"], g["margin_y"],
                         g["margin_y_c"], g["width"], 0, g["height"] // 64, shapes, k, subme, chroma, cand, sad_costs=sad)
    d_f, d_r = _device_picture(fenc, dev), _device_picture(ref, dev)
    d_ph = _device_phases(depth, d_r, g, dev)
    rec = A.cost_record_bytes(subme, sad)
    d_t = torch.full((nctu, npu, k, rec), 0xAB, dtype=torch.uint8, device=dev)
    es = 1 if depth == 8 else 2
    for r0, n in ((0, 1), (1, g["height"] // 64 - 1)):                         # two bands
        c0 = r0 * (g["width"] // 64)
        A.cost_tables(depth, g["width"], g["stride"], g["margin_x"], g["margin_y"], g["stride_c"], g["margin_y_c"], r0, n, d_f, d_r, d_ph,
                      g["stride"] * g["rows"] * es, g["stride_c"] * g["rows_c"] * es, shapes, k, subme, chroma,
                      torch.from_numpy(cand[c0:c0 + n * (g["width"] // 64)].copy()).to(dev), d_t[c0:], sad_costs=sad)
    torch.cuda.synchronize()
    got = d_t.cpu().numpy()
    used = C.used_mask(subme, sad)                                             # the padding bytes of a record are not written
    bad = np.argwhere((got[..., used] != want[..., used]).any(axis=-1))
    assert len(bad) == 0, f"{len(bad)} records differ, first (ctu, pu, candidate) {bad[0].tolist()}: got {got[tuple(bad[0])][:24].tolist()} want {want[tuple(bad[0])][:24].tolist()}"
    mv, cost = C.parse_records(got, subme)
    assert (cost[mv[..., 0] != -32768] != 0xffffffff).mean() > 0.5            # most deltas are representable (random far-off vectors may saturate)


class _Stream:
    """ctypes handle of x265hip_cost_stream for the tests."""

    def __init__(self, depth, g, centre_range, window, k, shapes, subme, chroma, slots=4, pictures=6, views=3, band_rows=2, sad_costs=0):
        L = A.lib()
        self.L, self.g, self.subme = L, g, subme
        p = A.CostStreamParams(depth, g["width"], g["height"], g["stride"], g["margin_x"], g["margin_y"], g["stride_c"], g["margin_y_c"], centre_range, window,
                               k, shapes, subme, int(chroma), int(sad_costs), slots, pictures, views, band_rows, 0)
        self.h = ctypes.c_void_p()
        L.x265hip_cost_stream_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(A.CostStreamParams)]
        A.check(L.x265hip_cost_stream_create(ctypes.byref(self.h), ctypes.byref(p)), "x265hip_cost_stream_create")
        L.x265hip_cost_stream_destroy.argtypes = [ctypes.c_void_p]
        L.x265hip_cost_stream_picture_rows.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.x265hip_cost_stream_pair_open.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
        L.x265hip_cost_stream_tables.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.x265hip_cost_stream_tables.restype = ctypes.c_void_p
        L.x265hip_cost_stream_ready.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.x265hip_cost_stream_ready.restype = ctypes.POINTER(ctypes.c_int)
        L.x265hip_cost_stream_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(A.CostStreamStats)]
        self.ctu_rows = g["height"] // 64
        self.sad_costs = int(sad_costs)
        self.row_bytes = A.cost_ctu_bytes(subme, shapes, k, sad_costs) * (g["width"] // 64)
        self.shape = ((g["width"] // 64) * self.ctu_rows, len(A.cost_pu_list(shapes)), k, A.cost_record_bytes(subme, sad_costs))

    def rows(self, key, pic, r0, n):
        return self.L.x265hip_cost_stream_picture_rows(self.h, key, pic["y"].ctypes.data, pic["cb"].ctypes.data, pic["cr"].ctypes.data, r0, n)

    def open(self, slot, fkey, rkey, w=None, mask=0, mv_cost=None):
        buf = None
        if w is not None:
            buf = (ctypes.c_int * 12)(*[v for c in range(3) for v in w[c]])
        mc = None if mv_cost is None else np.ascontiguousarray(mv_cost, np.uint16)
        return self.L.x265hip_cost_stream_pair_open(self.h, slot, fkey, rkey, buf, mask, None if mc is None else mc.ctypes.data)

    def wait(self, slot, gen, rows=None, timeout=60):
        rdy = self.L.x265hip_cost_stream_ready(self.h, slot)
        rows = range(self.ctu_rows) if rows is None else rows
        t0 = time.time()
        while any(rdy[r] != gen for r in rows) and time.time() - t0 < timeout:
            time.sleep(0.002)
        return [rdy[r] for r in range(self.ctu_rows)]

    def tables(self, slot):
        raw = (ctypes.c_uint8 * (self.row_bytes * self.ctu_rows)).from_address(self.L.x265hip_cost_stream_tables(self.h, slot))
        return np.frombuffer(raw, dtype=np.uint8).reshape(self.shape).copy()

    def stats(self):
        st = A.CostStreamStats()
        self.L.x265hip_cost_stream_stats(self.h, ctypes.byref(st))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def close(self):
        self.L.x265hip_cost_stream_destroy(self.h)


def _equal_records(got, want, subme, sad_costs=0):
    used = C.used_mask(subme, sad_costs)
    return np.argwhere((got[..., used] != want[..., used]).any(axis=-1))


@pytest.mark.parametrize("depth", [8, 10])
def test_cost_stream_follows_the_rows_and_serves_weighted_views(depth):
    """The service under the frame-thread protocol: the source picture in one piece, the reference CTU row by CTU row - a pair opened BEFORE any row
    exists, one opened half way, one on a WEIGHTED view (luma and Cb) of the same picture; rows of a pair become ready only once the reference rows
    its candidates can reach have arrived, and every record equals the oracle's chain on the whole pictures."""
    clip = F.synth_clip(256, 256, 3, depth=depth, seed=810 + depth)
    fenc, ref, other = C.picture(clip[2]), C.picture(clip[0]), C.picture(clip[1])
    g = fenc
    cr, window, k, shapes, subme, chroma = 20, 4, 2, 2, 3, 1
    corr = 14 - depth
    w3 = [(48, 1 << (5 + corr), 6 + corr, 3), (70, 1 << (5 + corr), 6 + corr, -2), (64, 1 << (5 + corr), 6 + corr, 0)]
    wref = dict(ref)
    wref["y"] = C.weight_plane(ref["y"], depth, w3[0]); wref["cb"] = C.weight_plane(ref["cb"], depth, w3[1])
    mvc = (np.abs(np.arange(-window, window + 1)) * 37 + 5).astype(np.uint16)
    sad = int(depth == 10)                                                     # the 10-bit run carries the SAD-typed costs too
    want_plain = C.chain(depth, fenc, ref, cr, window, shapes, k, subme, chroma, sad_costs=sad)
    want_w = C.chain(depth, fenc, wref, cr, window, shapes, k, subme, chroma, sad_costs=sad)
    want_other = C.chain(depth, other, ref, cr, window, shapes, k, subme, chroma, mv_cost=mvc, sad_costs=sad)
    S = _Stream(depth, g, cr, window, k, shapes, subme, chroma, sad_costs=sad)
    try:
        gen0 = S.open(0, 1001, 2000)                                           # before anything has arrived
        assert gen0 > 0
        assert S.rows(1001, fenc, 0, S.ctu_rows) == 0
        assert S.rows(2000, ref, 0, 1) == 0
        time.sleep(0.3)
        assert S.wait(0, gen0, rows=[], timeout=0) == [0] * S.ctu_rows, "a row was served before the reference row below it existed"
        gen1 = S.open(1, 1001, 2000, w3, 3)                                    # weighted view of the same picture, half way
        assert S.rows(2000, ref, 1, 1) == 0                                    # rows 0 .. 1 there: row 0 of both pairs can be served (needs <= r + 1)
        assert S.wait(0, gen0, rows=[0])[0] == gen0 and S.wait(1, gen1, rows=[0])[0] == gen1
        assert S.wait(0, gen0, rows=[], timeout=0)[1:] == [0] * (S.ctu_rows - 1)
        assert S.rows(1002, other, 0, S.ctu_rows) == 0
        gen2 = S.open(2, 1002, 2000, mv_cost=mvc)                              # a second source picture on the same (shared) view, candidates ranked with a vector cost
        assert S.rows(2000, ref, 3, 1) == 0 and S.rows(2000, ref, 2, 1) == 0   # out of order
        for slot, gen in ((0, gen0), (1, gen1), (2, gen2)):
            assert S.wait(slot, gen) == [gen] * S.ctu_rows, f"slot {slot} never completed: {S.stats()}"
        for slot, want in ((0, want_plain), (1, want_w), (2, want_other)):
            bad = _equal_records(S.tables(slot), want[2], subme, sad)
            assert len(bad) == 0, f"slot {slot}: {len(bad)} records differ, first {bad[0].tolist()}"
        assert len(_equal_records(want_other[2], C.chain(depth, other, ref, cr, window, shapes, k, subme, chroma, sad_costs=sad)[2], subme, sad)) > 0, "the vector cost changed no record"
        st = S.stats()
        assert st["failed"] == 0 and st["pairs_completed"] == 3 and st["views_opened"] == 2 and st["views_shared"] == 1 and st["lines_weighted"] > 0, st
        # reopening a slot clears its flags before anything is rewritten; the new pair is served again
        gen0b = S.open(0, 1002, 2000, w3, 3)
        assert gen0b == gen0 + 1
        assert S.wait(0, gen0b) == [gen0b] * S.ctu_rows
        assert len(_equal_records(S.tables(0), C.chain(depth, other, wref, cr, window, shapes, k, subme, chroma, sad_costs=sad)[2], subme, sad)) == 0
        assert S.open(9, 1, 2) < 0 and S.rows(5, fenc, 3, 4) < 0                # bad slot / rows past the picture: refused
    finally:
        S.close()


def test_cost_stream_at_4k():
    """BASELINE configs[2] size (3840x2160, preset slow: --subme 3, rectangles, chroma SATD): one pair through the service with the rows of the reference
    arriving one at a time; a sample of CTU rows against the oracle's chain on the same pictures and size-independent properties of ALL records."""
    O = C.oracle()
    clip = F.synth_clip(3840, 2160, 2, depth=8, seed=265)
    fenc, ref = C.picture(clip[1]), C.picture(clip[0])
    g = fenc
    cr, window, k, shapes, subme, chroma = 57, 8, 1, 1, 3, 1
    S = _Stream(8, g, cr, window, k, shapes, subme, chroma, slots=2, pictures=4, views=2, band_rows=8)
    try:
        t0 = time.time()
        gen = S.open(0, 11, 22)
        assert S.rows(11, fenc, 0, S.ctu_rows) == 0
        for r in range(S.ctu_rows):
            assert S.rows(22, ref, r, 1) == 0
        assert S.wait(0, gen, timeout=120) == [gen] * S.ctu_rows, S.stats()
        dt = time.time() - t0
        got = S.tables(0)
        st = S.stats()
        print(f"\n[cost stream 4K] one pair, {S.ctu_rows} rows: {dt * 1e3:.1f} ms wall, worker busy {st['us_busy'] / 1e3:.1f} ms, {st['bytes_downloaded'] / 1e6:.1f} MB of records, "
              f"{st['bands']} bands")
        assert st["failed"] == 0
        mv, cost = C.parse_records(got, subme)
        assert (mv[..., 0] != -32768).all() and (np.abs(mv[..., 0]) <= 57 + 8).all() and (mv[..., 1] >= -(52 + 8)).all() and (mv[..., 1] <= 34 + 8).all()
        assert (cost != 0xffffffff).mean() > 0.999
        # the clip moves by (3, 2) samples per picture: nearly every PU's candidate is that displacement
        assert ((mv[..., 0] == 3) & (mv[..., 1] == 2)).mean() > 0.9
        # additivity: a 64x64 PU's cost at every position = the sum of its four 32x32 PUs' costs when all five records sit on the same vector
        m1, c1 = mv[:, :, 0], cost[:, :, 0].astype(np.int64)                   # candidate 0: [ctu][pu][2], [ctu][pu][position]
        same = (m1[:, 80:85] == m1[:, 84:85]).all(axis=(1, 2)) & (cost[:, 80:85, 0] != 0xffffffff).all(axis=(1, 2))
        assert same.mean() > 0.8
        assert np.array_equal(c1[same][:, 84], c1[same][:, 80:84].sum(axis=1))
        # sample: the first, a middle and the last CTU row against the oracle's chain on a three-row crop would need the same window context; use the
        # oracle's TABLES on the service's own candidates instead (the candidates are covered by the small-picture test)
        ctus_w = g["width"] // 64
        for r in (0, 17, S.ctu_rows - 1):
            cand = np.ascontiguousarray(mv[r * ctus_w:(r + 1) * ctus_w])
            want = O.cost_tables(8, [fenc["y"], fenc["cb"], fenc["cr"]], [ref["y"], ref["cb"], ref["cr"]], g["stride"], g["stride_c"], g["margin_x"], g["margin_y"],
                                 g["margin_y_c"], g["width"], r, 1, shapes, k, subme, chroma, cand)
            bad = _equal_records(got[r * ctus_w:(r + 1) * ctus_w], want, subme)
            assert len(bad) == 0, f"CTU row {r}: {len(bad)} records differ"
    finally:
        S.close()
