"""GPU parity: CTU-tiled exhaustive motion search (x265hip_me_fullsearch) vs the oracle's
restatement of the reference full search (motion.cpp:1397-1445 over pu[].sad / sad_x4)."""
import ctypes
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

pkg = importlib.import_module("x265-yuuki-asuna_amd")
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _valid(surf, ms):
    """[ctu*mvy, mvx, 85] view of a surface buffer with the pad column of the last group dropped."""
    v = surf.reshape(ms.nctu * ms.nc, ms.ng, 85, 4).transpose(0, 1, 3, 2).reshape(ms.nctu * ms.nc, ms.ng * 4, 85)
    return v[:, :ms.nc, :]


def _run(width, height, rng, depth, seed, extreme=None, packed=False):
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(width, height, 2, depth=depth, seed=seed)
    y0, y1 = clip[0][0], clip[1][0]
    if extreme == "flat":
        y0 = np.zeros_like(y0); y1 = np.full_like(y1, (1 << depth) - 1)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(y0, dev)
    ms = P.MotionSearch(cur.w64, cur.h64, rng, depth, dev, packed=packed)
    ms.run(cur, ref)
    torch.cuda.synchronize()
    O = _oracle()
    nctu = ms.nctu
    surf, best = O.me_fullsearch(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org,
                                 cur.w64, cur.h64, rng, 0, nctu, ms.cost_host, ms.cost_host)
    e = _valid(surf, ms)
    if packed:      # X265HIP_SURF_PACKED: same values, u16 records for the 8x8 / 16x16 levels
        assert ms.surf.numel() * 4 == (nctu * ((ms.nc * ms.ng + 63) // 64) * 46080 if packed == "b" else nctu * ms.nc * ms.ng * 720)      # x265hip_surf_ctu_bytes
        for level in range(4):
            b, n = P.LEVEL_BASE[level], P.LEVEL_PUS[level]
            g = ms.level_view(level)[0].cpu().numpy()
            assert np.array_equal(g, e[:, :, b:b + n].reshape(-1, n)), f"packed surface level {level} differs"
    else:
        g = _valid(ms.surf.cpu().numpy(), ms)
        assert np.array_equal(g, e), f"SAD surface differs ({np.count_nonzero(g != e)} of {g.size})"
    gb = ms.best.cpu().numpy().view(np.uint64)
    assert np.array_equal(gb, best), f"best differs ({np.count_nonzero(gb != best)} of {gb.size})"


@pytest.mark.parametrize("depth", [8, 10])
def test_me_small_range(depth):
    _run(128, 128, 8, depth, seed=1)


@pytest.mark.parametrize("rng", [12, 13, 14])
def test_me_10bit_ranges_around_an_lds_pitch_step(rng):
    """+-13 at 16-bit samples: the generic kernel's window row + skew is exactly 64 dwords, the column-group kernel's one more - the
    launch used to fail with an "internal" error there (found by tools/r3_soak.py); its neighbours on either side of the pitch step."""
    _run(128, 64, rng, 10, seed=30 + rng)


def test_me_non_ctu_multiple_picture():
    _run(200, 136, 12, 8, seed=2)       # padded to 256x192 like the reference's whole-CTU allocation


def test_me_extremes():
    _run(128, 64, 5, 8, seed=3, extreme="flat")   # all-min vs all-max (TestBench cases [1]/[2])
    _run(64, 64, 5, 10, seed=3, extreme="flat")


def test_me_packed_surface_format():
    _run(128, 128, 8, 8, seed=1, packed=True)
    _run(200, 136, 12, 8, seed=2, packed=True)
    _run(128, 64, 5, 8, seed=3, extreme="flat", packed=True)     # 16x16 SAD = 65280: the u16 maximum
    _run(256, 64, 57, 8, seed=4, packed=True)


@pytest.mark.parametrize("case", [(128, 128, 8, 1, None), (200, 136, 12, 2, None), (128, 64, 5, 3, "flat"), (256, 64, 57, 4, None), (128, 128, 24, 5, None),
                                  (64, 64, 1, 6, None), (192, 64, 32, 7, None), (128, 128, 3, 8, "flat")])
def test_me_chunk_major_packed_format_record_per_lane_kernel(case):
    """X265HIP_SURF_PACKED_T: written by the record-per-lane kernel (csrc/me_cand_kernel.hip) - surfaces and minima against the oracle,
    ranges from one candidate group (a single live lane group) to the default merange and beyond one LDS bank period."""
    w, h, rng, seed, extreme = case
    _run(w, h, rng, 8, seed=seed, extreme=extreme, packed="t")


@pytest.mark.parametrize("case", [(128, 128, 8, 11, None), (200, 136, 12, 12, None), (128, 64, 5, 13, "flat"), (256, 64, 57, 14, None), (128, 128, 24, 15, None),
                                  (64, 64, 1, 16, None), (192, 64, 32, 17, None), (256, 128, 15, 18, None)])
def test_me_block_major_packed_format_record_per_lane_kernel(case):
    """X265HIP_SURF_PACKED_B (round 3): the records of a CTU in blocks of 64, chunk-major inside a block - what the default bench line times.
    Ranges whose record count is a multiple of 64 (+-15: 31 x 8 = 248 is not, +-8: 17 x 5 = 85 ...), one record group, the default merange."""
    w, h, rng, seed, extreme = case
    _run(w, h, rng, 8, seed=seed, extreme=extreme, packed="b")


def test_me_record_per_lane_kernel_serves_the_other_formats(monkeypatch):
    """X265HIP_ME_KERNEL=cand routes every 8-bit launch to the record-per-lane kernel: int32 and record-contiguous packed surfaces,
    minima alone, surfaces alone."""
    import torch
    monkeypatch.setenv("X265HIP_ME_KERNEL", "cand")
    A.lib().x265hip_me_env_refresh()          # the switches are read once per process: this test flips one
    _run(128, 128, 8, 8, seed=21)
    _run(200, 136, 12, 8, seed=22, packed=True)
    _run(256, 64, 57, 8, seed=23)
    dev = torch.device("cuda:0")
    clip = F.synth_clip(192, 128, 2, depth=8, seed=24)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    O = _oracle()
    for kw in (dict(want_surf=False), dict(want_best=False, packed="t"), dict(want_best=False, packed="b"), dict(want_best=False)):
        ms = P.MotionSearch(cur.w64, cur.h64, 14, 8, dev, **kw)
        ms.run(cur, ref)
        torch.cuda.synchronize()
        surf, best = O.me_fullsearch(8, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, 14, 0, ms.nctu, ms.cost_host, ms.cost_host)
        if ms.best is not None:
            assert np.array_equal(ms.best.cpu().numpy().view(np.uint64), best)
        if ms.surf is not None:
            e = _valid(surf, ms)
            for level in range(4):
                b, n = P.LEVEL_BASE[level], P.LEVEL_PUS[level]
                assert np.array_equal(ms.level_view(level)[0].cpu().numpy(), e[:, :, b:b + n].reshape(-1, n)), f"{kw} level {level}"


def test_me_chunk_major_format_refused_when_the_window_does_not_fit():
    """A window the record-per-lane kernel cannot hold in LDS: X265HIP_SURF_PACKED_T is refused with an error, nothing is written."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(64, 64, 2, depth=8, seed=8)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    A.lib().x265hip_me_fullsearch.restype = ctypes.c_int
    p = A.MEParams()
    p.depth, p.width, p.height, p.range = 8, 64, 64, 130
    p.fenc = p.fref = cur.t.data_ptr() + cur.org
    p.fenc_stride = p.fref_stride = cur.stride
    surf = torch.zeros(1024, dtype=torch.int32, device=dev)
    p.surf, p.surf_format = surf.data_ptr(), A.SURF_PACKED_T
    assert A.lib().x265hip_me_fullsearch(ctypes.byref(p), None) < 0
    err = A.lib().x265hip_last_error()
    assert (b"PACKED_T" in err or b"LDS" in err) and int(surf.abs().sum()) == 0, err


def test_me_packed_surface_rejected_for_high_bit_depth():
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(64, 64, 2, depth=10, seed=7)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, 4, 10, dev, packed=True)
    with pytest.raises(A.X265HipError):
        ms.run(cur, ref)


def test_me_10bit_column_group_kernel():
    _run(128, 128, 16, 10, seed=11)     # 512-byte LDS pitch
    _run(256, 64, 57, 10, seed=12)      # the reference's default merange
    _run(64, 64, 20, 10, seed=13, extreme="flat")


def test_me_default_merange_one_ctu_row():
    _run(256, 64, 57, 8, seed=4)        # the reference's default merange (param.cpp:198)


def test_me_hierarchy_property_full_size():
    """Size-independent property at a BASELINE size (1080p): every parent SAD equals the sum of its four
    children at the same mv, and the best key is <= every candidate's key."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(1920, 1080, 2, depth=8, seed=5)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, 16, 8, dev)
    ms.run(cur, ref)
    torch.cuda.synchronize()
    nmv = ms.nctu * ms.nc * ms.nc
    for l in range(3):
        child = ms.level_view(l)[0].reshape(nmv, P.LEVEL_PUS[l] // 4, 4).sum(dim=2)
        parent = ms.level_view(l + 1)[0]
        assert torch.equal(child, parent)
    # spot-check one CTU against the oracle at full picture size
    O = _oracle()
    ctu = 257
    surf, best = O.me_fullsearch(8, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org,
                                 cur.w64, cur.h64, 16, ctu, ctu + 1, ms.cost_host, ms.cost_host)
    rows = slice(ctu * ms.nc, (ctu + 1) * ms.nc)
    assert np.array_equal(_valid(ms.surf.cpu().numpy(), ms)[rows], _valid(surf, ms)[rows])
    assert np.array_equal(ms.best[ctu * 85:(ctu + 1) * 85].cpu().numpy().view(np.uint64), best[ctu * 85:(ctu + 1) * 85])


@pytest.mark.parametrize("packed", [True, "t"])
def test_me_4k_default_config_properties(packed):
    """BASELINE configs[2] size (3840x2160, merange 57, packed records - record-contiguous from the row-walking kernel, chunk-major
    from the record-per-lane kernel), through size-independent properties:
    every parent SAD = sum of its four children at the same mv; best = min over the surface of (sad + mv cost, raster
    index) for every PU of every CTU; three CTUs spot-checked against the oracle at full picture size."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(3840, 2160, 2, depth=8, seed=9)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, 57, 8, dev, packed=packed)
    ms.run(cur, ref)
    torch.cuda.synchronize()
    nmv = ms.nctu * ms.nc * ms.nc
    cost = torch.from_numpy(ms.cost_host.astype(np.int64)).to(dev)
    mvcost = (cost[:, None] + cost[None, :]).reshape(1, ms.nc * ms.nc, 1)            # [mvy * nc + mvx]
    idx = torch.arange(ms.nc * ms.nc, device=dev, dtype=torch.int64).reshape(1, -1, 1)
    views = [ms.level_view(l)[0] for l in range(4)]
    for l in range(3):
        child = views[l].reshape(nmv, P.LEVEL_PUS[l] // 4, 4).sum(dim=2)
        assert torch.equal(child, views[l + 1]), f"level {l + 1} is not the sum of its children"
    for l in range(4):
        n = P.LEVEL_PUS[l]
        key = ((views[l].reshape(ms.nctu, ms.nc * ms.nc, n).to(torch.int64) + mvcost) << 32) | idx
        want = key.min(dim=1).values
        got = ms.level_view(l)[1]
        assert torch.equal(got, want), f"best of level {l} is not the surface minimum"
        del key, want
    O = _oracle()
    for ctu in (0, 1017, ms.nctu - 1):
        _, best = O.me_fullsearch(8, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org,
                                  cur.w64, cur.h64, 57, ctu, ctu + 1, ms.cost_host, ms.cost_host, want_surf=False)
        assert np.array_equal(ms.best[ctu * 85:(ctu + 1) * 85].cpu().numpy().view(np.uint64), best[ctu * 85:(ctu + 1) * 85])


@pytest.mark.parametrize("variant", ["", "v0", "v1", "q0", "q1", "q2", "q4", "q8", "q32", "q62", "q254", "q256", "q446"])
@pytest.mark.parametrize("case", [(192, 128, 57, 4.0, None), (128, 128, 8, 0.0, None), (256, 64, 58, 16.0, None), (128, 64, 5, 4.0, "flat"), (192, 192, 12, 2.0, "centres"),
                                  (128, 128, 80, 1.0, None), (128, 64, 58, 0.0, "noise")])
def test_me_minima_only_launch_every_kernel_variant(case, variant, monkeypatch):
    """The launch the closed loop times: per-PU minima only (no surfaces), 8-bit - the library's default, round 4's kernel (v0 / v1 =
    X265HIP_ME_BEST_VARIANT) and every instantiated flag set of round 5's me_ctu_q2_kernel (qN = X265HIP_ME_Q2_FLAGS) against the oracle: zero
    motion-vector cost (every tie decided by raster order alone), a flat picture (every candidate ties), +-58 = the widest window the 8-bit fast path
    stages (256-byte LDS pitch), +-80 = the widest the picture's margins allow (the generic kernel answers), windows centred per CTU, and uniform noise with zero
    cost (thousands of SAD ties per PU, any candidate of the window can win - found by tools/r5_me_minima_soak.py to be the case that tells)."""
    import torch
    width, height, rng, lam, special = case
    monkeypatch.delenv("X265HIP_ME_BEST_VARIANT", raising=False)
    monkeypatch.delenv("X265HIP_ME_Q2_FLAGS", raising=False)
    if variant.startswith("v"):
        monkeypatch.setenv("X265HIP_ME_BEST_VARIANT", variant[1:])
    elif variant.startswith("q"):
        monkeypatch.setenv("X265HIP_ME_Q2_FLAGS", variant[1:])
        if int(variant[1:]) & 256 and rng > 59:
            pytest.skip("two window copies (flag 256) hold +-59 at most")
    A.lib().x265hip_me_env_refresh()          # the switches are read once per process: this test flips them (tests/conftest.py re-reads them after every test)
    dev = torch.device("cuda:0")
    clip = F.synth_clip(width, height, 2, depth=8, seed=40 + rng)
    y0, y1 = clip[0][0], clip[1][0]
    if special == "flat":
        y0 = np.zeros_like(y0); y1 = np.full_like(y1, 255)
    if special == "noise":
        r = np.random.default_rng(11)
        y0 = r.integers(0, 256, y0.shape).astype(y0.dtype); y1 = r.integers(0, 256, y1.shape).astype(y1.dtype)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(y0, dev)
    ms = P.MotionSearch(cur.w64, cur.h64, rng, 8, dev, want_surf=False, lam=lam)
    O = _oracle()
    if special == "centres":
        r = np.random.default_rng(5)
        cen = r.integers(-20, 21, size=(ms.nctu, 2)).astype(np.int16)
        ms.run(cur, ref, centres=torch.from_numpy(cen).to(dev))
        torch.cuda.synchronize()
        gb = ms.best.cpu().numpy().view(np.uint64).reshape(ms.nctu, 85)
        cw = cur.w64 // 64
        for c in range(ms.nctu):
            o = cur.org + (c // cw) * 64 * cur.stride + (c % cw) * 64
            _, best = O.me_fullsearch(8, cur.host, cur.stride, o, ref.host, ref.stride, o + int(cen[c, 1]) * ref.stride + int(cen[c, 0]), 64, 64, rng, 0, 1,
                                      ms.cost_host, ms.cost_host, want_surf=False, want_best=True)
            assert np.array_equal(gb[c], best.reshape(-1)), (c, variant)
        return
    ms.run(cur, ref)
    torch.cuda.synchronize()
    _, best = O.me_fullsearch(8, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, rng, 0, ms.nctu, ms.cost_host, ms.cost_host,
                              want_surf=False, want_best=True)
    gb = ms.best.cpu().numpy().view(np.uint64)
    assert np.array_equal(gb, best), f"variant {variant!r}: {np.count_nonzero(gb != best)} of {gb.size} minima differ"


@pytest.mark.parametrize("w2", ["0", "1"])
@pytest.mark.parametrize("case", [(192, 128, 57, 4.0, None), (128, 128, 8, 0.0, None), (128, 64, 16, 16.0, None), (128, 64, 5, 4.0, "flat"), (192, 192, 12, 2.0, "centres"),
                                  (128, 64, 72, 1.0, None), (128, 64, 57, 0.0, "noise")])
def test_me_minima_only_launch_10bit_both_kernels(case, w2, monkeypatch):
    """The 16-bit minima-only launch: round 4's me_ctu_w_kernel<best> (X265HIP_ME_W2=0) and round 5's me_ctu_w2_kernel (row constants from an LDS table,
    costX once per group, the 64x64 level four rows at a time, the 16x16 level summed by a transposing butterfly) against the oracle - both LDS pitches
    (+-5 .. +-16: 256 bytes, +-57 / +-72: 512 - +-75 is the widest window the 16-bit fast path stages), ties decided by raster order alone, a flat picture, windows centred per CTU."""
    import torch
    width, height, rng, lam, special = case
    monkeypatch.delenv("X265HIP_ME_BEST_VARIANT", raising=False)
    monkeypatch.setenv("X265HIP_ME_W2", w2)
    A.lib().x265hip_me_env_refresh()
    dev = torch.device("cuda:0")
    clip = F.synth_clip(width, height, 2, depth=10, seed=60 + rng)
    y0, y1 = clip[0][0], clip[1][0]
    if special == "flat":
        y0 = np.zeros_like(y0); y1 = np.full_like(y1, 1023)
    if special == "noise":
        r = np.random.default_rng(12)
        y0 = r.integers(0, 1024, y0.shape).astype(y0.dtype); y1 = r.integers(0, 1024, y1.shape).astype(y1.dtype)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(y0, dev)
    ms = P.MotionSearch(cur.w64, cur.h64, rng, 10, dev, want_surf=False, lam=lam)
    O = _oracle()
    if special == "centres":
        r = np.random.default_rng(6)
        cen = r.integers(-20, 21, size=(ms.nctu, 2)).astype(np.int16)
        ms.run(cur, ref, centres=torch.from_numpy(cen).to(dev))
        torch.cuda.synchronize()
        gb = ms.best.cpu().numpy().view(np.uint64).reshape(ms.nctu, 85)
        cw = cur.w64 // 64
        for c in range(ms.nctu):
            o = cur.org + (c // cw) * 64 * cur.stride + (c % cw) * 64
            _, best = O.me_fullsearch(10, cur.host, cur.stride, o, ref.host, ref.stride, o + int(cen[c, 1]) * ref.stride + int(cen[c, 0]), 64, 64, rng, 0, 1,
                                      ms.cost_host, ms.cost_host, want_surf=False, want_best=True)
            assert np.array_equal(gb[c], best.reshape(-1)), (c, w2)
        return
    ms.run(cur, ref)
    torch.cuda.synchronize()
    _, best = O.me_fullsearch(10, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, rng, 0, ms.nctu, ms.cost_host, ms.cost_host,
                              want_surf=False, want_best=True)
    gb = ms.best.cpu().numpy().view(np.uint64)
    assert np.array_equal(gb, best), f"X265HIP_ME_W2={w2}: {np.count_nonzero(gb != best)} of {gb.size} minima differ"


@pytest.mark.parametrize("depth", [8, 10])
def test_me_minima_xcd_aware_ctu_order_equals_raster_order_and_the_oracle(depth, monkeypatch):
    """Round 6: a whole-picture search launch maps workgroup b to a CTU so that every XCD (b % 8) owns a contiguous band of CTUs (xcd_swizzle; the last
    n % 8 workgroups keep their own index).  60 CTUs = 56 swizzled + 4 not, per-CTU centres included: the minima must equal the raster-order launch
    (X265HIP_ME_XCD_OFF=1) and the oracle."""
    import torch
    dev = torch.device("cuda:0")
    width, height, rng = 640, 384, 24
    clip = F.synth_clip(width, height, 2, depth=depth, seed=77)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    r = np.random.default_rng(6)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("X265HIP_ME_XCD_OFF", "1")
        else:
            monkeypatch.delenv("X265HIP_ME_XCD_OFF", raising=False)
        A.lib().x265hip_me_env_refresh()
        ms = P.MotionSearch(cur.w64, cur.h64, rng, depth, dev, want_surf=False, lam=4.0)
        cen = np.random.default_rng(6).integers(-12, 13, size=(ms.nctu, 2)).astype(np.int16)
        ms.run(cur, ref, centres=torch.from_numpy(cen).to(dev))
        torch.cuda.synchronize()
        outs.append(ms.best.cpu().numpy().view(np.uint64).reshape(ms.nctu, 85).copy())
    assert ms.nctu == 60 and np.array_equal(outs[0], outs[1])
    O = _oracle()
    cw = cur.w64 // 64
    for c in (0, 7, 8, 55, 56, 59):
        o = cur.org + (c // cw) * 64 * cur.stride + (c % cw) * 64
        _, best = O.me_fullsearch(depth, cur.host, cur.stride, o, ref.host, ref.stride, o + int(cen[c, 1]) * ref.stride + int(cen[c, 0]), 64, 64, rng, 0, 1,
                                  ms.cost_host, ms.cost_host, want_surf=False, want_best=True)
        assert np.array_equal(outs[0][c], best.reshape(-1)), c
