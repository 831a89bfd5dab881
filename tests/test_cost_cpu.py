"""CPU checks of the sub-sample cost tables (round 6): the product's host-side lists (PU shapes, refinement position sets) against the
oracle's restatement, and the oracle's table route - the reference's own subpelCompare steps through the pinned primitive table
(oracle/x265_oracle_pipeline8.c) - against an independent one: 4x4 Hadamard tiles of the oracle's phase planes in numpy, which is the
route the HIP kernel takes.  The GPU parity proper is tests/test_gpu_cost.py; the values are pinned against the REAL reference's
MotionEstimate::subpelCompare in flight by tests/test_seam_cpu.py (every served comparison re-evaluated by the reference)."""
import importlib

import numpy as np
import pytest

import cost_oracle as C

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
F = importlib.import_module("x265-yuuki-asuna_amd.frames")


def test_product_lists_equal_the_oracle_lists():
    O = C.oracle()
    for shapes, n in ((0, 85), (1, 169), (2, 209)):
        got, want = A.cost_pu_list(shapes), O.cost_pu_list(shapes)
        assert got.shape == (n, 4) and np.array_equal(got, want[:, :4])
        assert (want[:, 4] >= 1).all(), "every listed PU is one of the reference's 25 partition sizes (primitives.h:41-55), none of them 4x4"
        assert ((got[:, 0] % 8 == 0) & (got[:, 1] % 8 == 0) & (got[:, 2] % 8 == 0) & (got[:, 3] % 8 == 0)).all()
    for subme, n in ((0, 5), (1, 21), (2, 21), (3, 49), (4, 85), (5, 49), (6, 121), (7, 169)):
        got = A.cost_positions(subme)
        assert len(got) == n and np.array_equal(got, O.cost_positions(subme))
        assert A.cost_record_bytes(subme) == (8 + 2 * n + 3) // 4 * 4 == O.cost_record_bytes(subme)
        assert A.cost_record_bytes(subme, 1) == (8 + 2 * n + 3) // 4 * 4 + (4 + 2 * n + 3) // 4 * 4 == O.cost_record_bytes(subme, 1)
    assert A.cost_ctu_bytes(3, 1, 1) == 169 * 108 and A.cost_ctu_bytes(4, 2, 2) == 209 * 2 * 180 and A.cost_ctu_bytes(4, 2, 2, 1) == 209 * 2 * (180 + 176)
    assert A.lib().x265hip_cost_pu_count(3) == 0 and A.cost_ctu_bytes(9, 0, 1) == 0


def test_position_set_is_what_the_refinement_loop_can_reach():
    """Replay of motion.cpp:1515-1561 with an adversarial cost function: whichever neighbour the loop moves to, every vector it measures is in
    the set (and every member of the set is measured by some run)."""
    wl = {0: (1, 4, 0, 4), 1: (1, 4, 1, 4), 2: (1, 4, 1, 4), 3: (2, 4, 1, 4), 4: (2, 4, 2, 4), 5: (1, 8, 1, 8), 6: (2, 8, 1, 8), 7: (2, 8, 2, 8)}
    sq = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1)]
    rng = np.random.default_rng(5)
    for subme, (hi, hd, qi, qd) in wl.items():
        allowed = {tuple(p) for p in A.cost_positions(subme).tolist()}
        seen = set()
        for _ in range(3000):
            bmv = (0, 0)
            seen.add(bmv)
            for iters, dirs, step in ((hi, hd, 2), (qi, qd, 1)):
                for _it in range(iters):
                    cands = [(bmv[0] + step * dx, bmv[1] + step * dy) for dx, dy in sq[:dirs]]
                    seen.update(cands)
                    pick = int(rng.integers(0, dirs + 1))
                    if pick == dirs:
                        break
                    bmv = cands[pick]
        assert seen <= allowed, (subme, sorted(seen - allowed))
        assert seen == allowed, (subme, sorted(allowed - seen))


def _hadamard_tiles(diff):
    """sum over 4x4 tiles of (sum |H d H'|) >> 1 for an (h, w) int array, h and w multiples of 4."""
    h4 = np.array([[1, 1, 1, 1], [1, -1, 1, -1], [1, 1, -1, -1], [1, -1, -1, 1]], np.int64)
    h, w = diff.shape
    t = diff.reshape(h // 4, 4, w // 4, 4).transpose(0, 2, 1, 3).astype(np.int64)
    coef = h4 @ t @ h4.T
    return int((np.abs(coef).sum(axis=(2, 3)) >> 1).sum())


@pytest.mark.parametrize("depth,chroma,subme", [(8, 1, 3), (10, 1, 4), (12, 0, 3), (8, 0, 7)])
def test_oracle_tables_equal_tilewise_satd_of_the_phase_planes(depth, chroma, subme):
    O = C.oracle()
    clip = F.synth_clip(128, 64, 2, depth=depth, seed=600 + depth)
    fenc, ref = C.picture(clip[1]), C.picture(clip[0])
    g = fenc
    shapes, k = 2, 2
    rng = np.random.default_rng(depth)
    npu, nctu = 209, 2
    cand = rng.integers(-6, 7, (nctu, npu, k, 2)).astype(np.int16)
    cand[0, 3, 1, 0] = -32768
    tables = O.cost_tables(depth, [fenc["y"], fenc["cb"], fenc["cr"]], [ref["y"], ref["cb"], ref["cr"]], g["stride"], g["stride_c"], g["margin_x"], g["margin_y"],
                           g["margin_y_c"], g["width"], 0, 1, shapes, k, subme, chroma, cand, sad_costs=1)
    mv, cost = C.parse_records(tables, subme)
    _, cost_sad = C.parse_records(tables, subme, sad_typed=True)
    assert mv[0, 3, 1, 0] == -32768
    ph = [O.phase_planes(depth, ref["y"], g["stride"], g["rows"]), O.phase_planes(depth, ref["cb"], g["stride_c"], g["rows_c"], chroma=True),
          O.phase_planes(depth, ref["cr"], g["stride_c"], g["rows_c"], chroma=True)]
    src = [ref["y"].reshape(g["rows"], g["stride"]), ref["cb"].reshape(g["rows_c"], g["stride_c"]), ref["cr"].reshape(g["rows_c"], g["stride_c"])]
    fsrc = [fenc["y"].reshape(g["rows"], g["stride"]), fenc["cb"].reshape(g["rows_c"], g["stride_c"]), fenc["cr"].reshape(g["rows_c"], g["stride_c"])]
    pos = O.cost_positions(subme)
    rects = O.cost_pu_list(shapes)
    checked = 0
    for ctu, pu, kk in [(0, 84, 0), (1, 0, 1), (0, 100, 1), (1, 150, 0), (0, 170, 0), (1, 208, 1), (0, 63, 0), (1, 90, 0), (0, 181, 1), (1, 199, 0)]:
        x, y, w, h, _ = rects[pu]
        assert tuple(mv[ctu, pu, kk]) == tuple(cand[ctu, pu, kk])
        for i in range(0, len(pos), 3):
            qx, qy = int(cand[ctu, pu, kk, 0]) * 4 + int(pos[i, 0]), int(cand[ctu, pu, kk, 1]) * 4 + int(pos[i, 1])
            X, Y = g["margin_x"] + ctu * 64 + x, g["margin_y"] + y
            p = (qy & 3) * 4 + (qx & 3)
            plane = ph[0][p - 1] if p else src[0]
            dl = fsrc[0][Y:Y + h, X:X + w].astype(np.int64) - plane[Y + (qy >> 2):Y + (qy >> 2) + h, X + (qx >> 2):X + (qx >> 2) + w]
            want = _hadamard_tiles(dl)
            want_sad = int(np.abs(dl).sum())
            if chroma:
                Xc, Yc = g["margin_x"] + (ctu * 64 + x) // 2, g["margin_y_c"] + y // 2
                p = (qy & 7) * 8 + (qx & 7)
                for c in (1, 2):
                    plane = ph[c][p - 1] if p else src[c]
                    ch = _hadamard_tiles(fsrc[c][Yc:Yc + h // 2, Xc:Xc + w // 2].astype(np.int64) -
                                         plane[Yc + (qy >> 3):Yc + (qy >> 3) + h // 2, Xc + (qx >> 3):Xc + (qx >> 3) + w // 2])
                    want += ch
                    want_sad += ch          # subpelCompare adds chromaSatd whatever the luma comparison is (motion.cpp:1601-1661)
            if cost_sad[ctu, pu, kk, i] != 0xffffffff:
                assert cost_sad[ctu, pu, kk, i] == want_sad, (ctu, pu, kk, i, int(cost_sad[ctu, pu, kk, i]), want_sad)
            if cost[ctu, pu, kk, i] == 0xffffffff:          # a delta that does not fit 16 bits (random vectors on a 64x64 block at 10 bits): the host's to compute
                assert want - int(cost[ctu, pu, kk].min()) >= 65535
                continue
            assert cost[ctu, pu, kk, i] == want, (ctu, pu, kk, i, int(cost[ctu, pu, kk, i]), want)
            checked += 1
    assert checked > 100


def test_oracle_candidates_take_the_smallest_sads_in_scan_order():
    O = C.oracle()
    rng = np.random.default_rng(11)
    window, nctu = 3, 2
    nc, ng = 7, 2
    s8 = rng.integers(0, 4, (nctu, nc, ng * 4, 64)).astype(np.int64)          # few distinct values: ties everywhere
    lv = [s8]
    for n in (16, 4, 1):                                                       # z-order: four consecutive children make a parent
        lv.append(lv[-1].reshape(nctu, nc, ng * 4, n, 4).sum(axis=4))
    surf = np.concatenate(lv, axis=3).reshape(nctu, nc, ng, 4, 85).transpose(0, 1, 2, 4, 3).astype(np.int32)
    centres = np.array([[2, -1], [-3, 4]], np.int16)
    cand = O.cost_candidates(np.ascontiguousarray(surf), centres, nctu, window, 1, 2)
    rects = O.cost_pu_list(1)
    full = np.concatenate(lv, axis=3)                                          # [ctu][row][col][85]
    for ctu in range(nctu):
        for pu in (0, 63, 64, 84, 85, 100, 168):
            x, y, w, h, _ = rects[pu]
            sad = np.zeros((nc, nc), np.int64)
            for by in range(y // 8, (y + h) // 8):
                for bx in range(x // 8, (x + w) // 8):
                    z = sum(((bx >> b) & 1) << (2 * b) | ((by >> b) & 1) << (2 * b + 1) for b in range(3))
                    sad += full[ctu, :, :nc, z]
            order = np.argsort(sad.reshape(-1), kind="stable")[:2]
            for kk in range(2):
                assert tuple(cand[ctu, pu, kk]) == (centres[ctu, 0] + order[kk] % nc - window, centres[ctu, 1] + order[kk] // nc - window)
