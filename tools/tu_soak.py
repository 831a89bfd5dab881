#!/usr/bin/env python3
"""Randomised soak of the bi-predictive / weighted TU stages (luma + chroma) and of --hevc-aq against the oracle: random weights over the
whole legal range, block sizes, bit depths, directions, fractional phases, picture sizes.  python tools/tu_soak.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
import oracle_api as O


def rand_weight(rng):
    if rng.integers(0, 5) == 0:
        return None
    return (int(rng.integers(0, 2)), int(rng.integers(-128, 128)), int(rng.integers(-128, 128)), int(rng.integers(0, 8)))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        depth, level = int(rng.choice([8, 8, 10, 12])), int(rng.integers(0, 3))
        W, Hh = 64 * int(rng.integers(1, 5)), 64 * int(rng.integers(1, 4))
        qp = int(rng.integers(10, 40)) + 6 * (depth - 8)
        clip = F.synth_clip(W, Hh, 3, depth=depth, seed=int(rng.integers(1, 1 << 30)))
        pics = [P.DevicePicture(c[0], dev, c[1], c[2]) for c in clip]
        cur, r0, r1 = pics[1], pics[0], pics[2]
        nctu = (cur.w64 // 64) * (cur.h64 // 64)
        mvs = []
        for _ in range(2):
            qx, qy = rng.integers(-60, 61, size=nctu * 85), rng.integers(-60, 61, size=nctu * 85)
            qx[::4] &= ~7; qy[::3] &= ~7
            m = np.zeros((nctu * 85, 2), np.int32)
            m[:, 1] = (qx & 0xffff) | (qy << 16)
            mvs.append(m)
        nblk = (64 >> (3 + level)) ** 2
        dirs = rng.integers(1, 4, size=nctu * nblk).astype(np.uint8)
        d_mv = [torch.from_numpy(m.reshape(-1)).to(dev) for m in mvs]
        d_dir = torch.from_numpy(dirs).to(dev)
        w = (rand_weight(rng), rand_weight(rng))
        weights = None if w == (None, None) else w
        flags = int(rng.choice([0, 2]))
        desc = f"{W}x{Hh} d{depth} level {level} qp {qp} weights {w} flags {flags}"
        st = S.InterReconBi(nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=flags)
        recon = torch.zeros_like(cur.t)
        st.run(cur, r0, r1, recon, d_mv[0], d_mv[1], dir_flags=d_dir, weights=weights)
        erec, elev, ens, edist = O.inter_recon_bi(depth, cur.host.reshape(-1), cur.stride, cur.org, r0.host.reshape(-1), r1.host.reshape(-1), cur.w64, cur.h64, level,
                                                  mvs[0], mvs[1], qp, dir_flags=dirs, intra_slice=flags, weights=weights)
        torch.cuda.synchronize()
        bad = []
        if not np.array_equal(st.levels.cpu().numpy(), elev): bad.append(f"luma levels ({np.count_nonzero(st.levels.cpu().numpy() != elev)})")
        if not np.array_equal(recon.cpu().numpy().view(cur.host.dtype).reshape(-1), erec.reshape(-1)): bad.append("luma recon")
        if not np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist): bad.append("luma dist")
        ok = not bad
        c = int(rng.integers(0, 2))
        sc = S.InterReconChromaBi(nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=flags)
        rc = torch.zeros_like(cur.c[c])
        sc.run(cur.c[c], r0.c[c], r1.c[c], rc, cur.stride_c, cur.org_c, d_mv[0], d_mv[1], dir_flags=d_dir, weights=weights)
        crec, clev, _, cdist = O.inter_recon_chroma_bi(depth, cur.c_host[c].reshape(-1), r0.c_host[c].reshape(-1), r1.c_host[c].reshape(-1), cur.stride_c, cur.org_c,
                                                       cur.w64, cur.h64, level, mvs[0], mvs[1], qp, dir_flags=dirs, intra_slice=flags, weights=weights)
        torch.cuda.synchronize()
        if not np.array_equal(sc.levels.cpu().numpy(), clev): bad.append(f"chroma levels ({np.count_nonzero(sc.levels.cpu().numpy() != clev)})")
        if not np.array_equal(rc.cpu().numpy().view(cur.host.dtype).reshape(-1), crec.reshape(-1)): bad.append("chroma recon")
        if not np.array_equal(sc.dist.cpu().numpy().view(np.uint64), cdist): bad.append("chroma dist")
        ok = not bad
        # --hevc-aq on a picture of any even size
        w2, h2, qg, rg = 2 * int(rng.integers(20, 160)), 2 * int(rng.integers(20, 120)), int(rng.choice([8, 16, 32, 64])), float(rng.uniform(1.0, 6.0))
        yimg = F.synth_clip(w2, h2, 1, depth=depth, seed=int(rng.integers(1, 1 << 30)))[0][0]
        pic = P.DevicePicture(yimg, dev)
        layers, inv, _, _ = S.HevcAq(w2, h2, depth, dev, qg_size=qg, qp_adaptation_range=rg, weightp=False).run(pic)
        parts, act, qpo, avg, einv, _, _ = O.aq_hevc_frame(depth, pic.host.reshape(-1), pic.stride, pic.org, w2, h2, qg_size=qg, qp_adaptation_range=rg, weightp=False)
        at = 0
        for d in range(4):
            if parts[d]:
                a, q, g = layers[64 >> d]
                if not (np.array_equal(a, act[at:at + parts[d]]) and np.array_equal(q, qpo[at:at + parts[d]]) and g == avg[d]): bad.append(f"hevc-aq layer {d}")
                at += parts[d]
        if not np.array_equal(inv, einv): bad.append("hevc-aq invQscale")
        ok = not bad
        if not ok:
            print(f"MISMATCH {bad}: {desc}; hevc-aq {w2}x{h2} qg {qg} range {rg}", flush=True)
            sys.exit(1)
        cases += 1
    print(f"tu soak: {cases} random cases (weighted bi luma + chroma stage, --hevc-aq) equal to the oracle ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
