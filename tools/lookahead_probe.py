#!/usr/bin/env python3
"""[bench.py --lookahead-probe: part of bench.py's cpu_baseline leg - loads oracle/ only to time the CPU path beside the kernel]

Latency / throughput probe of x265hip_lowres_cost (the lookahead's P-frame cost estimate): one picture pair, then batches of
independent pairs in one launch (the picture itself is a wavefront of dependent 8x8 blocks, one workgroup walks it)."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
dev = torch.device("cuda:0")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
clip = F.synth_clip(W, H, 2, depth=8, seed=5)
cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
lc, lr = S.Lookahead(W, H, 8, dev), S.Lookahead(W, H, 8, dev)
lc.run(cur); lr.run(ref)
torch.cuda.synchronize()
print(f"# x265hip_lowres_cost: {W}x{H} source, {lc.wcu} x {lc.hcu} blocks of 8x8 at half resolution, {lc.wcu + 2 * (lc.hcu - 1)} wavefront steps; "
      f"CPU column = oracle/x265_oracle_search.c::x265oracle_lowres_cost (serial by construction, -march=x86-64-v3), one thread per picture")
nmax = 64
stages = [S.LookaheadCost(lc, dev) for _ in range(nmax)]
for n in (1, 4, 16, 64, 256, 512):
    while len(stages) < n:
        stages.append(S.LookaheadCost(lc, dev))
    def go():
        S.LookaheadCost.run_batch(stages[:n], [lc] * n, [lr] * n)
    go(); torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    print(f"GPU {n:3d} picture pair(s) per launch: {t * 1e3:8.2f} ms per batch -> {n / t:8.1f} pairs/s ({n * lc.wcu * lc.hcu / t / 1e6:.2f} M blocks/s)", flush=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_api as O
from bench import effective_cpus
dt = clip[0][0].dtype
cp = lc.planes[0].cpu().numpy().view(dt)
rp = [p.cpu().numpy().view(dt) for p in lr.planes]
cq = stages[0].cost_q.cpu().numpy().view(np.uint16)
ic = lc.intra_cost.cpu().numpy()
avx2 = O.host_has_avx2()
O.lowres_cost(8, cp, rp, lc.stride, lc.org, lc.wcu, lc.hcu, cq, stages[0].qoff, ic, avx2=avx2)
t0 = time.perf_counter()
res = O.lowres_cost(8, cp, rp, lc.stride, lc.org, lc.wcu, lc.hcu, cq, stages[0].qoff, ic, avx2=avx2)
tc = time.perf_counter() - t0
thr = effective_cpus()
print(f"CPU one picture pair on one thread: {tc * 1e3:.1f} ms -> {1 / tc:.1f} pairs/s per thread, {thr / tc:.1f} pairs/s if {thr} threads each take a picture")
assert np.array_equal(stages[0].frame.cpu().numpy()[:3], res[4]), "GPU and CPU frame costs differ"
print(f"frame cost (costEst, costEstAq, intraMbs) = {res[4].tolist()} on both")

# ---- the other lookahead stages of the same picture: adaptive-quantisation energies, weighted-reference analysis
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
yimg, cbimg, crimg = clip[1]
pad = lambda a: np.ascontiguousarray(np.pad(a, 16, mode="edge"))
cbp, crp = pad(cbimg), pad(crimg)
d_cb, d_cr = torch.from_numpy(cbp.reshape(-1)).to(dev), torch.from_numpy(crp.reshape(-1)).to(dev)
aq = S.AdaptiveQuant(W, H, 8, dev, qg_size=16, aq_mode=2)
t_aq = timed(lambda: A.aq_energy(8, cur.t, cur.stride, cur.org, W, H, 16, aq.energy, aq.wp, d_cb, d_cr, cbp.shape[1], 16 * cbp.shape[1] + 16))
px = W * H * 3 // 2
print(f"x265hip_aq_energy (16x16 blocks, 4:2:0): {t_aq * 1e3:.1f} us per picture -> {px / t_aq / 1e6:.1f} GB/s of source samples")
t0 = time.perf_counter(); qp, inv, wsum, wssd = aq.run(cur, d_cb, d_cr, cbp.shape[1], 16 * cbp.shape[1] + 16); t_host = time.perf_counter() - t0
e_ref = O.aq_frame(8, cur.host.reshape(-1), cur.stride, cur.org, W, H, cbp.reshape(-1), crp.reshape(-1), cbp.shape[1], 16 * cbp.shape[1] + 16, 16, 2, 1.0, True)
assert np.array_equal(qp, e_ref[1]) and np.array_equal(inv, e_ref[2]), "AQ offsets differ from the CPU restatement"
t0 = time.perf_counter(); O.aq_frame(8, cur.host.reshape(-1), cur.stride, cur.org, W, H, cbp.reshape(-1), crp.reshape(-1), cbp.shape[1], 16 * cbp.shape[1] + 16, 16, 2, 1.0, True); tc = time.perf_counter() - t0
print(f"stages.AdaptiveQuant.run (launch, copy back, {len(qp)} double-precision offsets on the host): {t_host * 1e3:.1f} ms; CPU restatement of the whole pass (one thread): {tc * 1e3:.1f} ms")
cost = torch.zeros(4, dtype=torch.int32, device=dev)
cands = [None, (60, 6, 3), (3, 2, 6), (83, 6, -19)]
t_w = timed(lambda: A.lowres_weight_cost(8, lc.planes[0], lr.planes[0], lc.stride, lc.org, lc.width, lc.lines, lc.intra_cost, cands, cost))
t0 = time.perf_counter(); ec = [O.lowres_weight_cost(8, cp, rp[0], lc.stride, lc.org, lc.width, lc.lines, ic, c) for c in cands]; tc = time.perf_counter() - t0
assert cost.cpu().numpy().view(np.uint32).tolist() == ec
print(f"x265hip_lowres_weight_cost (4 candidate weights, {lc.wcu * lc.hcu} blocks each): {t_w * 1e3:.1f} us per launch; CPU restatement (one thread): {tc * 1e3:.1f} ms")
wa = S.WeightAnalysis(lc, dev)
t_a = timed(lambda: A.lowres_weight_apply(8, lr.planes, wa.weighted, lc.stride, lc.lines + 2 * lc.my, (60, 6, 3)))
print(f"x265hip_lowres_weight_apply (four planes): {t_a * 1e3:.1f} us -> {2 * 4 * lr.planes[0].numel() / t_a / 1e6:.1f} GB/s read + written")
