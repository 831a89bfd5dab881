import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
dev="cuda:0"
pw, ph, pst = 3840, 2160, 4032
for nj in (1, 8, 24):
    src = torch.randint(0, 255, (nj*pst*ph,), dtype=torch.uint8, device=dev)
    dst = torch.zeros(nj*pst*ph, dtype=torch.uint8, device=dev)
    src16 = torch.randint(0, 1023, (nj*pst*ph,), dtype=torch.int16, device=dev)
    dst16 = torch.zeros(nj*pst*ph, dtype=torch.int16, device=dev)
    jobs = A.make_jobs([([k*pst*ph, k*pst*ph], [0, 1023]) for k in range(nj)], dev)
    def t(fn, it=20):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)/it*1e-3
    for name, kind, depth, s, d, bytes_ in (("cp u8->u8", A.FR_PLANECOPY_CP, 8, src, dst, 2), ("sp u16->u16 (10 bit)", A.FR_PLANECOPY_SP, 10, src16, dst16, 4), ("pp_shr 10 bit", A.FR_PLANECOPY_PP_SHR, 10, src16, dst16, 4)):
        tt = t(lambda: A.frame_batch(kind, depth, pw, ph, [A.Plane(s.data_ptr(), pst), A.Plane(d.data_ptr(), pst)], jobs, nj))
        print(f"{nj:3d} planes {name:22s} {tt*1e6:8.1f} us  {nj*pw*ph*bytes_/tt/1e9:8.1f} GB/s")
    tt = t(lambda: dst.copy_(src))
    print(f"{nj:3d} planes torch copy_ (incl. padding) {tt*1e6:8.1f} us  {2*nj*pst*ph/tt/1e9:8.1f} GB/s")
