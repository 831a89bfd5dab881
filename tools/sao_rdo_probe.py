"""Times x265hip_sao_rdo (HIP events) on random-but-plausible statistics at 4K and 8K geometry, next to the distortion-only stand-in.  Measurement aid."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
HT = importlib.import_module("x265-yuuki-asuna_amd.host_tables")


def main():
    import torch
    dev = torch.device("cuda:0")
    tabs = HT.load()
    rng = np.random.default_rng(5)
    for name, cw, ch in (("4K", 60, 34), ("8K", 120, 68), ("1080p", 30, 17)):
        nctu = cw * ch
        cnt = [torch.from_numpy(rng.integers(0, 600, size=nctu * 160).astype(np.int32)).to(dev) for _ in range(3)]
        org = [torch.from_numpy(rng.integers(-900, 900, size=nctu * 160).astype(np.int32)).to(dev) for _ in range(3)]
        par = [torch.zeros(nctu * 7, dtype=torch.int32, device=dev) for _ in range(3)]
        scratch = torch.zeros(A.sao_rdo_scratch_bytes(cw, ch), dtype=torch.uint8, device=dev)
        lam = HT.sao_lambdas(tabs, 27)
        cm, ct = HT.sao_contexts(1, 27)
        run = lambda: A.sao_rdo(8, cnt, org, cw, ch, lam, cm, ct, tabs["entropy_bits"], par, scratch)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        t_rdo = e0.elapsed_time(e1) / 20
        e0.record()
        for _ in range(20):
            for i in range(3):
                A.sao_decide(8, cnt[i], org[i], nctu, par[i])
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: x265hip_sao_rdo (3 planes, {cw} x {ch} CTUs) {t_rdo * 1e3:.1f} us; stand-in x265hip_sao_decide x 3 {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
