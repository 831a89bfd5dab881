"""Debug aid: one small SAO RDO case on the device, the prep kernel's candidate records decoded and checked against a Python restatement."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
HT = importlib.import_module("x265-yuuki-asuna_amd.host_tables")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
import oracle_api as O

def rd(d, bits, lam): return d + ((bits * lam + 128) >> 8)
def i32(v): return ((v + 2**31) % 2**32) - 2**31
def uvlc(code, mx): return 1 + ((code - 1 + (1 if mx > code else 0)) if code else 0)

def main():
    import torch
    from test_oracle_classes_vs_reference import sao_rdo_case
    depth, width, height, slice_type, qp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1, int(sys.argv[4])
    dev = torch.device("cuda:0")
    tabs = HT.load()
    src, rec = sao_rdo_case(depth, width, height, 11, 2 + qp // 12)
    cur = P.DevicePicture(src[0], dev, src[1], src[2]); dbl = P.DevicePicture(rec[0], dev, rec[1], rec[2])
    cw, ch = cur.w64 // 64, cur.h64 // 64
    nctu = cw * ch
    st = [S.Sao(width, height, depth, dev)] + [S.Sao(width // 2, height // 2, depth, dev, ctu=(32, 32), plane_offset=2) for _ in range(2)]
    st[0].stats(cur, dbl.t, cur.stride, cur.org)
    for i in range(2): st[1 + i].stats(None, dbl.c[i], cur.stride_c, cur.org_c, src_plane=cur.c[i])
    lam = HT.sao_lambdas(tabs, qp)
    cm, ct = HT.sao_contexts(slice_type, qp)
    scratch = torch.zeros(A.sao_rdo_scratch_bytes(cw, ch), dtype=torch.uint8, device=dev)
    A.sao_rdo(depth, [s.count for s in st], [s.offset_org for s in st], cw, ch, lam, cm, ct, tabs["entropy_bits"], [s.params for s in st], scratch)
    torch.cuda.synchronize()
    raw = scratch.cpu().numpy().reshape(nctu, 512)
    dist = raw[:, 0:120].copy().view(np.int64).reshape(nctu, 3, 5)
    qy = raw[:, 120:160].copy().view(np.int64).reshape(nctu, 5)
    qc = raw[:, 160:200].copy().view(np.int64).reshape(nctu, 5)
    off = raw[:, 200:440].copy().view(np.int32).reshape(nctu, 3, 5, 4)
    bins = raw[:, 440:500].copy().view(np.int32).reshape(nctu, 3, 5)
    bop = raw[:, 500:512].copy().view(np.int32).reshape(nctu, 3)
    thresh = 1 << min(depth - 5, 5)
    nbad = 0
    for a in range(nctu):
        for pl in range(3):
            cnt = st[pl].count.cpu().numpy().reshape(nctu, 5, 32)[a]; org = st[pl].offset_org.cpu().numpy().reshape(nctu, 5, 32)[a]
            L = lam[1 if pl else 0]
            o_, d_, c_ = np.zeros((5, 32), int), np.zeros((5, 32), int), np.zeros((5, 32), object)
            for t in range(5):
                for c in (range(1, 5) if t < 4 else range(32)):
                    n, e = int(cnt[t, c]), int(org[t, c]); o = 0
                    if n:
                        o = (e * 2 + n) // (n * 2) if e >= 0 else -((-e * 2 + n) // (n * 2))
                        o = max(-thresh + 1, min(thresh - 1, o))
                        if t < 4: o = max(o, 0) if c < 3 else min(o, 0)
                    bo, dc, bc = 0, 0, rd(0, 1, L)
                    while o:
                        rate = abs(o) + (2 if t == 4 else 1)
                        if abs(o) == thresh - 1: rate -= 1
                        dd = i32(i32(i32(n * o) - i32(e * 2)) * o)
                        cc = rd(dd, rate, L)
                        if cc < bc: bc, bo, dc = cc, o, i32(dd)
                        o = o - 1 if o > 0 else o + 1
                    o_[t, c], d_[t, c], c_[t, c] = bo, dc, bc
            for t in range(4):
                ed = int(d_[t, 1:5].sum()); eb = sum(uvlc(int(o_[t, 1 + k]) if k < 2 else -int(o_[t, 1 + k]), thresh - 1) for k in range(4))
                if ed != dist[a, pl, t] or eb != bins[a, pl, t] or list(o_[t, 1:5]) != list(off[a, pl, t]):
                    nbad += 1; print("EO mismatch ctu", a, "plane", pl, "type", t, "dev", dist[a, pl, t], bins[a, pl, t], off[a, pl, t], "exp", ed, eb, o_[t, 1:5])
            cur_ = sum(c_[4, 0:4]); best, pos = cur_, 0
            for i in range(1, 29):
                cur_ += c_[4, i + 3] - c_[4, i - 1]
                if cur_ < best: best, pos = cur_, i
            ed = int(d_[4, pos:pos + 4].sum()); eb = 5 + sum(uvlc(abs(int(o_[4, pos + k])), thresh - 1) + (1 if o_[4, pos + k] else 0) for k in range(4))
            if ed != dist[a, pl, 4] or eb != bins[a, pl, 4] or pos != bop[a, pl] or list(o_[4, pos:pos + 4]) != list(off[a, pl, 4]):
                nbad += 1; print("BO mismatch ctu", a, "plane", pl, "dev", dist[a, pl, 4], bins[a, pl, 4], bop[a, pl], off[a, pl, 4], "exp", ed, eb, pos, o_[4, pos:pos + 4])
    print("prep mismatches:", nbad)
    lamarr = np.tile(np.array(lam, np.int64), (nctu, 1))
    ep, _ = O.sao_rdo(depth, [s.count.cpu().numpy() for s in st], [s.offset_org.cpu().numpy() for s in st], cw, ch, lamarr, cm, ct, tabs["entropy_bits"])
    for pl in range(3):
        got = st[pl].params.cpu().numpy().reshape(nctu, 7)
        for a in range(nctu):
            if (got[a] != ep[pl][a]).any(): print("param mismatch plane", pl, "ctu", a, "(row", a // cw, "col", a % cw, ") dev", got[a], "oracle", ep[pl][a]); break

if __name__ == "__main__":
    main()
