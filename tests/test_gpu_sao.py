"""GPU parity: the SAO pixel passes (x265hip_sao_stats / x265hip_sao_apply) vs the oracle's restatement of SAO::calcSaoStatsCTU and
SAO::generateLumaOffsets / applyPixelOffsets (sao.cpp:735-917, 572-630, 274-570), which tests/test_oracle_classes_vs_reference.py pins
against the real SAO class."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
H = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def _case(depth, width, height, seed):
    from test_oracle_classes_vs_reference import sao_case
    return sao_case(depth, width, height, seed)


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 200, 150), (10, 192, 136), (8, 64, 64), (8, 1920, 1080), (10, 832, 480), (8, 70, 66), (12, 200, 150)])
def test_sao_passes_match_oracle(depth, width, height):
    import torch
    dev = torch.device("cuda:0")
    y, rec, params = _case(depth, width, height, 6)
    fenc, stride, org, w64, h64 = F.pad_plane(y)
    recp = F.pad_plane(rec)[0]
    nctu = params.shape[0]
    dt = y.dtype
    d_f = torch.from_numpy(fenc.view(np.uint8).reshape(-1)).to(dev)
    d_r = torch.from_numpy(recp.view(np.uint8).reshape(-1)).to(dev)
    d_cnt = torch.full((nctu * 160,), -1, dtype=torch.int32, device=dev)
    d_off = torch.full((nctu * 160,), -1, dtype=torch.int32, device=dev)
    H.sao_stats(depth, d_f, stride, org, d_r, stride, org, width, height, d_cnt, d_off)
    d_out = d_r.clone()
    d_par = torch.from_numpy(params.reshape(-1)).to(dev)
    H.sao_apply(depth, d_r, stride, org, d_out, stride, org, width, height, d_par)
    torch.cuda.synchronize()
    O = _oracle()
    cnt, off = O.sao_stats(depth, fenc, recp, stride, org, width, height)
    gc, go = d_cnt.cpu().numpy().reshape(nctu, 5, 32), d_off.cpu().numpy().reshape(nctu, 5, 32)
    assert np.array_equal(gc, cnt), f"count differs in CTU/type {np.argwhere((gc != cnt).any(axis=2))[:6].tolist()}"
    assert np.array_equal(go, off), f"offsetOrg differs in CTU/type {np.argwhere((go != off).any(axis=2))[:6].tolist()}"
    out = O.sao_apply(depth, recp, stride, org, width, height, params)
    gout = d_out.cpu().numpy().view(dt).reshape(out.shape)
    assert np.array_equal(gout, out), f"{np.count_nonzero(gout != out)} samples differ"
    assert (out != recp).any() and cnt[:, :4, :5].sum() > 0 and cnt[:, 4].sum() > 0


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (8, 200, 152), (10, 192, 136), (8, 1920, 1080)])
def test_sao_chroma_planes_match_oracle(depth, width, height):
    """The chroma planes of a 4:2:0 picture: 32x32 CTU footprint, plane_offset 2."""
    import torch
    from test_oracle_classes_vs_reference import sao_chroma_case, pad_any
    dev = torch.device("cuda:0")
    src, rec, params = sao_chroma_case(depth, width, height, 8)
    cw, ch = width // 2, height // 2
    nctu = params[0].shape[0]
    O = _oracle()
    for c in range(2):
        fp, st, og = pad_any(src[c])
        rp = pad_any(rec[c])[0]
        d_f = torch.from_numpy(fp.view(np.uint8)).to(dev)
        d_r = torch.from_numpy(rp.view(np.uint8)).to(dev)
        d_cnt = torch.full((nctu * 160,), -1, dtype=torch.int32, device=dev)
        d_off = torch.full((nctu * 160,), -1, dtype=torch.int32, device=dev)
        H.sao_stats(depth, d_f, st, og, d_r, st, og, cw, ch, d_cnt, d_off, ctu=(32, 32), plane_offset=2)
        d_out = d_r.clone()
        H.sao_apply(depth, d_r, st, og, d_out, st, og, cw, ch, torch.from_numpy(params[c].reshape(-1)).to(dev), ctu=(32, 32))
        torch.cuda.synchronize()
        cnt, off = O.sao_stats(depth, fp, rp, st, og, cw, ch, ctu=(32, 32), plane_offset=2)
        assert np.array_equal(d_cnt.cpu().numpy().reshape(cnt.shape), cnt) and np.array_equal(d_off.cpu().numpy().reshape(off.shape), off)
        out = O.sao_apply(depth, rp, st, og, cw, ch, params[c], ctu=(32, 32))
        assert np.array_equal(d_out.cpu().numpy().view(rp.dtype), out)


def test_sao_apply_refuses_in_place():
    import torch
    dev = torch.device("cuda:0")
    t = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
    par = torch.zeros(7, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError):
        H.sao_apply(8, t, 256, 0, t, 256, 0, 64, 64, par)


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (10, 192, 136), (8, 1920, 1080), (12, 200, 150)])
def test_sao_decide_matches_oracle_and_closes_the_loop(depth, width, height):
    """x265hip_sao_decide (saoStatsInitialOffset + the distortion-only type choice) on the device's own statistics equals the oracle
    (whose initial offsets are pinned against the real SAO class), also on random statistics incl. empty classes and huge sums; the
    chosen parameters then go through x265hip_sao_apply and must lower the distortion against the source."""
    import torch
    dev = torch.device("cuda:0")
    y, rec, _ = _case(depth, width, height, 8)
    fenc, stride, org, w64, h64 = F.pad_plane(y)
    recp = F.pad_plane(rec)[0]
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    d_f = torch.from_numpy(fenc.view(np.uint8).reshape(-1)).to(dev)
    d_r = torch.from_numpy(recp.view(np.uint8).reshape(-1)).to(dev)
    d_cnt = torch.zeros(nctu * 160, dtype=torch.int32, device=dev)
    d_off = torch.zeros(nctu * 160, dtype=torch.int32, device=dev)
    H.sao_stats(depth, d_f, stride, org, d_r, stride, org, width, height, d_cnt, d_off)
    d_par = torch.full((nctu * 7,), 99, dtype=torch.int32, device=dev)
    d_init = torch.full((nctu * 160,), 99, dtype=torch.int32, device=dev)
    H.sao_decide(depth, d_cnt, d_off, nctu, d_par, init_offset=d_init)
    d_out = d_r.clone()
    H.sao_apply(depth, d_r, stride, org, d_out, stride, org, width, height, d_par)
    torch.cuda.synchronize()
    O = _oracle()
    init, params = O.sao_decide(depth, d_cnt.cpu().numpy(), d_off.cpu().numpy())
    assert np.array_equal(d_par.cpu().numpy().reshape(nctu, 7), params)
    assert np.array_equal(d_init.cpu().numpy().reshape(nctu, 5, 32), init)
    assert (params[:, 0] >= 0).any()
    out = O.sao_apply(depth, recp, stride, org, width, height, params)
    gout = d_out.cpu().numpy().view(y.dtype).reshape(out.shape)
    assert np.array_equal(gout, out)
    r0 = org // stride
    a = fenc[r0:r0 + height, org % stride:org % stride + width].astype(np.int64)
    sse = lambda p: int(((p.reshape(fenc.shape)[r0:r0 + height, org % stride:org % stride + width].astype(np.int64) - a) ** 2).sum())
    assert sse(out) < sse(recp), "the chosen offsets must lower the distortion"
    # random statistics: empty classes, negative / large sums
    rng = np.random.default_rng([9, depth, width])
    n = 300
    cnt = rng.integers(0, 4097, size=(n, 5, 32)).astype(np.int32)
    cnt[rng.random(cnt.shape) < 0.3] = 0
    off = (rng.integers(-40, 41, size=(n, 5, 32)) * cnt * rng.random((n, 5, 32))).astype(np.int32)
    off[::7] *= 3
    g_par = torch.zeros(n * 7, dtype=torch.int32, device=dev)
    g_init = torch.zeros(n * 160, dtype=torch.int32, device=dev)
    H.sao_decide(depth, torch.from_numpy(cnt.reshape(-1)).to(dev), torch.from_numpy(off.reshape(-1)).to(dev), n, g_par, init_offset=g_init)
    torch.cuda.synchronize()
    init, params = O.sao_decide(depth, cnt, off)
    assert np.array_equal(g_par.cpu().numpy().reshape(n, 7), params) and np.array_equal(g_init.cpu().numpy().reshape(n, 5, 32), init)
    assert len(set(params[:, 0].tolist())) >= 4


@pytest.mark.parametrize("depth,width,height", [(8, 256, 128), (10, 192, 136), (8, 1920, 1080), (8, 200, 152)])
def test_sao_planes_fused_equals_the_single_plane_entries(depth, width, height):
    """x265hip_sao_planes (Y + two half-size planes, one launch per SAO step) against the per-plane x265hip_sao_stats / _decide / _apply
    the tests above pin on the oracle: statistics, parameters and filtered planes identical."""
    import torch
    dev = torch.device("cuda:0")
    S = importlib.import_module("x265-yuuki-asuna_amd.stages")
    geo = [(width, height, (64, 64), 0), (width // 2, height // 2, (32, 32), 2), (width // 2, height // 2, (32, 32), 2)]
    fused, single, bufs = [], [], []
    for i, (w, h, ctu, po) in enumerate(geo):
        y, rec, _ = _case(depth, w, h, 11 + i)
        fenc, stride, org, _, _ = F.pad_plane(y)
        recp = F.pad_plane(rec)[0]
        d_f = torch.from_numpy(fenc.view(np.uint8).reshape(-1)).to(dev)
        d_r = torch.from_numpy(recp.view(np.uint8).reshape(-1)).to(dev)
        a, b = S.Sao(w, h, depth, dev, ctu=ctu, plane_offset=po), S.Sao(w, h, depth, dev, ctu=ctu, plane_offset=po)
        oa, ob = d_r.clone(), d_r.clone()
        if depth > 8:
            d_f, d_r, oa, ob = d_f.view(torch.int16), d_r.view(torch.int16), oa.view(torch.int16), ob.view(torch.int16)
        fused.append(a.plane(d_f, stride, org, d_r, stride, org, oa))
        single.append((b, d_f, d_r, stride, org, ob))
        bufs.append((a, oa, b, ob))
    H.sao_planes(depth, fused)
    for b, d_f, d_r, stride, org, ob in single:
        b.stats(None, d_r, stride, org, src_plane=d_f)
        b.decide()
        b.apply(d_r, stride, org, ob)
    torch.cuda.synchronize()
    for i, (a, oa, b, ob) in enumerate(bufs):
        assert torch.equal(a.count, b.count) and torch.equal(a.offset_org, b.offset_org), f"plane {i}: statistics differ"
        assert torch.equal(a.params, b.params), f"plane {i}: parameters differ"
        assert torch.equal(oa, ob), f"plane {i}: filtered plane differs"
        assert int((a.params.view(-1, 7)[:, 0] >= 0).sum()) > 0
    # statistics only (no application records): the same counts
    for a, _, _, _ in bufs:
        a.count.fill_(-1)
    H.sao_planes(depth, [dict(q, out=None) for q in fused])
    torch.cuda.synchronize()
    for i, (a, _, b, _) in enumerate(bufs):
        assert torch.equal(a.count, b.count), f"plane {i}: statistics-only call differs"


# ---- round 3: the real rate-distortion decision of the parameters on the device (x265hip_sao_rdo) ----------------------------------
@pytest.mark.parametrize("depth,width,height,slice_type,qp,planes", [(8, 256, 192, 1, 27, 3), (8, 200, 150, 2, 22, 3), (8, 1920, 1080, 1, 30, 3), (10, 192, 136, 0, 30, 3),
                                                                     (12, 192, 136, 2, 14, 3), (8, 256, 128, 1, 32, 1), (8, 128, 4352, 1, 27, 3),
                                                                     (8, 3840, 2160, 1, 27, 3)])
def test_sao_rdo_matches_oracle(depth, width, height, slice_type, qp, planes):
    """x265hip_sao_rdo on the device's own statistics == the oracle's restatement of SAO::rdoSaoUnitCu (pinned against the real class,
    tests/test_oracle_classes_vs_reference.py) on the same statistics: type, band position, offsets and merge mode of every CTU and plane,
    and the count of CTUs left without SAO.  B / P / I context initialisation, 4:2:0 and luma-only, per-CTU lambdas, a 68-row picture
    (two wavefronts of rows, the 8K case) and a whole 4K picture."""
    import torch
    HT = importlib.import_module("x265-yuuki-asuna_amd.host_tables")
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
    S = importlib.import_module("x265-yuuki-asuna_amd.stages")
    A = H
    from test_oracle_classes_vs_reference import sao_rdo_case
    O = _oracle()
    dev = torch.device("cuda:0")
    tabs = HT.load()
    rng = np.random.default_rng([37, depth, width, qp])
    src, rec = sao_rdo_case(depth, width, height, 11, 2 + qp // 12)
    cur = P.DevicePicture(src[0], dev, src[1], src[2])
    dbl = P.DevicePicture(rec[0], dev, rec[1], rec[2])
    ctus_w, ctus_h = cur.w64 // 64, cur.h64 // 64
    nctu = ctus_w * ctus_h
    stages = [S.Sao(width, height, depth, dev)] + ([S.Sao(width // 2, height // 2, depth, dev, ctu=(32, 32), plane_offset=2) for _ in range(2)] if planes == 3 else [])
    stages[0].stats(cur, dbl.t, cur.stride, cur.org)
    for i in range(planes - 1):
        stages[1 + i].stats(None, dbl.c[i], cur.stride_c, cur.org_c, src_plane=cur.c[i])
    per_ctu = width <= 256                              # per-CTU QPs on the small cases, one QP on the large ones
    ctu_qp = np.clip(qp + rng.integers(-3, 4, size=nctu), 0, 51) if per_ctu else np.full(nctu, qp)
    lam = np.array([HT.sao_lambdas(tabs, int(q), csp400=planes == 1) for q in ctu_qp], dtype=np.int64)
    ctx_m, ctx_t = HT.sao_contexts(slice_type, qp)
    scratch = torch.zeros(A.sao_rdo_scratch_bytes(ctus_w, ctus_h), dtype=torch.uint8, device=dev)
    nos = torch.full((2,), -1, dtype=torch.int32, device=dev)
    for st in stages:
        st.params.fill_(0x5a5a5a5a)
    A.sao_rdo(depth, [st.count for st in stages], [st.offset_org for st in stages], ctus_w, ctus_h, lam[0], ctx_m, ctx_t, tabs["entropy_bits"],
              [st.params for st in stages], scratch, lambda_ctu=torch.from_numpy(lam).to(dev) if per_ctu else None, sao_flag=(1, 1 if planes == 3 else 0), num_no_sao=nos)
    torch.cuda.synchronize()
    eparams, enos = O.sao_rdo(depth, [st.count.cpu().numpy() for st in stages], [st.offset_org.cpu().numpy() for st in stages], ctus_w, ctus_h, lam, ctx_m, ctx_t,
                              tabs["entropy_bits"], sao_flag=(1, 1 if planes == 3 else 0))
    for pl in range(planes):
        got = stages[pl].params.cpu().numpy().reshape(nctu, 7)
        bad = np.argwhere((got != eparams[pl]).any(axis=1))[:5].reshape(-1).tolist()
        assert np.array_equal(got, eparams[pl]), f"plane {pl}: CTUs {bad}: device {got[bad].tolist()} oracle {eparams[pl][bad].tolist()}"
    gn = nos.cpu().numpy()
    assert int(gn[0]) == int(enos[0]) and (planes == 1 or int(gn[1]) == int(enos[1]))
    assert (eparams[0][:, 0] >= 0).any()
    if nctu >= 12:
        assert (eparams[0][:, 6] > 0).any()
