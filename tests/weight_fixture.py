"""tests/golden/weight_analyse_d{8,10}.npz -> the (cur, refs, ...) arguments the oracle's and the library's weightAnalyse entries take, with the weights the
REFERENCE's own function chose for the same slice (tools/gen_weight_golden.py made them from real encodes of fading clips)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases(depth):
    z = np.load(os.path.join(GOLDEN, f"weight_analyse_d{depth}.npz"))
    out = []
    for i in range(int(z["count"][0])):
        g = lambda k: z[f"s{i}_{k}"]
        d, lstride, lw, lh, mx, my, cstride, cmx, cmy, pw, ph, nlists = [int(v) for v in g("geo")]
        assert d == depth
        lorg, corg = my * lstride + mx, cmy * cstride + cmx
        cur = dict(lowres=(g("cur_lowres"), lorg), lowres_stride=lstride, lowres_width=lw, lowres_lines=lh, cb=(g("cur_cb"), corg), cr=(g("cur_cr"), corg), stride_c=cstride,
                   wp_ssd=[int(v) for v in g("cur_wp_ssd")], wp_sum=[int(v) for v in g("cur_wp_sum")])
        refs, expected = [], np.zeros((2, 3, 4), np.int32)
        for l in range(nlists):
            mvs = g(f"ref{l}_mvs")
            refs.append(dict(lowres=[(g(f"ref{l}_lowres{k}"), lorg) for k in range(4)], cb=(g(f"ref{l}_cb"), corg), cr=(g(f"ref{l}_cr"), corg),
                             mvs=mvs.reshape(-1, 2) if mvs.size else None, wp_ssd=[int(v) for v in g(f"ref{l}_wp_ssd")], wp_sum=[int(v) for v in g(f"ref{l}_wp_sum")]))
            expected[l] = g(f"ref{l}_expected").reshape(3, 4)
        out.append(dict(cur=cur, refs=refs, pic=(pw, ph), intra=g("intra_cost"), lowres_margin=(mx, my), chroma_margin=(cmx, cmy), nlists=nlists, expected=expected))
    return out


def aq_cases(depth):
    """tests/golden/aq_frame_d{8,10}.npz: the source picture calcAdaptiveQuantFrame was handed in a real encode and the Lowres arrays the REFERENCE's function filled."""
    z = np.load(os.path.join(GOLDEN, f"aq_frame_d{depth}.npz"))
    out = []
    for i in range(int(z["count"][0])):
        g = lambda k: z[f"s{i}_{k}"]
        d, stride, mx, my, cstride, cmx, cmy, w, h, qg, mode, weightp = [int(v) for v in g("geo")]
        assert d == depth
        out.append(dict(y=g("y"), stride=stride, org=my * stride + mx, cb=g("cb"), cr=g("cr"), stride_c=cstride, org_c=cmy * cstride + cmx, width=w, height=h, qg=qg, mode=mode,
                        strength=float(g("strength")[0]), weightp=bool(weightp), grid=tuple(int(v) for v in g("grid")),
                        qp_aq_offset=g("qp_aq_offset"), qp_cutree_offset=g("qp_cutree_offset"), inv_qscale=g("inv_qscale"),
                        inv_qscale_8x8=z[f"s{i}_inv_qscale_8x8"] if f"s{i}_inv_qscale_8x8" in z.files else None, wp_sum=g("wp_sum"), wp_ssd=g("wp_ssd")))
    return out
