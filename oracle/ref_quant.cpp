/* oracle/ref_quant.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * C-ABI window onto the REAL reference transform-coding round trip: Quant::transformNxN (common/quant.cpp:397-480, the non-RDOQ
 * branch with the flat scaling list, sign hiding off) followed by Quant::invtransformNxN (:543-605) for a list of residual blocks.
 * The tests use it to pin the residual half of oracle/x265_oracle_pipeline2.c's inter / intra TU stages.
 */
#include "common.h"
#include "primitives.h"
#include "frame.h"
#include "framedata.h"
#include "cudata.h"
#include "slice.h"
#include "quant.h"
#include "scalinglist.h"
#include "entropy.h"
#include "x265.h"

#include <cstring>
#include <vector>

using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

namespace { struct QuantProbe : public Quant { using Quant::setChromaQP; }; }     /* setChromaQP is protected (quant.h:153) */

extern "C" {

/* resi: int16 [njobs][n*n] residual blocks (stride n).  qpScaled: the QP the quantiser works with (qp + QP_BD_OFFSET, what the
 * device stages take).  intraCU: the CU's prediction mode (4x4 intra luma uses DST-VII); intraSlice: I slice (rounding 171 vs 85).
 * Outputs: levels int16 [njobs][n*n], numSig uint32 [njobs], resiOut int16 [njobs][n*n] (zero when numSig == 0, as the callers
 * skip the inverse transform then).  Returns 0 on success. */
struct TuExtras          /* scaling lists / denoiser of the extended entry; all optional */
{
    int useScalingList;              /* HEVC default lists (ScalingList::setDefaultScalingList), m_bEnabled = true */
    const uint16_t* nrOffset;        /* denoiser offsets for this TU category, n * n */
    int32_t* quantCoefOut;           /* out: m_quantCoef[size][list][rem], n * n */
    int32_t* dequantCoefOut;         /* out: m_dequantCoef[size][list][rem], n * n */
    uint32_t* nrSumOut;              /* out: the category's residual sums after the jobs, n * n */
};
static int tu_roundtrip_core(const int16_t* resi, int n, int qpScaled, int intraCU, int intraSlice, int njobs,
                             int16_t* levels, uint32_t* numSig, int16_t* resiOut, TextType ttype, int signHide = 0, int intraDir = 1,
                             const TuExtras* ex = NULL);

/* + scaling lists (Quant::transformNxN :463 / invtransformNxN :562-567 with dequant_scaling) and the denoiser (:444-451) */
int x265ref_tu_roundtrip_ex2(const int16_t* resi, int n, int qpScaled, int intraCU, int intraSlice, int signHide, int intraDir, int chroma,
                             int useScalingList, const uint16_t* nrOffset, int njobs, int16_t* levels, uint32_t* numSig, int16_t* resiOut,
                             int32_t* quantCoefOut, int32_t* dequantCoefOut, uint32_t* nrSumOut)
{
    TuExtras ex = { useScalingList, nrOffset, quantCoefOut, dequantCoefOut, nrSumOut };
    return tu_roundtrip_core(resi, n, qpScaled, intraCU, intraSlice, njobs, levels, numSig, resiOut, chroma ? TEXT_CHROMA_U : TEXT_LUMA, signHide, intraDir, &ex);
}

/* the x265 default: pps.bSignHideEnabled = 1 (Quant::signBitHidingHDQ after the quantiser, quant.cpp:471-476).  intraDir is the
 * intra direction the TU's scan order depends on (CUData::getTUEntropyCodingParameters, cudata.cpp:2067-2089); chroma != 0 runs
 * TEXT_CHROMA_U blocks of a 4:2:0 picture (mode-dependent scans for the 4x4 TU only). */
int x265ref_tu_roundtrip_ex(const int16_t* resi, int n, int qpScaled, int intraCU, int intraSlice, int signHide, int intraDir, int chroma,
                            int njobs, int16_t* levels, uint32_t* numSig, int16_t* resiOut)
{
    return tu_roundtrip_core(resi, n, qpScaled, intraCU, intraSlice, njobs, levels, numSig, resiOut, chroma ? TEXT_CHROMA_U : TEXT_LUMA, signHide, intraDir);
}

int x265ref_tu_roundtrip(const int16_t* resi, int n, int qpScaled, int intraCU, int intraSlice, int njobs,
                         int16_t* levels, uint32_t* numSig, int16_t* resiOut)
{
    return tu_roundtrip_core(resi, n, qpScaled, intraCU, intraSlice, njobs, levels, numSig, resiOut, TEXT_LUMA);
}

/* the same with TEXT_CHROMA_U blocks (no DST for the 4x4 intra TU); qpScaled is the chroma QP + QP_BD_OFFSET the quantiser is
 * to work with: Quant::setChromaQP stores it per text type, here the CU's QP is simply set to it */
int x265ref_tu_roundtrip_chroma(const int16_t* resi, int n, int qpScaled, int intraCU, int intraSlice, int njobs,
                                int16_t* levels, uint32_t* numSig, int16_t* resiOut)
{
    return tu_roundtrip_core(resi, n, qpScaled, intraCU, intraSlice, njobs, levels, numSig, resiOut, TEXT_CHROMA_U);
}

static int tu_roundtrip_core(const int16_t* resi, int n, int qpScaled, int intraCU, int intraSlice, int njobs,
                             int16_t* levels, uint32_t* numSig, int16_t* resiOut, TextType ttype, int signHide, int intraDir, const TuExtras* ex)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    const int log2n = n == 4 ? 2 : (n == 8 ? 3 : (n == 16 ? 4 : (n == 32 ? 5 : 0)));
    if (!log2n) return -10;
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = 64;
    param->sourceHeight = 64;
    param->internalCsp = ((signHide || ex) && ttype != TEXT_LUMA) ? X265_CSP_I420 : X265_CSP_I400;   /* the chroma scan rule reads m_hChromaShift */
    param->frameNumThreads = 1;                     /* Quant::init allocates one NoiseReduction per frame encoder */
    param->maxCUSize = 64;
    param->minCUSize = 8;
    param->maxLog2CUSize = 6;
    param->unitSizeDepth = 4;
    param->num4x4Partitions = 256;
    param->rdoqLevel = 0;
    param->bLossless = 0;
    SPS sps;
    memset((void*)&sps, 0, sizeof(sps));
    sps.numCuInWidth = sps.numCuInHeight = sps.numCUsInFrame = 1;
    sps.numPartInCUSize = 16;
    sps.numPartitions = 256;
    sps.quadtreeTULog2MaxSize = 5;
    PPS pps;
    memset((void*)&pps, 0, sizeof(pps));
    pps.bSignHideEnabled = signHide != 0;
    Frame frame;
    frame.m_param = param;
    FrameData encData;
    Slice slice;
    slice.m_sps = &sps;
    slice.m_pps = &pps;
    slice.m_param = param;
    slice.m_sliceType = intraSlice ? I_SLICE : P_SLICE;
    encData.m_param = param;
    encData.m_frameEncoderID = 0;
    encData.m_slice = &slice;
    CUData ctu;
    encData.m_picCTU = &ctu;
    frame.m_encData = &encData;
    CUDataMemPool pool;
    if (!pool.create(0, param->internalCsp, 1, *param)) return -2;
    ctu.initialize(pool, 0, *param, 0);
    ctu.initCTU(frame, 0, qpScaled - QP_BD_OFFSET, 1, 1, 1);
    for (int p = 0; p < 256; p++)
    {
        ctu.m_predMode[p] = intraCU ? MODE_INTRA : MODE_INTER;
        ctu.m_lumaIntraDir[p] = (uint8_t)intraDir;
        if (param->internalCsp != X265_CSP_I400) ctu.m_chromaIntraDir[p] = (uint8_t)intraDir;
    }

    ScalingList scalingList;
    if (!scalingList.init()) return -3;
    scalingList.m_bEnabled = false;
    if (ex && ex->useScalingList)
    {
        scalingList.setDefaultScalingList();
        scalingList.m_bEnabled = true;
    }
    scalingList.setupQuantMatrices(param->internalCsp);
    Entropy entropy;
    QuantProbe quant;
    if (!quant.init(0.0, scalingList, entropy)) return -4;
    if (ex && ex->nrOffset && !quant.allocNoiseReduction(*param)) return -5;
    quant.setQPforQuant(ctu, qpScaled - QP_BD_OFFSET);
    if (ttype != TEXT_LUMA)          /* the I400 CTU above leaves the chroma QPs unset; chFmt 4:4:4 = no mapping table: the caller passes the mapped QP */
        quant.setChromaQP(qpScaled - QP_BD_OFFSET, ttype, X265_CSP_I444);

    const int sizeIdx = log2n - 2, listType = (intraCU ? 0 : 3) + (int)ttype, rem = qpScaled % 6;
    const int cat = sizeIdx + 4 * (ttype != TEXT_LUMA) + 8 * !intraCU;              /* quant.cpp:447 */
    if (ex)
    {
        if (ex->quantCoefOut) memcpy(ex->quantCoefOut, scalingList.m_quantCoef[sizeIdx][listType][rem], sizeof(int32_t) * n * n);
        if (ex->dequantCoefOut) memcpy(ex->dequantCoefOut, scalingList.m_dequantCoef[sizeIdx][listType][rem], sizeof(int32_t) * n * n);
        if (ex->nrOffset && quant.m_nr)
        {
            NoiseReduction* nr = quant.m_nr;
            nr->offset = nr->nrOffsetDenoise; nr->residualSum = nr->nrResidualSum; nr->count = nr->nrCount;
            memcpy(nr->nrOffsetDenoise[cat], ex->nrOffset, sizeof(uint16_t) * n * n);
        }
    }
    std::vector<pixel> fencDummy(n * n, 0);
    for (int j = 0; j < njobs; j++)
    {
        int16_t* lv = levels + (size_t)j * n * n;
        int16_t* out = resiOut + (size_t)j * n * n;
        const uint32_t ns = quant.transformNxN(ctu, fencDummy.data(), n, resi + (size_t)j * n * n, n, lv, log2n, ttype, 0, false);
        numSig[j] = ns;
        memset(out, 0, sizeof(int16_t) * n * n);
        if (ns) quant.invtransformNxN(ctu, out, n, lv, log2n, ttype, !!intraCU, false, ns);
    }
    if (ex && ex->nrSumOut && ex->nrOffset && quant.m_nr) memcpy(ex->nrSumOut, quant.m_nr->nrResidualSum[cat], sizeof(uint32_t) * n * n);
    frame.m_encData = NULL;
    encData.m_picCTU = NULL; encData.m_slice = NULL;
    pool.destroy();
    x265_param_free(param);
    return 0;
}

} // extern "C"
