/* oracle/ref_sao.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * C-ABI window onto the REAL reference SAO class (encoder/sao.cpp): builds a Frame / FrameData / CUData skeleton around
 * caller-supplied source and deblocked luma planes and runs SAO::calcSaoStatsCTU (sao.cpp:735-917) for every CTU, then
 * SAO::generateLumaOffsets / applyPixelOffsets (sao.cpp:572-630, 274-570) CTU by CTU in raster order with the above-row
 * buffer filled the way FrameFilter::ParallelFilter::copySaoAboveRef does (framefilter.cpp:300-322: from the not yet
 * offset picture).  The tests use it to pin oracle/x265_oracle_pipeline4.c's SAO restatement.
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "frame.h"
#include "framedata.h"
#include "cudata.h"
#include "slice.h"
#include "sao.h"
#include "entropy.h"
#include "constants.h"
#include "x265.h"

#include <cstring>
#include <vector>

namespace X265_NS { uint8_t sbacInit(int qp, int initValue); }        /* entropy.cpp:1297-1308, not declared in a header */
using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

namespace {
struct SaoProbe : public SAO
{
    using SAO::m_count;
    using SAO::m_offsetOrg;
    using SAO::m_tmpU;
    using SAO::m_offset;
    using SAO::m_numNoSao;
};
std::vector<int32_t> g_lastInitialOffsets;      /* SAO::m_offset[0] after saoStatsInitialOffset(addr, 0), per CTU of the last x265ref_sao call */
}

extern "C" {

/* fencPlane / recPlane: ALLOCATION STARTS of padded luma planes with the reference's PicYuv geometry (as x265ref_lowres_intra);
 * recPlane holds the deblocked picture on entry and the sample-adaptive-offset picture on return.
 * params: int32 [numCtu][7] = { typeIdx (-1 none, 0..3 EO, 4 BO), bandPos, offset[4], mergeLeft (0 / 1: take the left CTU's
 * offsets through SAO_MERGE_LEFT instead of re-deriving them) }.
 * count / offsetOrg: int32 [numCtu][5][32] = SAO::m_count[0] / m_offsetOrg[0] after calcSaoStatsCTU(addr, 0). */
int x265ref_sao(const void* fencPlane, void* recPlane, int width, int height, const int32_t* params, int32_t* count, int32_t* offsetOrg)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->maxLog2CUSize = 6;
    param->unitSizeDepth = 4;
    param->num4x4Partitions = 256;
    param->bEnableSAO = 1;
    param->bSaoNonDeblocked = 0;
    param->bLimitSAO = 0;
    SPS sps;
    memset((void*)&sps, 0, sizeof(sps));
    sps.numCuInWidth = (width + 63) / 64;
    sps.numCuInHeight = (height + 63) / 64;
    sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    const int numCtu = sps.numCUsInFrame;
    const int h64 = sps.numCuInHeight * 64;

    Frame frame;
    frame.m_param = param;
    PicYuv fenc, recon;
    PicYuv* pics[2] = { &fenc, &recon };
    const void* srcs[2] = { fencPlane, recPlane };
    for (int i = 0; i < 2; i++)
    {
        pics[i]->m_param = param;
        if (!pics[i]->create(param, true) || !pics[i]->createOffsets(sps)) return -1;
        memcpy(pics[i]->m_picOrg[0] - pics[i]->m_lumaMarginY * pics[i]->m_stride - pics[i]->m_lumaMarginX, srcs[i],
               sizeof(pixel) * pics[i]->m_stride * (h64 + 2 * pics[i]->m_lumaMarginY));
    }
    frame.m_fencPic = &fenc;
    frame.m_reconPic = &recon;
    FrameData encData;
    Slice slice;
    slice.m_sps = &sps;
    slice.m_param = param;
    slice.m_sliceType = P_SLICE;
    encData.m_param = param;
    encData.m_slice = &slice;
    encData.m_reconPic = &recon;
    std::vector<CUData> ctus(numCtu);
    encData.m_picCTU = ctus.data();
    for (int a = 0; a < numCtu; a++)
    {
        const int row = a / sps.numCuInWidth, col = a % sps.numCuInWidth;
        ctus[a].m_encData = &encData;
        ctus[a].m_slice = &slice;
        ctus[a].m_cuAddr = a;
        ctus[a].m_cuPelX = col * 64;
        ctus[a].m_cuPelY = row * 64;
        ctus[a].m_bFirstRowInSlice = row == 0;
        ctus[a].m_bLastRowInSlice = row == (int)sps.numCuInHeight - 1;
    }
    frame.m_encData = &encData;

    SaoProbe sao;
    if (!sao.create(param, 1)) return -2;
    sao.m_frame = &frame;
    for (int a = 0; a < numCtu; a++)
    {
        sao.resetStats();
        sao.calcSaoStatsCTU(a, 0);
        memcpy(count + (size_t)a * 5 * 32, sao.m_count[0], sizeof(int32_t) * 5 * 32);
        memcpy(offsetOrg + (size_t)a * 5 * 32, sao.m_offsetOrg[0], sizeof(int32_t) * 5 * 32);
        sao.saoStatsInitialOffset(a, 0);                       /* sao.cpp:1378-1433 */
        g_lastInitialOffsets.resize((size_t)numCtu * 5 * 32);
        memcpy(g_lastInitialOffsets.data() + (size_t)a * 5 * 32, sao.m_offset[0], sizeof(int32_t) * 5 * 32);
    }

    /* apply: the above-row reference comes from the picture before any offset was applied */
    const intptr_t stride = recon.m_stride;
    std::vector<pixel> pristine(recon.m_picOrg[0], recon.m_picOrg[0] + stride * h64);
    std::vector<SaoCtuParam> cp(numCtu);
    for (int a = 0; a < numCtu; a++)
    {
        const int32_t* p = params + (size_t)a * 7;
        cp[a].reset();
        cp[a].typeIdx = p[0];
        cp[a].bandPos = p[1];
        for (int i = 0; i < 4; i++) cp[a].offset[i] = p[2 + i];
        cp[a].mergeMode = p[6] ? SAO_MERGE_LEFT : SAO_MERGE_NONE;
    }
    for (int row = 0; row < (int)sps.numCuInHeight; row++)
    {
        const pixel* above = pristine.data() + (row == 0 ? 0 : (intptr_t)(row * 64 - 1) * stride);
        memcpy(sao.m_tmpU[0], above, sizeof(pixel) * sps.numCuInWidth * 64);
        for (int col = 0; col < (int)sps.numCuInWidth; col++)
            sao.generateLumaOffsets(cp.data(), row, col);
    }
    memcpy(recPlane, recon.m_picOrg[0] - recon.m_lumaMarginY * recon.m_stride - recon.m_lumaMarginX,
           sizeof(pixel) * recon.m_stride * (h64 + 2 * recon.m_lumaMarginY));

    /* the skeleton borrows stack objects: detach them before the destructors run */
    frame.m_fencPic = NULL; frame.m_reconPic = NULL; frame.m_encData = NULL;
    encData.m_picCTU = NULL; encData.m_slice = NULL;
    sao.destroy(1);
    fenc.destroy(); recon.destroy();
    x265_param_free(param);
    return 0;
}

/* The chroma planes of a 4:2:0 picture through the same real class: calcSaoStatsCTU(addr, 1 / 2) and generateChromaOffsets.
 * fencC / recC: [2] pointers to UNPADDED (width / 2) x (height / 2) planes (Cb, Cr); recC is replaced by the offset planes.
 * params: [2] pointers to int32 [numCtu][7] as in x265ref_sao (the reference applies Cb's typeIdx to Cr as well, sao.cpp:723: pass
 * the same type for both, as the encoder's decision always does).  count / offsetOrg: [2] pointers to int32 [numCtu][5][32]. */
int x265ref_sao_chroma(const void* const* fencC, void* const* recC, int width, int height, const int32_t* const* params,
                       int32_t* const* count, int32_t* const* offsetOrg)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I420;
    param->maxCUSize = 64;
    param->maxLog2CUSize = 6;
    param->unitSizeDepth = 4;
    param->num4x4Partitions = 256;
    param->bEnableSAO = 1;
    param->bSaoNonDeblocked = 0;
    param->bLimitSAO = 0;
    SPS sps;
    memset((void*)&sps, 0, sizeof(sps));
    sps.numCuInWidth = (width + 63) / 64;
    sps.numCuInHeight = (height + 63) / 64;
    sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    const int numCtu = sps.numCUsInFrame;
    const int cw = width / 2, ch = height / 2;

    Frame frame;
    frame.m_param = param;
    PicYuv fenc, recon;
    PicYuv* pics[2] = { &fenc, &recon };
    for (int i = 0; i < 2; i++)
    {
        pics[i]->m_param = param;
        if (!pics[i]->create(param, true) || !pics[i]->createOffsets(sps)) return -1;
        for (int c = 0; c < 2; c++)
        {
            const pixel* src = (const pixel*)(i ? recC[c] : fencC[c]);
            for (int y = 0; y < ch; y++)
                memcpy(pics[i]->m_picOrg[1 + c] + (intptr_t)y * pics[i]->m_strideC, src + (size_t)y * cw, sizeof(pixel) * cw);
        }
    }
    frame.m_fencPic = &fenc;
    frame.m_reconPic = &recon;
    FrameData encData;
    Slice slice;
    slice.m_sps = &sps;
    slice.m_param = param;
    slice.m_sliceType = P_SLICE;
    encData.m_param = param;
    encData.m_slice = &slice;
    encData.m_reconPic = &recon;
    std::vector<CUData> ctus(numCtu);
    encData.m_picCTU = ctus.data();
    for (int a = 0; a < numCtu; a++)
    {
        const int row = a / sps.numCuInWidth, col = a % sps.numCuInWidth;
        ctus[a].m_encData = &encData;
        ctus[a].m_slice = &slice;
        ctus[a].m_cuAddr = a;
        ctus[a].m_cuPelX = col * 64;
        ctus[a].m_cuPelY = row * 64;
        ctus[a].m_bFirstRowInSlice = row == 0;
        ctus[a].m_bLastRowInSlice = row == (int)sps.numCuInHeight - 1;
    }
    frame.m_encData = &encData;

    SaoProbe sao;
    if (!sao.create(param, 1)) return -2;
    sao.m_frame = &frame;
    for (int a = 0; a < numCtu; a++)
        for (int c = 0; c < 2; c++)
        {
            sao.resetStats();
            sao.calcSaoStatsCTU(a, 1 + c);
            memcpy(count[c] + (size_t)a * 5 * 32, sao.m_count[1 + c], sizeof(int32_t) * 5 * 32);
            memcpy(offsetOrg[c] + (size_t)a * 5 * 32, sao.m_offsetOrg[1 + c], sizeof(int32_t) * 5 * 32);
        }

    const intptr_t strideC = recon.m_strideC;
    const int h32 = sps.numCuInHeight * 32;
    std::vector<pixel> pristine[2];
    std::vector<SaoCtuParam> cp[3];
    for (int c = 0; c < 3; c++) cp[c].resize(numCtu);
    for (int c = 0; c < 2; c++)
    {
        pristine[c].assign(recon.m_picOrg[1 + c], recon.m_picOrg[1 + c] + strideC * h32);
        for (int a = 0; a < numCtu; a++)
        {
            const int32_t* p = params[c] + (size_t)a * 7;
            cp[1 + c][a].reset();
            cp[1 + c][a].typeIdx = p[0];
            cp[1 + c][a].bandPos = p[1];
            for (int i = 0; i < 4; i++) cp[1 + c][a].offset[i] = p[2 + i];
            cp[1 + c][a].mergeMode = p[6] ? SAO_MERGE_LEFT : SAO_MERGE_NONE;
        }
    }
    SaoCtuParam* cps[3] = { cp[0].data(), cp[1].data(), cp[2].data() };
    for (int row = 0; row < (int)sps.numCuInHeight; row++)
    {
        for (int c = 0; c < 2; c++)
        {
            const pixel* above = pristine[c].data() + (row == 0 ? 0 : (intptr_t)(row * 32 - 1) * strideC);
            memcpy(sao.m_tmpU[1 + c], above, sizeof(pixel) * sps.numCuInWidth * 32);
        }
        for (int col = 0; col < (int)sps.numCuInWidth; col++)
            sao.generateChromaOffsets(cps, row, col);
    }
    for (int c = 0; c < 2; c++)
        for (int y = 0; y < ch; y++)
            memcpy((pixel*)recC[c] + (size_t)y * cw, recon.m_picOrg[1 + c] + (intptr_t)y * strideC, sizeof(pixel) * cw);

    frame.m_fencPic = NULL; frame.m_reconPic = NULL; frame.m_encData = NULL;
    encData.m_picCTU = NULL; encData.m_slice = NULL;
    sao.destroy(1);
    fenc.destroy(); recon.destroy();
    x265_param_free(param);
    return 0;
}

/* The REAL rate-distortion decision of the SAO parameters: SAO::rdoSaoUnitCu (sao.cpp:1225-1376) over a whole 4:2:0 (or 4:0:0) picture,
 * driven the way FrameFilter does - one SAO object per CTU row, every row's entropy contexts starting from the slice's initial state
 * (sao.cpp:245-247; framefilter.cpp:239), CTUs of a row left to right, rows top to bottom so that the merge-up candidate sees the row
 * above.  The class computes its own statistics (calcSaoStatsCTU inside rdoSaoUnitCu) from the planes handed in.
 *   fenc / rec : [3] ALLOCATION STARTS of padded planes with the PicYuv geometry of (width, height); chroma may be NULL with csp400
 *   sliceType  : 0 B, 1 P, 2 I (slice.h);  sliceQp initialises the contexts (entropy.cpp:1297-1308);  ctuQp: int [numCtu] = cu->m_qp[0]
 *   params     : [3] int32 [numCtu][7] out = { typeIdx, bandPos, offset[4], mergeMode (0 none, 1 left, 2 up) }
 *   info       : int64 out [8] = { lambda luma, lambda chroma of CTU 0, initial sao_merge ctx state, initial sao_type ctx state,
 *                m_numNoSao[0] summed over the rows, m_numNoSao[1], 0, 0 } */
int x265ref_sao_rdo(const void* const* fencPlanes, const void* const* recPlanes, int width, int height, int csp400, int sliceType, int sliceQp,
                    const int32_t* ctuQp, int cbQpOffset, int32_t* const* params, int64_t* info)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = csp400 ? X265_CSP_I400 : X265_CSP_I420;
    param->maxCUSize = 64;
    param->maxLog2CUSize = 6;
    param->unitSizeDepth = 4;
    param->num4x4Partitions = 256;
    param->bEnableSAO = 1;
    param->bSaoNonDeblocked = 0;
    param->bLimitSAO = 0;
    param->frameNumThreads = 2;               /* no automatic turn-off from earlier pictures' statistics (sao.cpp:263-270) */
    SPS sps;
    memset((void*)&sps, 0, sizeof(sps));
    sps.numCuInWidth = (width + 63) / 64;
    sps.numCuInHeight = (height + 63) / 64;
    sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    PPS pps;
    memset((void*)&pps, 0, sizeof(pps));
    pps.chromaQpOffset[0] = cbQpOffset;
    const int numCtu = sps.numCUsInFrame;
    const int h64 = sps.numCuInHeight * 64;
    const int planes = csp400 ? 1 : 3;

    Frame frame;
    frame.m_param = param;
    PicYuv fenc, recon;
    PicYuv* pics[2] = { &fenc, &recon };
    for (int i = 0; i < 2; i++)
    {
        pics[i]->m_param = param;
        if (!pics[i]->create(param, true) || !pics[i]->createOffsets(sps)) return -1;
        const void* const* srcs = i ? recPlanes : fencPlanes;
        memcpy(pics[i]->m_picOrg[0] - pics[i]->m_lumaMarginY * pics[i]->m_stride - pics[i]->m_lumaMarginX, srcs[0],
               sizeof(pixel) * pics[i]->m_stride * (h64 + 2 * pics[i]->m_lumaMarginY));
        for (int c = 1; c < planes; c++)
            memcpy(pics[i]->m_picOrg[c] - pics[i]->m_chromaMarginY * pics[i]->m_strideC - pics[i]->m_chromaMarginX, srcs[c],
                   sizeof(pixel) * pics[i]->m_strideC * (h64 / 2 + 2 * pics[i]->m_chromaMarginY));
    }
    frame.m_fencPic = &fenc;
    frame.m_reconPic = &recon;
    FrameData encData;
    Slice slice;
    slice.m_sps = &sps;
    slice.m_pps = &pps;
    slice.m_param = param;
    slice.m_sliceType = (SliceType)sliceType;
    slice.m_sliceQp = sliceQp;
    slice.m_chromaQpOffset[0] = slice.m_chromaQpOffset[1] = 0;
    encData.m_param = param;
    encData.m_slice = &slice;
    encData.m_reconPic = &recon;
    encData.m_saoParam = NULL;
    std::vector<CUData> ctus(numCtu);
    std::vector<int8_t> qps(numCtu);
    std::vector<uint8_t> skip(numCtu, 0);
    encData.m_picCTU = ctus.data();
    for (int a = 0; a < numCtu; a++)
    {
        const int row = a / sps.numCuInWidth, col = a % sps.numCuInWidth;
        ctus[a].m_encData = &encData;
        ctus[a].m_slice = &slice;
        ctus[a].m_cuAddr = a;
        ctus[a].m_cuPelX = col * 64;
        ctus[a].m_cuPelY = row * 64;
        ctus[a].m_bFirstRowInSlice = row == 0;
        ctus[a].m_bLastRowInSlice = row == (int)sps.numCuInHeight - 1;
        qps[a] = (int8_t)ctuQp[a];
        ctus[a].m_qp = &qps[a];                    /* rdoSaoUnitCu reads cu->m_qp[0] (sao.cpp:1229) */
        ctus[a].m_predMode = &skip[a];             /* isSkipped(0) is only consulted for B slices; never true here */
    }
    frame.m_encData = &encData;
    frame.m_lowres.sliceType = sliceType == 2 ? X265_TYPE_I : sliceType == 1 ? X265_TYPE_P : X265_TYPE_BREF;

    Entropy initState;
    initState.resetEntropy(slice);
    initState.zeroFract();
    SaoProbe root;
    if (!root.create(param, 1)) return -2;
    root.startSlice(&frame, initState);           /* allocates encData.m_saoParam, sets bSaoFlag */
    SAOParam* saoParam = encData.m_saoParam;
    saoParam->bSaoFlag[0] = true;
    saoParam->bSaoFlag[1] = !csp400;
    int noSao[2] = { 0, 0 };
    for (int row = 0; row < (int)sps.numCuInHeight; row++)
    {
        SaoProbe sao;
        if (!sao.create(param, 0)) return -3;
        sao.createFromRootNode(&root);
        sao.startSlice(&frame, initState);
        saoParam->bSaoFlag[0] = true; saoParam->bSaoFlag[1] = !csp400;
        for (int col = 0; col < (int)sps.numCuInWidth; col++)
            sao.rdoSaoUnitCu(saoParam, row * sps.numCuInWidth, col, row * sps.numCuInWidth + col);
        noSao[0] += sao.m_numNoSao[0]; noSao[1] += sao.m_numNoSao[1];
        sao.destroy(0);
    }
    for (int pl = 0; pl < planes; pl++)
        for (int a = 0; a < numCtu; a++)
        {
            const SaoCtuParam& q = saoParam->ctuParam[pl][a];
            int32_t* o = params[pl] + (size_t)a * 7;
            o[0] = q.typeIdx; o[1] = q.bandPos;
            for (int i = 0; i < 4; i++) o[2 + i] = q.offset[i];
            o[6] = q.mergeMode == SAO_MERGE_LEFT ? 1 : q.mergeMode == SAO_MERGE_UP ? 2 : 0;
        }
    {
        const int qp = ctuQp[0];
        int qpCb = qp + cbQpOffset;
        qpCb = csp400 ? x265_clip3(param->rc.qpMin, param->rc.qpMax, qpCb)
                      : x265_clip3(param->rc.qpMin, param->rc.qpMax, (int)g_chromaScale[x265_clip3(QP_MIN, QP_MAX_MAX, qpCb)]);
        info[0] = (int64_t)floor(256.0 * x265_lambda2_tab[qp]);
        info[1] = (int64_t)floor(256.0 * x265_lambda2_tab[qpCb]);
        info[2] = sbacInit(sliceQp, sliceType == 0 ? 153 : 153);      /* INIT_SAO_MERGE_FLAG: 153 for every slice type (entropy.cpp:196-201) */
        static const int typeInit[3] = { 160, 185, 200 };              /* INIT_SAO_TYPE_IDX by slice type B, P, I (entropy.cpp:203-208) */
        info[3] = sbacInit(sliceQp, typeInit[sliceType]);
        info[4] = noSao[0]; info[5] = noSao[1]; info[6] = info[7] = 0;
    }
    delete saoParam;                          /* ~SAOParam frees the per-plane arrays (sao.h) */
    encData.m_saoParam = NULL;
    frame.m_fencPic = NULL; frame.m_reconPic = NULL; frame.m_encData = NULL;
    encData.m_picCTU = NULL; encData.m_slice = NULL;
    for (int a = 0; a < numCtu; a++) { ctus[a].m_qp = NULL; ctus[a].m_predMode = NULL; }
    root.destroy(1);
    fenc.destroy(); recon.destroy();
    x265_param_free(param);
    return 0;
}

/* the host's per-state CABAC bit costs (g_entropyBits, entropy.cpp:2611) - the table a host hands to x265hip_sao_rdo */
const uint32_t* x265ref_entropy_bits_table(void) { return g_entropyBits; }

} // extern "C"

extern "C" int x265ref_sao_last_initial_offsets(int32_t* out, int numCtu)
{
    if ((size_t)numCtu * 160 != g_lastInitialOffsets.size()) return -1;
    memcpy(out, g_lastInitialOffsets.data(), sizeof(int32_t) * g_lastInitialOffsets.size());
    return 0;
}
