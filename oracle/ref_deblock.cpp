/* oracle/ref_deblock.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * C-ABI window onto the REAL reference deblocking filter (common/deblock.cpp): builds a Frame / FrameData / CUData picture made of
 * square 2Nx2N inter CUs of one size (one PU, one TU each: the picture x265hip_inter_recon produces) with the caller's motion
 * vectors and coded-block flags, and runs Deblock::deblockCTU over every CTU - all vertical edges, then all horizontal edges.
 * The tests use it to pin oracle/x265_oracle_pipeline4.c's x265oracle_deblock_bs_inter + x265oracle_deblock_luma.
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "frame.h"
#include "framedata.h"
#include "cudata.h"
#include "slice.h"
#include "deblock.h"
#include "x265.h"

#include <cstring>
#include <vector>

using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

extern "C" {

/* recPlane: ALLOCATION START of a padded luma plane (reference PicYuv geometry, width / height multiples of 64), filtered in place.
 * level 0..2 = 8x8 / 16x16 / 32x32 blocks; mv: int32 [numCtu * 85][2] = { cost, qx | qy << 16 } (the sub-pel stage's records, the
 * level's blocks in z-order at offsets 0 / 64 / 80); numSig: uint32 [numCtu][blocks per CTU].  Returns 0 on success. */
int x265ref_deblock420(void* recPlane, void* cbPlane, void* crPlane, int width, int height, int level, const int32_t* mv,
                       const uint32_t* numSig, const uint8_t* intra, int qp, int betaOffsetDiv2, int tcOffsetDiv2, int cbQpOffset, int crQpOffset);
int x265ref_deblock_b(void* recPlane, int width, int height, int level, int sliceB, const int32_t* mv0, const int32_t* mv1,
                      const int8_t* ref0, const int8_t* ref1, const uint32_t* numSig, const uint8_t* intra, int qp,
                      int betaOffsetDiv2, int tcOffsetDiv2);

int x265ref_deblock(void* recPlane, int width, int height, int level, const int32_t* mv, const uint32_t* numSig, int qp,
                    int betaOffsetDiv2, int tcOffsetDiv2)
{
    return x265ref_deblock420(recPlane, NULL, NULL, width, height, level, mv, numSig, NULL, qp, betaOffsetDiv2, tcOffsetDiv2, 0, 0);
}

/* The general form: cbPlane / crPlane = UNPADDED (width / 2) x (height / 2) chroma planes of a 4:2:0 picture (NULL: luma only),
 * filtered in place; intra: optional uint8 [numCtu][blocks per CTU], non-zero = the block is an intra CU (Bs 2 on its edges, the
 * only edges the chroma filter touches); cbQpOffset / crQpOffset = pps->chromaQpOffset[]. */
static int deblock_core(void* recPlane, void* cbPlane, void* crPlane, int width, int height, int level, const int32_t* mv,
                        const uint32_t* numSig, const uint8_t* intra, int qp, int betaOffsetDiv2, int tcOffsetDiv2, int cbQpOffset, int crQpOffset,
                        int sliceB, const int32_t* mv1, const int8_t* ref0, const int8_t* ref1);

int x265ref_deblock420(void* recPlane, void* cbPlane, void* crPlane, int width, int height, int level, const int32_t* mv,
                       const uint32_t* numSig, const uint8_t* intra, int qp, int betaOffsetDiv2, int tcOffsetDiv2, int cbQpOffset, int crQpOffset)
{
    return deblock_core(recPlane, cbPlane, crPlane, width, height, level, mv, numSig, intra, qp, betaOffsetDiv2, tcOffsetDiv2, cbQpOffset, crQpOffset,
                        0, NULL, NULL, NULL);
}

/* Pictures with several references / B pictures (luma): ref0 / ref1 = int8 [numCtu][blocks] reference PICTURE ids (0..3, -1 = list
 * unused; equal ids denote the same picture in either list), mv1 = list-1 records; sliceB selects B_SLICE. */
int x265ref_deblock_b(void* recPlane, int width, int height, int level, int sliceB, const int32_t* mv0, const int32_t* mv1,
                      const int8_t* ref0, const int8_t* ref1, const uint32_t* numSig, const uint8_t* intra, int qp,
                      int betaOffsetDiv2, int tcOffsetDiv2)
{
    return deblock_core(recPlane, NULL, NULL, width, height, level, mv0, numSig, intra, qp, betaOffsetDiv2, tcOffsetDiv2, 0, 0, sliceB, mv1, ref0, ref1);
}

static int deblock_core(void* recPlane, void* cbPlane, void* crPlane, int width, int height, int level, const int32_t* mv,
                        const uint32_t* numSig, const uint8_t* intra, int qp, int betaOffsetDiv2, int tcOffsetDiv2, int cbQpOffset, int crQpOffset,
                        int sliceB, const int32_t* mv1, const int8_t* ref0, const int8_t* ref1)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    if ((width | height) & 63) return -10;
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = cbPlane ? X265_CSP_I420 : X265_CSP_I400;
    param->maxCUSize = 64;
    param->minCUSize = 8;
    param->maxLog2CUSize = 6;
    param->unitSizeDepth = 4;
    param->num4x4Partitions = 256;
    param->bEnableLoopFilter = 1;
    param->bLossless = 0;
    SPS sps;
    memset((void*)&sps, 0, sizeof(sps));
    sps.numCuInWidth = width / 64;
    sps.numCuInHeight = height / 64;
    sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    sps.numPartInCUSize = 16;
    sps.numPartitions = 256;
    sps.picWidthInLumaSamples = width;
    sps.picHeightInLumaSamples = height;
    PPS pps;
    memset((void*)&pps, 0, sizeof(pps));
    pps.deblockingFilterBetaOffsetDiv2 = betaOffsetDiv2;
    pps.deblockingFilterTcOffsetDiv2 = tcOffsetDiv2;
    pps.bTransquantBypassEnabled = 0;
    pps.chromaQpOffset[0] = cbQpOffset;
    pps.chromaQpOffset[1] = crQpOffset;
    const int numCtu = sps.numCUsInFrame;

    PicYuv recon;
    recon.m_param = param;
    if (!recon.create(param, true) || !recon.createOffsets(sps)) return -1;
    const size_t planeBytes = sizeof(pixel) * recon.m_stride * (height + 2 * recon.m_lumaMarginY);
    pixel* planeStart = recon.m_picOrg[0] - recon.m_lumaMarginY * recon.m_stride - recon.m_lumaMarginX;
    memcpy(planeStart, recPlane, planeBytes);
    void* chroma[2] = { cbPlane, crPlane };
    const int cw = width / 2, ch = height / 2;
    if (cbPlane)
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < ch; y++)
                memcpy(recon.m_picOrg[1 + c] + (intptr_t)y * recon.m_strideC, (const pixel*)chroma[c] + (size_t)y * cw, sizeof(pixel) * cw);

    Frame frame, refFrames[4];
    frame.m_param = param;
    frame.m_reconPic = &recon;
    FrameData encData;
    Slice slice;
    slice.m_sps = &sps;
    slice.m_pps = &pps;
    slice.m_param = param;
    slice.m_sliceType = sliceB ? B_SLICE : P_SLICE;
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < 4; i++) slice.m_refFrameList[l][i] = &refFrames[i];       /* refIdx = picture id in both lists */
    encData.m_param = param;
    encData.m_slice = &slice;
    encData.m_reconPic = &recon;
    std::vector<CUData> ctus(numCtu);
    encData.m_picCTU = ctus.data();
    frame.m_encData = &encData;
    CUDataMemPool pool;
    if (!pool.create(0, param->internalCsp, numCtu, *param)) return -2;

    const int n = 8 << level, log2n = 3 + level, depth = 3 - level;
    const int npu = (64 / n) * (64 / n), partsPerBlock = (n / 4) * (n / 4);
    const int lbase = level == 0 ? 0 : (level == 1 ? 64 : 80);
    for (int a = 0; a < numCtu; a++)
    {
        const int row = a / sps.numCuInWidth;
        ctus[a].initialize(pool, 0, *param, a);
        ctus[a].initCTU(frame, a, qp, row == 0, row == (int)sps.numCuInHeight - 1, a == numCtu - 1);
        for (int p = 0; p < 256; p++)
        {
            const int z = p / partsPerBlock;
            const int32_t pk = mv[((size_t)a * 85 + lbase + z) * 2 + 1];
            ctus[a].m_predMode[p] = (intra && intra[(size_t)a * npu + z]) ? MODE_INTRA : MODE_INTER;
            ctus[a].m_cuDepth[p] = (uint8_t)depth;
            ctus[a].m_log2CUSize[p] = (uint8_t)log2n;
            ctus[a].m_partSize[p] = SIZE_2Nx2N;
            ctus[a].m_tuDepth[p] = 0;
            ctus[a].m_cbf[0][p] = numSig[(size_t)a * npu + z] ? 1 : 0;
            ctus[a].m_mv[0][p] = MV((int16_t)(pk & 0xffff), (int16_t)(pk >> 16));
            ctus[a].m_refIdx[0][p] = ref0 ? ref0[(size_t)a * npu + z] : 0;
            if (mv1)
            {
                const int32_t pk1 = mv1[((size_t)a * 85 + lbase + z) * 2 + 1];
                ctus[a].m_mv[1][p] = MV((int16_t)(pk1 & 0xffff), (int16_t)(pk1 >> 16));
            }
            ctus[a].m_refIdx[1][p] = ref1 ? ref1[(size_t)a * npu + z] : -1;
            ctus[a].m_qp[p] = (int8_t)qp;
        }
    }
    CUGeom geoms[CUGeom::MAX_GEOMS];
    CUData::calcCTUGeoms(64, 64, 64, 8, geoms);
    Deblock deblock;
    for (int a = 0; a < numCtu; a++) deblock.deblockCTU(&ctus[a], geoms[0], Deblock::EDGE_VER);
    for (int a = 0; a < numCtu; a++) deblock.deblockCTU(&ctus[a], geoms[0], Deblock::EDGE_HOR);
    memcpy(recPlane, planeStart, planeBytes);
    if (cbPlane)
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < ch; y++)
                memcpy((pixel*)chroma[c] + (size_t)y * cw, recon.m_picOrg[1 + c] + (intptr_t)y * recon.m_strideC, sizeof(pixel) * cw);

    frame.m_reconPic = NULL; frame.m_encData = NULL;
    encData.m_picCTU = NULL; encData.m_slice = NULL;
    pool.destroy();
    recon.destroy();
    x265_param_free(param);
    return 0;
}

} // extern "C"
