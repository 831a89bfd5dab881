/* oracle/ref_predict.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * C-ABI window onto the REAL reference motion compensation (common/predict.cpp, Predict::motionCompensation): builds a Frame /
 * FrameData / Slice / CUData picture of square 2Nx2N inter CUs of one size (the blocks of the fused TU stages) with the caller's
 * motion vectors, per-block prediction direction and explicit weight tables, runs motionCompensation for every CU - luma and 4:2:0
 * chroma - and hands the predicted picture back.  The tests use it to pin the PREDICTION half of oracle/x265_oracle_pipeline2.c's
 * inter stages (uni- and bi-directional, weighted and not, luma and chroma) against the real class.
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "frame.h"
#include "framedata.h"
#include "cudata.h"
#include "slice.h"
#include "predict.h"
#include "yuv.h"
#include "x265.h"

#include <cstring>
#include <vector>

using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

extern "C" {

/* ref0 / ref1: ALLOCATION STARTS of padded luma planes (reference PicYuv geometry, width / height multiples of 64) + UNPADDED
 * (width / 2) x (height / 2) chroma planes [Cb, Cr] per list (their padding is produced here by edge replication, like the encoder's
 * extendPicBorder).  level 0..2 = 8x8 / 16x16 / 32x32 CUs; mv0 / mv1: int32 [numCtu * 85][2] records of the sub-pel stage; dir: uint8
 * [numCtu][blocks]: 1 = list 0, 2 = list 1, 3 = both (NULL: all 1).  sliceB 0: P slice (weights used when useWeightPred); 1: B slice
 * (weights used when useWeightedBiPred).  weights: int32 [2 lists][3 planes][4] = { wtPresent, inputWeight, inputOffset,
 * log2WeightDenom }.  Outputs: predY = unpadded width x height, predCb / predCr = unpadded (width / 2) x (height / 2). */
int x265ref_motion_compensation(const void* ref0, const void* ref0Cb, const void* ref0Cr, const void* ref1, const void* ref1Cb, const void* ref1Cr,
                                int width, int height, int level, const int32_t* mv0, const int32_t* mv1, const uint8_t* dir,
                                int sliceB, int useWeightPred, int useWeightedBiPred, const int32_t* weights,
                                void* predY, void* predCb, void* predCr)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    if ((width | height) & 63) return -10;
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I420;
    param->maxCUSize = 64;
    param->minCUSize = 8;
    param->maxLog2CUSize = 6;
    param->unitSizeDepth = 4;
    param->num4x4Partitions = 256;
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
    pps.bUseWeightPred = useWeightPred != 0;
    pps.bUseWeightedBiPred = useWeightedBiPred != 0;
    const int numCtu = sps.numCUsInFrame;

    PicYuv refs[2], recon;
    const void* luma[2] = { ref0, ref1 };
    const void* chroma[2][2] = { { ref0Cb, ref0Cr }, { ref1Cb, ref1Cr } };
    const int cw = width / 2, ch = height / 2;
    recon.m_param = param;
    if (!recon.create(param, true) || !recon.createOffsets(sps)) return -1;
    for (int l = 0; l < 2; l++)
    {
        refs[l].m_param = param;
        if (!refs[l].create(param, true) || !refs[l].createOffsets(sps)) return -1;
        PicYuv& r = refs[l];
        memcpy(r.m_picOrg[0] - r.m_lumaMarginY * r.m_stride - r.m_lumaMarginX, luma[l], sizeof(pixel) * r.m_stride * (height + 2 * r.m_lumaMarginY));
        for (int c = 0; c < 2; c++)
        {
            pixel* org = r.m_picOrg[1 + c];
            const int mx = r.m_chromaMarginX, my = r.m_chromaMarginY;
            for (int y = -my; y < ch + my; y++)
                for (int x = -mx; x < cw + mx; x++)
                {
                    const int sy = y < 0 ? 0 : (y >= ch ? ch - 1 : y), sx = x < 0 ? 0 : (x >= cw ? cw - 1 : x);
                    org[(intptr_t)y * r.m_strideC + x] = ((const pixel*)chroma[l][c])[(size_t)sy * cw + sx];
                }
        }
    }

    Frame frame;
    frame.m_param = param;
    frame.m_reconPic = &recon;
    FrameData encData;
    Slice slice;
    slice.m_sps = &sps;
    slice.m_pps = &pps;
    slice.m_param = param;
    slice.m_sliceType = sliceB ? B_SLICE : P_SLICE;
    slice.m_numRefIdx[0] = 1;
    slice.m_numRefIdx[1] = sliceB ? 1 : 0;
    for (int l = 0; l < 2; l++)
    {
        slice.m_refReconPicList[l][0] = &refs[l];
        for (int p = 0; p < 3; p++)
        {
            WeightParam& w = slice.m_weightPredTable[l][0][p];
            const int32_t* v = weights + (l * 3 + p) * 4;
            w.wtPresent = v[0]; w.inputWeight = v[1]; w.inputOffset = v[2]; w.log2WeightDenom = (uint32_t)v[3];
        }
    }
    encData.m_param = param;
    encData.m_slice = &slice;
    encData.m_reconPic = &recon;
    std::vector<CUData> ctus(numCtu);
    encData.m_picCTU = ctus.data();
    frame.m_encData = &encData;
    const int n = 8 << level, log2n = 3 + level, depth = 3 - level;
    const int npu = (64 / n) * (64 / n), partsPerBlock = (n / 4) * (n / 4);
    const int lbase = level == 0 ? 0 : (level == 1 ? 64 : 80);
    CUDataMemPool pool, subPool;
    if (!pool.create(0, param->internalCsp, numCtu, *param) || !subPool.create(depth, param->internalCsp, 1, *param)) return -2;
    CUGeom geoms[CUGeom::MAX_GEOMS];
    CUData::calcCTUGeoms(64, 64, 64, 8, geoms);
    /* geoms of one depth, in z-order: walk the quad tree */
    std::vector<int> leaves;
    {
        std::vector<int> stack(1, 0);
        while (!stack.empty())
        {
            const int g = stack.back(); stack.pop_back();
            if ((int)geoms[g].depth == depth) { leaves.push_back(g); continue; }
            for (int k = 3; k >= 0; k--) stack.push_back(geoms[g].childOffset + g + k);
        }
    }
    if ((int)leaves.size() != npu) return -3;
    Predict pred;
    if (!pred.allocBuffers(param->internalCsp)) return -4;
    Yuv predYuv;
    if (!predYuv.create(n, param->internalCsp)) return -5;
    CUData cu;
    cu.initialize(subPool, depth, *param, 0);
    int rc = 0;
    for (int a = 0; a < numCtu; a++)
    {
        const int row = a / sps.numCuInWidth;
        ctus[a].initialize(pool, 0, *param, a);
        ctus[a].initCTU(frame, a, 30, row == 0, row == (int)sps.numCuInHeight - 1, a == numCtu - 1);
        for (int z = 0; z < npu; z++)
        {
            const CUGeom& g = geoms[leaves[z]];
            if ((int)g.absPartIdx != z * partsPerBlock) { rc = -6; break; }
            cu.initSubCU(ctus[a], g, 30);
            const int d = dir ? dir[(size_t)a * npu + z] : 1;
            const int32_t pk0 = mv0[((size_t)a * 85 + lbase + z) * 2 + 1];
            const int32_t pk1 = mv1 ? mv1[((size_t)a * 85 + lbase + z) * 2 + 1] : 0;
            for (int p = 0; p < partsPerBlock; p++)
            {
                cu.m_predMode[p] = MODE_INTER;
                cu.m_partSize[p] = SIZE_2Nx2N;
                cu.m_log2CUSize[p] = (uint8_t)log2n;
                cu.m_cuDepth[p] = (uint8_t)depth;
                cu.m_mv[0][p] = MV((int16_t)(pk0 & 0xffff), (int16_t)(pk0 >> 16));
                cu.m_mv[1][p] = MV((int16_t)(pk1 & 0xffff), (int16_t)(pk1 >> 16));
                cu.m_refIdx[0][p] = (d & 1) ? 0 : -1;
                cu.m_refIdx[1][p] = (d & 2) ? 0 : -1;
            }
            PredictionUnit pu(cu, g, 0);
            pred.motionCompensation(cu, pu, predYuv, true, true);
            const int px = (a % sps.numCuInWidth) * 64 + g_zscanToPelX[g.absPartIdx], py = row * 64 + g_zscanToPelY[g.absPartIdx];
            for (int y = 0; y < n; y++)
                memcpy((pixel*)predY + (size_t)(py + y) * width + px, predYuv.m_buf[0] + (size_t)y * predYuv.m_size, sizeof(pixel) * n);
            for (int y = 0; y < n / 2; y++)
            {
                memcpy((pixel*)predCb + (size_t)(py / 2 + y) * cw + px / 2, predYuv.m_buf[1] + (size_t)y * predYuv.m_csize, sizeof(pixel) * (n / 2));
                memcpy((pixel*)predCr + (size_t)(py / 2 + y) * cw + px / 2, predYuv.m_buf[2] + (size_t)y * predYuv.m_csize, sizeof(pixel) * (n / 2));
            }
        }
        if (rc) break;
    }
    predYuv.destroy();
    frame.m_reconPic = NULL; frame.m_encData = NULL;
    encData.m_picCTU = NULL; encData.m_slice = NULL;
    pool.destroy(); subPool.destroy();
    recon.destroy(); refs[0].destroy(); refs[1].destroy();
    x265_param_free(param);
    return rc;
}

/* The REAL Predict::predIntraLumaAng / predIntraChromaAng (predict.cpp:579-598) on caller-supplied neighbour arrays (the layout of
 * Predict::intraNeighbourBuf: corner, 2N above, 2N left; [0] unfiltered, [1] filtered).  dst: n x n, stride n. */
int x265ref_pred_intra(int mode, int log2Size, const void* unfiltered, const void* filtered, int chroma, void* dst)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    const int n = 1 << log2Size;
    Predict pred;
    memcpy(pred.intraNeighbourBuf[0], unfiltered, sizeof(pixel) * (4 * n + 1));
    memcpy(pred.intraNeighbourBuf[1], filtered, sizeof(pixel) * (4 * n + 1));
    if (chroma) pred.predIntraChromaAng((uint32_t)mode, (pixel*)dst, n, (uint32_t)log2Size);
    else pred.predIntraLumaAng((uint32_t)mode, (pixel*)dst, n, (uint32_t)log2Size);
    return 0;
}

} // extern "C"
