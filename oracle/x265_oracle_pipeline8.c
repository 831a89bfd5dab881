/* oracle/x265_oracle_pipeline8.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * Stage: the SUB-SAMPLE COST TABLES (x265hip_cost_candidates / x265hip_cost_tables, include/x265hip.h).  Restated ON TOP OF THE ORACLE'S
 * PRIMITIVE TABLE - whose entries are pinned against the real reference - exactly the way the reference produces each value:
 *   MotionEstimate::subpelCompare (motion.cpp:1571-1664): the PU's own partition entries pu[part].luma_hpp / luma_vpp / luma_hvpp into a
 *   scratch block of stride blockwidth, pu[part].satd against the source block; with bChromaSATD (motion.cpp:212: subme > 2)
 *   chroma[I420].pu[part].filter_hpp / filter_vpp / filter_hps(isRowExt) + filter_vsp(row halfFilterSize - 1) and chroma[I420].pu[part].satd
 *   for Cb and Cr, the luma quarter-sample vector read as an eighth-sample chroma vector (:1606-1607).
 * i.e. NOT tile by tile and NOT from phase planes: the product's route (4x4 tiles of precomputed phase planes) must meet these integers.
 * The position set restates the refinement loop's reach (motion.cpp:1456-1561 with the SubpelWorkload rows :48-58 and square1 :44); the PU
 * list restates the partition geometry of primitives.h:41-55 restricted to unions of 8x8 blocks. */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef x265hip_pixel pixel;
#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

void EXPORT(x265oracle_prims_once)(x265hip_EncoderPrimitives* p, int* state);

typedef struct { int x, y, w, h, part; } OraclePu;

static int part_of(int w, int h)
{
    static const int dims[X265HIP_NUM_PU_SIZES][2] = { {4,4},{8,8},{16,16},{32,32},{64,64},{8,4},{4,8},{16,8},{8,16},{32,16},{16,32},{64,32},{32,64},
                                                        {16,12},{12,16},{16,4},{4,16},{32,24},{24,32},{32,8},{8,32},{64,48},{48,64},{64,16},{16,64} };      /* primitives.h:41-55 */
    for (int i = 0; i < X265HIP_NUM_PU_SIZES; i++)
        if (dims[i][0] == w && dims[i][1] == h) return i;
    return -1;
}

static void unz(int z, int* ux, int* uy)
{
    *ux = *uy = 0;
    for (int b = 0; b < 3; b++) { *ux |= ((z >> (2 * b)) & 1) << b; *uy |= ((z >> (2 * b + 1)) & 1) << b; }
}

/* the list of x265hip_cost_pu_rect: squares (surface order), then 2NxN / Nx2N of the 16 / 32 / 64 CUs, then the AMP parts of the 32 / 64 CUs */
static int pu_list(int shapes, OraclePu* out)
{
    int n = 0;
#define ADD(X, Y, W, H) do { out[n].x = (X); out[n].y = (Y); out[n].w = (W); out[n].h = (H); out[n].part = part_of((W), (H)); n++; } while (0)
    for (int size = 8; size <= 64; size <<= 1)
        for (int z = 0; z < (64 / size) * (64 / size); z++) { int ux, uy; unz(z, &ux, &uy); ADD(ux * size, uy * size, size, size); }
    if (shapes >= 1)
        for (int size = 16; size <= 64; size <<= 1)
            for (int z = 0; z < (64 / size) * (64 / size); z++)
            {
                int ux, uy; unz(z, &ux, &uy);
                const int x = ux * size, y = uy * size, hs = size >> 1;
                ADD(x, y, size, hs); ADD(x, y + hs, size, hs); ADD(x, y, hs, size); ADD(x + hs, y, hs, size);
            }
    if (shapes >= 2)
        for (int size = 32; size <= 64; size <<= 1)
            for (int z = 0; z < (64 / size) * (64 / size); z++)
            {
                int ux, uy; unz(z, &ux, &uy);
                const int x = ux * size, y = uy * size, q = size >> 2;
                ADD(x, y, size, q); ADD(x, y + q, size, size - q);           /* 2NxnU */
                ADD(x, y, size, size - q); ADD(x, y + size - q, size, q);    /* 2NxnD */
                ADD(x, y, q, size); ADD(x + q, y, size - q, size);           /* nLx2N */
                ADD(x, y, size - q, size); ADD(x + size - q, y, q, size);    /* nRx2N */
            }
#undef ADD
    return n;
}

int EXPORT(x265oracle_cost_pu_list)(int shapes, int* rects)      /* rects[5 n]: x, y, w, h, LumaPU enum */
{
    OraclePu pu[209];
    const int n = pu_list(shapes, pu);
    if (rects) for (int i = 0; i < n; i++) { rects[5 * i] = pu[i].x; rects[5 * i + 1] = pu[i].y; rects[5 * i + 2] = pu[i].w; rects[5 * i + 3] = pu[i].h; rects[5 * i + 4] = pu[i].part; }
    return n;
}

/* every quarter-sample offset the refinement of workload row `subme` can measure, raster order; returns the count.  A walk of the loop
 * itself: the start, then hpel_iters rounds that may move to any of the hpel_dirs neighbours at distance 2, then qpel_iters rounds at distance 1 */
int EXPORT(x265oracle_cost_positions)(int subme, int8_t* xy)
{
    static const int wl[8][4] = { {1,4,0,4},{1,4,1,4},{1,4,1,4},{2,4,1,4},{2,4,2,4},{1,8,1,8},{2,8,1,8},{2,8,2,8} };       /* hpel_iters, hpel_dirs, qpel_iters, qpel_dirs: motion.cpp:48-58 */
    static const int sq[9][2] = { {0,0},{0,-1},{0,1},{-1,0},{1,0},{-1,-1},{-1,1},{1,-1},{1,1} };                             /* square1, motion.cpp:44 */
    unsigned char here[13][13], seen[13][13], next[13][13];
    memset(here, 0, sizeof(here)); memset(seen, 0, sizeof(seen));
    here[6][6] = seen[6][6] = 1;
    for (int phase = 0; phase < 2; phase++)
        for (int it = 0; it < wl[subme][phase * 2]; it++)
        {
            memcpy(next, here, sizeof(next));
            for (int y = 0; y < 13; y++)
                for (int x = 0; x < 13; x++)
                    if (here[y][x])
                        for (int i = 1; i <= wl[subme][phase * 2 + 1]; i++)
                        {
                            const int nx = x + sq[i][0] * (phase ? 1 : 2), ny = y + sq[i][1] * (phase ? 1 : 2);
                            seen[ny][nx] = next[ny][nx] = 1;
                        }
            memcpy(here, next, sizeof(here));
        }
    int n = 0;
    for (int y = 0; y < 13; y++)
        for (int x = 0; x < 13; x++)
            if (seen[y][x]) { if (xy) { xy[2 * n] = (int8_t)(x - 6); xy[2 * n + 1] = (int8_t)(y - 6); } n++; }
    return n;
}

/* SAD rasters of the 85 squares (int32 [ctu][row][group][85][4], x265hip_me_fullsearch's X265HIP_SURF_I32) -> per PU the K displacements of
 * smallest SAD (+ the optional vector cost of the displacement relative to the window centre, uint16 [2 window + 1] per component), ties to the earlier raster position (the full search's scan order and strict '<', motion.cpp:1395-1430) */
void EXPORT(x265oracle_cost_candidates)(const int32_t* surf, const int16_t* centres, int nctu, int window, int shapes, int K, int16_t* cand, const uint16_t* mvCost)
{
    OraclePu pu[209];
    const int npu = pu_list(shapes, pu), nc = 2 * window + 1, ng = (nc + 3) >> 2;
    for (int ctu = 0; ctu < nctu; ctu++)
        for (int p = 0; p < npu; p++)
        {
            uint64_t b1 = ~0ull, b2 = ~0ull;
            for (int row = 0; row < nc; row++)
                for (int col = 0; col < nc; col++)
                {
                    const int32_t* rec = surf + (((size_t)ctu * nc + row) * ng + (col >> 2)) * 340 + (col & 3);
                    uint32_t sad = 0;
                    /* the 8x8 blocks the PU covers: pu[LUMA_WxH].sad is their sum */
                    for (int by = pu[p].y / 8; by < (pu[p].y + pu[p].h) / 8; by++)
                        for (int bx = pu[p].x / 8; bx < (pu[p].x + pu[p].w) / 8; bx++)
                        {
                            int z = 0;
                            for (int b = 0; b < 3; b++) z |= ((bx >> b) & 1) << (2 * b) | ((by >> b) & 1) << (2 * b + 1);
                            sad += (uint32_t)rec[z * 4];
                        }
                    if (mvCost) sad += (uint32_t)mvCost[col] + (uint32_t)mvCost[row];          /* the search's own objective: sad + mvcost(mv - predictor), motion.cpp:246-328 */
                    const uint64_t key = (uint64_t)sad << 32 | (uint32_t)(row * nc + col);
                    if (key < b1) { b2 = b1; b1 = key; } else if (key < b2) b2 = key;
                }
            for (int k = 0; k < K; k++)
            {
                const uint64_t key = k ? b2 : b1;
                int16_t* o = cand + ((size_t)(ctu * npu + p) * K + k) * 2;
                if (key == ~0ull) { o[0] = -32768; o[1] = 0; continue; }
                const int d = (int)(uint32_t)key;
                o[0] = (int16_t)((centres ? centres[2 * ctu] : 0) + d % nc - window);
                o[1] = (int16_t)((centres ? centres[2 * ctu + 1] : 0) + d / nc - window);
            }
        }
}

/* planes: ALLOCATION STARTS (PicYuv layout); tables: the band's first CTU first, records of x265hip_cost_record_bytes */
void EXPORT(x265oracle_cost_tables)(const pixel* const* fenc, const pixel* const* ref, intptr_t stride, intptr_t strideC, int marginX, int marginY, int marginYC,
                                    int width, int ctuRow0, int ctuRows, int shapes, int K, int subme, int chroma, const int16_t* cand, uint8_t* tables, int sadCosts)
{
    static x265hip_EncoderPrimitives prim;
    static int ready;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    OraclePu pu[209];
    int8_t pos[169 * 2];
    const int npu = pu_list(shapes, pu), npos = EXPORT(x265oracle_cost_positions)(subme, pos), rec2Off = (8 + 2 * npos + 3) & ~3, ctusW = width / 64;
    const int recBytes = sadCosts ? rec2Off + ((4 + 2 * npos + 3) & ~3) : rec2Off;
#pragma omp parallel for schedule(dynamic)
    for (int job = 0; job < ctuRows * ctusW * npu; job++)
    {
        const int ctuB = job / npu, p = job % npu, ctuX = ctuB % ctusW, ctuY = ctuRow0 + ctuB / ctusW;
        const OraclePu* P = &pu[p];
        pixel fencBuf[3][64 * 64];                                /* fencPUYuv: stride FENC_STRIDE = 64 (luma), 32 (chroma) - common.h:70, yuv.cpp:126-140 */
        pixel subpelbuf[64 * 64];
        int16_t immed[64 * (64 + 8 - 1)];
        const int X0 = ctuX * 64 + P->x, Y0 = ctuY * 64 + P->y;
        for (int y = 0; y < P->h; y++) memcpy(fencBuf[0] + y * 64, fenc[0] + (size_t)(marginY + Y0 + y) * stride + marginX + X0, P->w * sizeof(pixel));
        if (chroma)
            for (int c = 1; c < 3; c++)
                for (int y = 0; y < P->h / 2; y++) memcpy(fencBuf[c] + y * 32, fenc[c] + (size_t)(marginYC + Y0 / 2 + y) * strideC + marginX + X0 / 2, P->w / 2 * sizeof(pixel));
        for (int k = 0; k < K; k++)
        {
            const size_t recIdx = (size_t)(ctuB * npu + p) * K + k;
            uint8_t* rec = tables + recIdx * recBytes;
            const int mvx = cand[recIdx * 2], mvy = cand[recIdx * 2 + 1];
            memset(rec, 0, recBytes);
            if (mvx == -32768) { ((int16_t*)rec)[0] = -32768; continue; }
            for (int which = 0; which < (sadCosts ? 2 : 1); which++)
            {
            /* which = 0: cmp = the PU's satd (the refinement's comparisons); 1: cmp = its sad (the predictor candidates', motion.cpp:773-812) - the chroma part is
             * chromaSatd either way (:1601-1661) */
            uint32_t cost[169], lo = ~0u;
            for (int i = 0; i < npos; i++)
            {
                const int qx = mvx * 4 + pos[2 * i], qy = mvy * 4 + pos[2 * i + 1];
                /* subpelCompare, luma (motion.cpp:1573-1599) */
                const pixel* fref = ref[0] + (size_t)(marginY + Y0 + (qy >> 2)) * stride + marginX + X0 + (qx >> 2);
                int xFrac = qx & 3, yFrac = qy & 3, c;
                x265hip_pixelcmp_t cmp = which ? prim.pu[P->part].sad : prim.pu[P->part].satd;
                if (!(yFrac | xFrac)) c = cmp(fencBuf[0], 64, fref, stride);
                else
                {
                    if (!yFrac) prim.pu[P->part].luma_hpp(fref, stride, subpelbuf, P->w, xFrac);
                    else if (!xFrac) prim.pu[P->part].luma_vpp(fref, stride, subpelbuf, P->w, yFrac);
                    else prim.pu[P->part].luma_hvpp(fref, stride, subpelbuf, P->w, xFrac, yFrac);
                    c = cmp(fencBuf[0], 64, subpelbuf, P->w);
                }
                if (chroma)
                {
                    /* :1601-1661 at 4:2:0 (hshift = vshift = 1: mvx = qmv.x, mvy = qmv.y in eighth samples) */
                    const intptr_t refOffset = (qx >> 3) + (intptr_t)(qy >> 3) * strideC;
                    xFrac = qx & 7; yFrac = qy & 7;
                    const int wc = P->w >> 1;
                    for (int comp = 1; comp < 3; comp++)
                    {
                        const pixel* r = ref[comp] + (size_t)(marginYC + Y0 / 2) * strideC + marginX + X0 / 2 + refOffset;
                        if (!(yFrac | xFrac)) c += prim.chroma[1].pu[P->part].satd(fencBuf[comp], 32, r, strideC);
                        else
                        {
                            if (!yFrac) prim.chroma[1].pu[P->part].filter_hpp(r, strideC, subpelbuf, wc, xFrac);
                            else if (!xFrac) prim.chroma[1].pu[P->part].filter_vpp(r, strideC, subpelbuf, wc, yFrac);
                            else
                            {
                                prim.chroma[1].pu[P->part].filter_hps(r, strideC, immed, wc, xFrac, 1);
                                prim.chroma[1].pu[P->part].filter_vsp(immed + (4 / 2 - 1) * wc, wc, subpelbuf, wc, yFrac);
                            }
                            c += prim.chroma[1].pu[P->part].satd(fencBuf[comp], 32, subpelbuf, wc);
                        }
                    }
                }
                cost[i] = (uint32_t)c;
                if (cost[i] < lo) lo = cost[i];
            }
            uint8_t* part = which ? rec + rec2Off : rec + 4;
            *(uint32_t*)part = lo;
            for (int i = 0; i < npos; i++) ((uint16_t*)(part + 4))[i] = (uint16_t)(cost[i] - lo > 65535u ? 65535u : cost[i] - lo);
            }
            ((int16_t*)rec)[0] = (int16_t)mvx; ((int16_t*)rec)[1] = (int16_t)mvy;
        }
    }
}
