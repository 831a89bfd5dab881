/* oracle/x265_oracle_pipeline6.c - TEST INFRASTRUCTURE (checker), never part of the product path.
 *
 * CPU restatement of the rate-distortion decision of the sample-adaptive-offset parameters, SAO::rdoSaoUnitCu
 * (encoder/sao.cpp:1225-1376) with saoStatsInitialOffset (:1378-1433), estIterOffset (:1449-1483), saoLumaComponentParamDist
 * (:1484-1610) and saoChromaComponentParamDist (:1611-1760), over a whole picture the way FrameFilter drives it: every CTU row has
 * its own SAO object whose entropy contexts start from the slice's initial state (sao.cpp:245-247, framefilter.cpp:239), CTUs of a row
 * run left to right, the merge-up candidate reads the decision of the row above.
 *
 * What of the entropy coder the decision needs (bit-counting mode, m_bitIf == NULL): two context states - sao_merge_left/up_flag and
 * sao_type_idx (entropy.h:171-172) -, the 15 fractional bits that Entropy::resetBits keeps (entropy.cpp:2442-2451: m_fracBits &= 32767)
 * and Entropy::load / store copy (entropy.cpp:2432-2440); context-coded bins cost g_entropyBits[state ^ bin] (entropy.cpp:2457-2466),
 * bypass bins 32768 (:2503-2509); getNumberOfWrittenBits = m_fracBits >> 15 (entropy.h:120-124).  The state-transition table is derived
 * from the standard's transIdxLps (ITU-T H.265 table 9-46); the per-state bit costs are the host's own table and are handed in.
 * Limits: bLimitSAO = 0 and bSaoNonDeblocked = 0 (the x265 defaults).
 *
 * Pinned against the real class by tests/test_oracle_classes_vs_reference.py::test_sao_rdo_restatement_equals_the_real_class
 * (oracle/ref_sao.cpp::x265ref_sao_rdo). */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DEPTH        X265HIP_DEPTH
#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

enum { SAO_BO = 4, NUM_OFFSET = 4, NUM_CLASS = 32, OFFSET_THRESH = 1 << ((DEPTH - 5) < 5 ? (DEPTH - 5) : 5) };       /* sao.h:36-54 */

/* H.265 table 9-46, transIdxLps */
static const uint8_t kTransIdxLps6[64] = {
    0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
    24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };
static int next_state(int s, int bin)          /* context byte = pStateIdx << 1 | valMps (contexts.h:116 sbacNext) */
{
    const int p = s >> 1, mps = s & 1;
    if (bin == mps) return ((p < 62 ? p + 1 : p) << 1) | mps;
    return (kTransIdxLps6[p] << 1) | (p == 0 ? 1 - mps : mps);
}

typedef struct { int ctxMerge, ctxType; uint32_t frac; } Ent;          /* what Entropy::load / store move that SAO's syntax touches */
static const uint32_t* g_bits;
static void reset_bits(Ent* e) { e->frac &= 32767; }                                                   /* entropy.cpp:2442-2451 */
static void bin_ctx(Ent* e, int* ctx, int bin) { e->frac += g_bits[*ctx ^ bin]; *ctx = next_state(*ctx, bin); }   /* :2454-2466 */
static void bins_ep(Ent* e, int n) { e->frac += 32768u * (uint32_t)n; }                                /* :2500-2509, :2524-2530 */
static uint32_t written(const Ent* e) { return e->frac >> 15; }                                        /* entropy.h:120-124 */
static int iabs6(int v) { return v < 0 ? -v : v; }
static void max_uvlc(Ent* e, uint32_t code, uint32_t maxSymbol)                                        /* entropy.cpp:2198-2214 */
{
    bins_ep(e, 1);
    if (code) bins_ep(e, (int)(code - 1 + (maxSymbol > code)));
}
static void code_eo(Ent* e, const int* off, int plane)                                                 /* codeSaoOffsetEO, entropy.cpp:1258-1274 */
{
    if (plane != 2) { bin_ctx(e, &e->ctxType, 1); bins_ep(e, 1); }
    max_uvlc(e, (uint32_t)off[0], OFFSET_THRESH - 1); max_uvlc(e, (uint32_t)off[1], OFFSET_THRESH - 1);
    max_uvlc(e, (uint32_t)-off[2], OFFSET_THRESH - 1); max_uvlc(e, (uint32_t)-off[3], OFFSET_THRESH - 1);
    if (plane != 2) bins_ep(e, 2);
}
static void code_bo(Ent* e, const int* off, int plane)                                                 /* codeSaoOffsetBO, entropy.cpp:1276-1292 */
{
    if (plane != 2) { bin_ctx(e, &e->ctxType, 1); bins_ep(e, 1); }
    for (int i = 0; i < NUM_OFFSET; i++) max_uvlc(e, (uint32_t)iabs6(off[i]), OFFSET_THRESH - 1);
    for (int i = 0; i < NUM_OFFSET; i++) if (off[i]) bins_ep(e, 1);
    bins_ep(e, 5);
}
static void code_param(Ent* e, const int32_t* p, int plane)                                            /* codeSaoOffset, entropy.cpp:1221-1256 */
{
    const int typeIdx = p[0];
    if (plane != 2)
    {
        bin_ctx(e, &e->ctxType, typeIdx >= 0);
        if (typeIdx >= 0) bins_ep(e, 1);
    }
    if (typeIdx < 0) return;
    if (typeIdx == SAO_BO)
    {
        for (int i = 0; i < NUM_OFFSET; i++) max_uvlc(e, (uint32_t)iabs6(p[2 + i]), OFFSET_THRESH - 1);
        for (int i = 0; i < NUM_OFFSET; i++) if (p[2 + i]) bins_ep(e, 1);
        bins_ep(e, 5);
    }
    else
    {
        max_uvlc(e, (uint32_t)p[2], OFFSET_THRESH - 1); max_uvlc(e, (uint32_t)p[3], OFFSET_THRESH - 1);
        max_uvlc(e, (uint32_t)-p[4], OFFSET_THRESH - 1); max_uvlc(e, (uint32_t)-p[5], OFFSET_THRESH - 1);
        if (plane != 2) bins_ep(e, 2);
    }
}

static int64_t est_dist(int32_t count, int32_t offset, int32_t offsetOrg) { return (int64_t)(int32_t)((count * offset - offsetOrg * 2) * offset); }   /* sao.cpp:56-59: int arithmetic */
static int64_t rd_cost(int64_t dist, uint32_t bits, int64_t lambda) { return dist + (((int64_t)bits * lambda + 128) >> 8); }                          /* sao.cpp:1436-1447 */
static int32_t round_ibdi(int32_t num, int32_t den) { return num >= 0 ? ((num * 2 + den) / (den * 2)) : -((-num * 2 + den) / (den * 2)); }            /* sao.cpp:34-37 */
static int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* saoStatsInitialOffset for one plane of one CTU (sao.cpp:1378-1433; bLimitSAO 0: all four edge types) */
static void initial_offsets(const int32_t* cnt, const int32_t* org, int32_t (*off)[NUM_CLASS])
{
    memset(off, 0, sizeof(int32_t) * 5 * NUM_CLASS);
    for (int t = 0; t < 4; t++)
        for (int c = 1; c < NUM_OFFSET + 1; c++)
            if (cnt[t * 32 + c])
            {
                int o = clip3i(-OFFSET_THRESH + 1, OFFSET_THRESH - 1, round_ibdi(org[t * 32 + c], cnt[t * 32 + c]));
                off[t][c] = c < 3 ? (o > 0 ? o : 0) : (o < 0 ? o : 0);
            }
    for (int c = 0; c < NUM_CLASS; c++)
        if (cnt[SAO_BO * 32 + c])
            off[SAO_BO][c] = clip3i(-OFFSET_THRESH + 1, OFFSET_THRESH - 1, round_ibdi(org[SAO_BO * 32 + c], cnt[SAO_BO * 32 + c]));
}

/* estIterOffset (sao.cpp:1449-1483) */
static void iter_offset(int typeIdx, int64_t lambda, int32_t count, int32_t offsetOrg, int32_t* offset, int32_t* distClass, int64_t* costClass)
{
    int bestOffset = 0;
    *distClass = 0;
    int64_t bestCost = rd_cost(0, 1, lambda);
    int o = *offset;
    while (o != 0)
    {
        uint32_t rate = (typeIdx == SAO_BO) ? (uint32_t)(iabs6(o) + 2) : (uint32_t)(iabs6(o) + 1);
        if (iabs6(o) == OFFSET_THRESH - 1) rate--;
        const int64_t dist = est_dist(count, o, offsetOrg);
        const int64_t cost = rd_cost(dist, rate, lambda);
        if (cost < bestCost) { bestCost = cost; bestOffset = o; *distClass = (int)dist; }
        o = o > 0 ? o - 1 : o + 1;
    }
    *costClass = bestCost;
    *offset = bestOffset;
}

static void param_reset(int32_t* p) { p[0] = -1; p[1] = 0; p[2] = p[3] = p[4] = p[5] = 0; p[6] = 0; }   /* SaoCtuParam::reset, common.h:360-371 */

/* count / offsetOrg: [planes] pointers to int32 [nctu][5][32] (x265oracle_sao_stats_plane); lambdaCtu: int64 [nctu][2] =
 * floor(256 * x265_lambda2_tab[qp]) for luma and for chroma at the Cb QP (sao.cpp:1229-1238); saoFlag = saoParam->bSaoFlag;
 * params: [planes] int32 [nctu][7] = { typeIdx, bandPos, offset[4], mergeMode (0 none, 1 left, 2 up) }; numNoSao: int32 [2]. */
int EXPORT(x265oracle_sao_rdo)(const int32_t* const* count, const int32_t* const* offsetOrg, int planes, int ctusW, int ctusH, const int64_t* lambdaCtu,
                               int ctxMerge, int ctxType, const uint32_t* entropyBits, const int* saoFlag, int32_t* const* params, int32_t* numNoSao)
{
    if ((planes != 1 && planes != 3) || !entropyBits) return -1;
    g_bits = entropyBits;
    numNoSao[0] = numNoSao[1] = 0;
    const int chroma = planes == 3;
    for (int row = 0; row < ctusH; row++)
    {
        Ent cur = { ctxMerge, ctxType, 0 };                                  /* m_rdContexts.cur.load(initState), sao.cpp:247 */
        for (int col = 0; col < ctusW; col++)
        {
            const int addr = row * ctusW + col;
            const int64_t lambda[2] = { lambdaCtu[2 * addr], lambdaCtu[2 * addr + 1] };
            const int allowMerge[2] = { col != 0, row != 0 };
            const int addrMerge[2] = { col ? addr - 1 : -1, row ? addr - ctusW : -1 };
            int32_t off[3][5][NUM_CLASS];
            memset(off, 0, sizeof(off));
            for (int i = 0; i < planes; i++) param_reset(params[i] + (size_t)addr * 7);
            Ent e = cur, temp;
            reset_bits(&e);
            if (allowMerge[0]) bin_ctx(&e, &e.ctxMerge, 0);
            if (allowMerge[1]) bin_ctx(&e, &e.ctxMerge, 0);
            temp = e;
            int64_t bestCost = 0, rateDist = 0;

            if (saoFlag[0])
            {   /* ---- saoLumaComponentParamDist (sao.cpp:1484-1610) */
                const int32_t* cnt = count[0] + (size_t)addr * 160;
                const int32_t* org = offsetOrg[0] + (size_t)addr * 160;
                initial_offsets(cnt, org, off[0]);
                int32_t* lp = params[0] + (size_t)addr * 7;
                int64_t bestDist = 0;
                int bestTypeIdx = -1;
                int32_t distClasses[NUM_CLASS];
                int64_t costClasses[NUM_CLASS];
                e = temp; reset_bits(&e); bin_ctx(&e, &e.ctxType, 0);
                int64_t costPartBest = rd_cost(0, written(&e), lambda[0]);
                for (int t = 0; t < 4; t++)
                {
                    int64_t estDist = 0;
                    for (int c = 1; c < NUM_OFFSET + 1; c++)
                    {
                        iter_offset(t, lambda[0], cnt[t * 32 + c], org[t * 32 + c], &off[0][t][c], &distClasses[c], &costClasses[c]);
                        estDist += distClasses[c];
                    }
                    e = temp; reset_bits(&e); code_eo(&e, off[0][t] + 1, 0);
                    const int64_t cost = rd_cost(estDist, written(&e), lambda[0]);
                    if (cost < costPartBest) { costPartBest = cost; bestDist = estDist; bestTypeIdx = t; }
                }
                if (bestTypeIdx != -1)
                {
                    lp[6] = 0; lp[0] = bestTypeIdx; lp[1] = 0;
                    for (int c = 0; c < NUM_OFFSET; c++) lp[2 + c] = off[0][bestTypeIdx][c + 1];
                }
                for (int c = 0; c < NUM_CLASS; c++)
                    iter_offset(SAO_BO, lambda[0], cnt[SAO_BO * 32 + c], org[SAO_BO * 32 + c], &off[0][SAO_BO][c], &distClasses[c], &costClasses[c]);
                int bestClassBO = 0;
                int64_t currentRDCost = costClasses[0] + costClasses[1] + costClasses[2] + costClasses[3];
                int64_t bestRDCostBO = currentRDCost;
                for (int i = 1; i < NUM_CLASS - NUM_OFFSET + 1; i++)
                {
                    currentRDCost -= costClasses[i - 1];
                    currentRDCost += costClasses[i + 3];
                    if (currentRDCost < bestRDCostBO) { bestRDCostBO = currentRDCost; bestClassBO = i; }
                }
                int64_t estDist = 0;
                for (int c = bestClassBO; c < bestClassBO + NUM_OFFSET; c++) estDist += distClasses[c];
                e = temp; reset_bits(&e); code_bo(&e, off[0][SAO_BO] + bestClassBO, 0);
                const int64_t cost = rd_cost(estDist, written(&e), lambda[0]);
                if (cost < costPartBest)
                {
                    costPartBest = cost; bestDist = estDist;
                    lp[6] = 0; lp[0] = SAO_BO; lp[1] = bestClassBO;
                    for (int c = 0; c < NUM_OFFSET; c++) lp[2 + c] = off[0][SAO_BO][c + bestClassBO];
                }
                rateDist = (bestDist << 8) / lambda[0];
                e = temp; code_param(&e, lp, 0); temp = e;                  /* no resetBits here: the merge flags' bits stay counted (:1598-1600) */
                if (!chroma) bestCost = rateDist + written(&e);              /* X265_CSP_I400 (:1602-1605) */
            }
            if (chroma && saoFlag[1])
            {   /* ---- saoChromaComponentParamDist (sao.cpp:1611-1760) */
                int32_t* cp[2] = { params[1] + (size_t)addr * 7, params[2] + (size_t)addr * 7 };
                for (int k = 1; k < 3; k++) initial_offsets(count[k] + (size_t)addr * 160, offsetOrg[k] + (size_t)addr * 160, off[k]);
                int64_t bestDist = 0;
                int bestTypeIdx = -1;
                int32_t distClasses[NUM_CLASS];
                int64_t costClasses[NUM_CLASS];
                int bestClassBO[2] = { 0, 0 };
                e = temp; reset_bits(&e); bin_ctx(&e, &e.ctxType, 0);
                int64_t costPartBest = rd_cost(0, written(&e), lambda[1]);
                for (int t = 0; t < 4; t++)
                {
                    int64_t estDist[2] = { 0, 0 };
                    for (int k = 1; k < 3; k++)
                        for (int c = 1; c < NUM_OFFSET + 1; c++)
                        {
                            iter_offset(t, lambda[1], count[k][(size_t)addr * 160 + t * 32 + c], offsetOrg[k][(size_t)addr * 160 + t * 32 + c], &off[k][t][c],
                                        &distClasses[c], &costClasses[c]);
                            estDist[k - 1] += distClasses[c];
                        }
                    e = temp; reset_bits(&e);
                    for (int k = 0; k < 2; k++) code_eo(&e, off[k + 1][t] + 1, k + 1);
                    const int64_t cost = rd_cost(estDist[0] + estDist[1], written(&e), lambda[1]);
                    if (cost < costPartBest) { costPartBest = cost; bestDist = estDist[0] + estDist[1]; bestTypeIdx = t; }
                }
                if (bestTypeIdx != -1)
                    for (int k = 0; k < 2; k++)
                    {
                        cp[k][6] = 0; cp[k][0] = bestTypeIdx; cp[k][1] = 0;
                        for (int c = 0; c < NUM_OFFSET; c++) cp[k][2 + c] = off[k + 1][bestTypeIdx][c + 1];
                    }
                int64_t estDist[2];
                for (int k = 1; k < 3; k++)
                {
                    int64_t bestRDCostBO = INT64_MAX;
                    for (int c = 0; c < NUM_CLASS; c++)
                        iter_offset(SAO_BO, lambda[1], count[k][(size_t)addr * 160 + SAO_BO * 32 + c], offsetOrg[k][(size_t)addr * 160 + SAO_BO * 32 + c],
                                    &off[k][SAO_BO][c], &distClasses[c], &costClasses[c]);
                    for (int i = 0; i < NUM_CLASS - NUM_OFFSET + 1; i++)
                    {
                        int64_t currentRDCost = 0;
                        for (int j = i; j < i + NUM_OFFSET; j++) currentRDCost += costClasses[j];
                        if (currentRDCost < bestRDCostBO) { bestRDCostBO = currentRDCost; bestClassBO[k - 1] = i; }
                    }
                    estDist[k - 1] = 0;
                    for (int c = bestClassBO[k - 1]; c < bestClassBO[k - 1] + NUM_OFFSET; c++) estDist[k - 1] += distClasses[c];
                }
                e = temp; reset_bits(&e);
                for (int k = 0; k < 2; k++) code_bo(&e, off[k + 1][SAO_BO] + bestClassBO[k], k + 1);
                const int64_t cost = rd_cost(estDist[0] + estDist[1], written(&e), lambda[1]);
                if (cost < costPartBest)
                {
                    costPartBest = cost; bestDist = estDist[0] + estDist[1];
                    for (int k = 0; k < 2; k++)
                    {
                        cp[k][6] = 0; cp[k][0] = SAO_BO; cp[k][1] = bestClassBO[k];
                        for (int c = 0; c < NUM_OFFSET; c++) cp[k][2 + c] = off[k + 1][SAO_BO][c + bestClassBO[k]];
                    }
                }
                rateDist += (bestDist << 8) / lambda[1];
                e = temp;
                code_param(&e, cp[0], 1); code_param(&e, cp[1], 2);
                temp = e;
                bestCost = rateDist + written(&e);
            }
            if (saoFlag[0] || saoFlag[1])
            {   /* ---- merge candidates (sao.cpp:1314-1373) */
                for (int m = 0; m < 2; m++)
                {
                    if (!allowMerge[m]) continue;
                    int64_t mergeDist = 0;
                    for (int pl = 0; pl < planes; pl++)
                    {
                        int64_t estDist = 0;
                        const int32_t* src = params[pl] + (size_t)addrMerge[m] * 7;
                        const int typeIdx = src[0];
                        if (typeIdx >= 0)
                        {
                            const int bandPos = typeIdx == SAO_BO ? src[1] : 1;
                            for (int c = 0; c < NUM_OFFSET; c++)
                                estDist += est_dist(count[pl][(size_t)addr * 160 + typeIdx * 32 + c + bandPos], src[2 + c],
                                                    offsetOrg[pl][(size_t)addr * 160 + typeIdx * 32 + c + bandPos]);
                        }
                        mergeDist += (estDist << 8) / lambda[!!pl];
                    }
                    e = cur; reset_bits(&e);
                    if (allowMerge[0]) bin_ctx(&e, &e.ctxMerge, 1 - m);
                    if (allowMerge[1] && m == 1) bin_ctx(&e, &e.ctxMerge, 1);
                    const int64_t mergeCost = mergeDist + written(&e);
                    if (mergeCost < bestCost)
                    {
                        bestCost = mergeCost;
                        temp = e;
                        for (int pl = 0; pl < planes; pl++)
                            if (saoFlag[pl > 0])
                            {
                                int32_t* dst = params[pl] + (size_t)addr * 7;
                                const int32_t* src = params[pl] + (size_t)addrMerge[m] * 7;
                                dst[6] = m ? 2 : 1; dst[0] = src[0]; dst[1] = src[1];
                                for (int i = 0; i < NUM_OFFSET; i++) dst[2 + i] = src[2 + i];
                            }
                    }
                }
                if (params[0][(size_t)addr * 7] < 0) numNoSao[0]++;
                if (chroma && params[1][(size_t)addr * 7] < 0) numNoSao[1]++;
                cur = temp;
            }
        }
    }
    return 0;
}
