// lowres_cost_kernels.hip - the lookahead's P-frame cost estimate on gfx950 (the motion half).
//
// Reference semantics: CostEstimateGroup::estimateFrameCost / estimateCUCost for b == p1 (encoder/slicetype.cpp:3189-3198,
// 3216-3388) over MotionEstimate::motionEstimate's lowres flavour (encoder/motion.cpp:775-776 predictor measure, HEX :852-948,
// :1452-1469, sub-pel :1471-1503) and ReferencePlanes::lowresQPelCost / lowresMC (common/lowres.h:66-121: half-pel positions
// read one of the four phase planes, quarter-pel positions average two with pixelavg_pp).
//
// Every 8x8 block needs the final mvs of its right neighbour and of three blocks of the row below, so a picture is a wavefront:
// row r (counted from the bottom) works on block W-1-(t-2r) at step t.  One workgroup walks it in lockstep - a DPP quad per row
// (lane = one 4x4 tile of the block), one barrier per step, mvs exchanged through the output array with workgroup-scope accesses.
// The step time is the latency of one HEX search, so a single picture cannot fill the chip: callers batch independent pictures
// (frame-parallel encoding has them) on separate streams.
//
// SPLIT (round 3): one estimate alone - the lookahead seam's case, a pool thread waits for it - is bound by the ONE compute unit its workgroup
// sits on: 69 % of a wavefront's time is s_waitcnt, ~21 k cache-line reads per step go through that CU's L1 (profiles/r03_lowres_cost_counters.txt).
// The split form gives a picture K workgroups, each a band of R consecutive block rows on a CU of its own.  Inside a band nothing changes
// (lockstep steps, one barrier per step); a band's bottom row takes the three finished mvs of the row below from the band below through L2:
// the producer's top row publishes every finished mv as ONE tagged 64-bit word (agent-scope store), the consumer's bottom-row quad spins on
// the three words it needs (agent-scope loads).  Band k only ever waits for band k - 1, which has the lower workgroup index and is dispatched
// first, so the wait cannot starve (the ordering decoupled look-back scans rely on).  The frame sums meet in agent-scope accumulators; the
// last band to arrive writes them out.  Same integers as the one-workgroup walk, which stays in charge of batches.
#include "pu_eval.h"
#include <atomic>

namespace x265hip {

struct LowresCostArgs
{
    const x265hip_lowres_cost_pair* pairs;     // device copy, one per workgroup
    int strideB, W, H, depth, bFrameBias;
    const uint16_t* cost;
    int K, R;                                  // SPLIT: workgroups per picture, block rows per workgroup
    unsigned long long* sync;                  // SPLIT: per picture [2 lists][K][W] tagged boundary mvs + 3 frame sums + arrival count, zeroed before the launch
};

constexpr unsigned kSplitTag = 0xA5C3u;        // upper 16 bits of a published word (the buffer starts as zeros)
__host__ __device__ inline size_t split_sync_words(int K, int W) { return (size_t)2 * K * W + 4; }

// the four phase planes, biased like PuEval::base; passed by value so that they stay in registers
struct PhasePlanes { const uint8_t *p0, *p1, *p2, *p3; };

template <typename Px>
struct LowresPu
{
    static constexpr int BPP = sizeof(Px);
    PuEval<Px, 4, 1> c;

    static __device__ __forceinline__ const uint8_t* plane(const PhasePlanes pp, int h) { return h == 0 ? pp.p0 : (h == 1 ? pp.p1 : (h == 2 ? pp.p2 : pp.p3)); }
    __device__ __forceinline__ void load_tile(const uint8_t* base, uint32_t off, int (&p)[4][4]) const
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const uint8_t* rp = base + (off + (uint32_t)(r * c.strideB));
            if (BPP == 1)
            {
                const uint32_t w = ld_u32(rp);
#pragma unroll
                for (int x = 0; x < 4; x++) p[r][x] = (int)((w >> (8 * x)) & 0xff);
            }
            else
            {
                const uint32_t w0 = ld_u32(rp), w1 = ld_u32(rp + 4);
                p[r][0] = (int)(w0 & 0xffff); p[r][1] = (int)(w0 >> 16); p[r][2] = (int)(w1 & 0xffff); p[r][3] = (int)(w1 >> 16);
            }
        }
    }
    // lowresMC (lowres.h:66-93): this lane's 4x4 tile of the prediction at quarter-pel (qx, qy)
    __device__ __forceinline__ void predict(const PhasePlanes pp, int qx, int qy, int (&p)[4][4]) const
    {
        const int hA = (qy & 2) | ((qx & 2) >> 1);
        load_tile(plane(pp, hA), c.refOrg[0] + (uint32_t)((qy >> 2) * c.strideB + (qx >> 2) * BPP), p);
        if ((qx | qy) & 1)
        {
            int pb[4][4];
            const int qx2 = qx + (qx & 1), qy2 = qy + (qy & 1);
            const int hB = (qy2 & 2) | ((qx2 & 2) >> 1);
            load_tile(plane(pp, hB), c.refOrg[0] + (uint32_t)((qy2 >> 2) * c.strideB + (qx2 >> 2) * BPP), pb);
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) p[y][x] = (p[y][x] + pb[y][x] + 1) >> 1;          // pixelavg_pp, weight 32
        }
    }
    // sad / satd of the block against a prediction held tile by tile
    __device__ __forceinline__ int score(int (&p)[4][4], bool useSatd) const
    {
        int acc = 0;
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                const int s = BPP == 1 ? (int)((c.src[0][y][0] >> (8 * x)) & 0xff) : (int)((c.src[0][y][x >> 1] >> (16 * (x & 1))) & 0xffff);
                p[y][x] = s - p[y][x];
            }
        if (useSatd) acc = tile_satd4(p);
        else
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) acc += abs(p[y][x]);
        }
        return quad_sum(acc);
    }
    // lowresQPelCost (lowres.h:95-121) with sad or satd; also lowresMC + bufSATD (slicetype.cpp:3293-3295)
    __device__ __forceinline__ int qpel_cost(const PhasePlanes pp, int qx, int qy, bool useSatd) const
    {
        int p[4][4];
        predict(pp, qx, qy, p);
        return score(p, useSatd);
    }
};

// MotionEstimate::motionEstimate, lowres flavour, HEX, merange 16, subpelRefine 1, no extra candidates
template <typename Px>
__device__ __forceinline__ int lowres_motion_estimate(const LowresPu<Px>& L, const PhasePlanes pp, int& outx, int& outy)
{
    const PuEval<Px, 4, 1>& c = L.c;
    const int qminx = c.mvmin.x * 4, qminy = c.mvmin.y * 4, qmaxx = c.mvmax.x * 4, qmaxy = c.mvmax.y * 4;
    const int pmvx = s_clip3(qminx, qmaxx, c.mvpx), pmvy = s_clip3(qminy, qmaxy, c.mvpy);
    // the three start measures are independent: issue them together (one memory round trip), then decide in the reference's order
    SMv bmv = { (pmvx + 2) >> 2, (pmvy + 2) >> 2 };
    const int bprecost = L.qpel_cost(pp, pmvx, pmvy, false);
    const int roundedCost = c.cost_mv(bmv.x, bmv.y);
    int zeroCost = c.sad_at(0, 0) + c.mvcost_q(0, 0);
    pin_value(zeroCost);
    int bcost = bprecost;
    if ((pmvx | pmvy) & 3) bcost = roundedCost;
    if (pmvx | pmvy)
    {
        const int cost = zeroCost;
        if (cost < bcost)
        {
            bcost = cost;
            bmv.x = 0;
            const int zy = 0 < c.mvmax.y ? 0 : c.mvmax.y;
            bmv.y = zy > c.mvmin.y ? zy : c.mvmin.y;
        }
    }
    hex_search<Px, 4, 1>(c, bmv, bcost, 16);
    int bx, by;
    if (bprecost < bcost) { bx = pmvx; by = pmvy; bcost = bprecost; }
    else { bx = bmv.x * 4; by = bmv.y * 4; }
    if (!bcost)
        bcost = c.mvcost_q(bx, by);
    else
    {
        // each round scores its four directions (workload[1]: 4 half-pel, 4 quarter-pel) together, then picks in order
        int bdir = 0, cs[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int qx = bx + sSquare1(i + 1).x * 2, qy = by + sSquare1(i + 1).y * 2;
            cs[i] = L.qpel_cost(pp, qx, qy, false) + c.mvcost_q(qx, qy);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) pin_value(cs[i]);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int qy = by + sSquare1(i + 1).y * 2;
            if ((qy < qminy) | (qy > qmaxy)) continue;
            if (cs[i] < bcost) { bcost = cs[i]; bdir = i + 1; }
        }
        bx += sSquare1(bdir).x * 2; by += sSquare1(bdir).y * 2;
        bcost = L.qpel_cost(pp, bx, by, true) + c.mvcost_q(bx, by);
        bdir = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int qx = bx + sSquare1(i + 1).x, qy = by + sSquare1(i + 1).y;
            cs[i] = L.qpel_cost(pp, qx, qy, true) + c.mvcost_q(qx, qy);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) pin_value(cs[i]);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int qy = by + sSquare1(i + 1).y;
            if ((qy < qminy) | (qy > qmaxy)) continue;
            if (cs[i] < bcost) { bcost = cs[i]; bdir = i + 1; }
        }
        bx += sSquare1(bdir).x; by += sSquare1(bdir).y;
    }
    outx = bx; outy = by;
    return bcost;
}

// SO (round 6, "search only"): a B picture's two lists are searched independently of each other (each list's predictors are that list's finished vectors), yet the walk above
// ran them one after the other inside every lock-step, followed by the bi-directional candidates: a step of a (p0, b, p1) estimate that searches both lists cost two HEX
// searches + four predictions + two SATDs.  The SO form walks ONE list (blockIdx.y names which of the searched lists) and writes only that list's vectors and costs; the
// lists run side by side on their own compute units and the dependency-free kernel below finishes the estimate (candidates, AQ weighting, sums) in one short launch.
template <typename Px, bool BIDIR, bool SPLIT, bool SO = false>
__global__ void __launch_bounds__(SPLIT ? 512 : 1024) lowres_cost_kernel(LowresCostArgs g)
{
    const int pairIdx = SPLIT ? (int)blockIdx.x / g.K : (int)blockIdx.x;
    const int wk = SPLIT ? (int)blockIdx.x - pairIdx * g.K : 0;                  // this workgroup's band (0 = the bottom rows)
    const x265hip_lowres_cost_pair pr = g.pairs[pairIdx];
    const uint8_t* cur = (const uint8_t*)pr.cur;
    unsigned long long* mvsL[2] = { (unsigned long long*)pr.mvs, (unsigned long long*)pr.mvs1 };
    int32_t* mvCostsL[2] = { pr.mv_costs, pr.mv_costs1 };
    constexpr int BPP = sizeof(Px);
    constexpr uint32_t kBias = 1u << 30;
    constexpr int NL = SO ? 1 : (BIDIR ? 2 : 1);
    // SO: the list this workgroup walks - the blockIdx.y-th of the lists the estimate searches
    const int soList = SO ? ((pr.do_search[0] && blockIdx.y == 0) ? 0 : 1) : 0;
    const int Q = blockDim.x >> 2;                    // rows in flight: one quad each
    const int q = threadIdx.x >> 2, l = threadIdx.x & 3;
    const int tx = l & 1, ty = l >> 1;
    const int W = g.W, H = g.H;
    const int row0 = SPLIT ? wk * g.R : 0;                                       // first block row (from the bottom) of this band
    const int rowsHere = SPLIT ? (g.R < H - row0 ? g.R : H - row0) : H;
    const int tBegin = 2 * row0, tEnd = SPLIT ? 2 * (row0 + rowsHere - 1) + W : W + 2 * (H - 1);
    unsigned long long* const syncP = SPLIT ? g.sync + (size_t)pairIdx * split_sync_words(g.K, W) : nullptr;
    __shared__ long long sFrame[3];
    if (threadIdx.x < 3) sFrame[threadIdx.x] = 0;
    __syncthreads();
    long long costEst = 0, costEstAq = 0;
    int intraMbs = 0, rowSatd = 0;
    int rightx[2] = { 0, 0 }, righty[2] = { 0, 0 };   // per list: the finished block to the right (this quad's previous result)
    LowresPu<Px> L;
    const PhasePlanes pp0 = { (const uint8_t*)pr.ref[0] - kBias, (const uint8_t*)pr.ref[1] - kBias, (const uint8_t*)pr.ref[2] - kBias, (const uint8_t*)pr.ref[3] - kBias };
    const PhasePlanes pp1 = BIDIR ? PhasePlanes{ (const uint8_t*)pr.ref1[0] - kBias, (const uint8_t*)pr.ref1[1] - kBias, (const uint8_t*)pr.ref1[2] - kBias, (const uint8_t*)pr.ref1[3] - kBias } : pp0;
    // --weightp: the bi-directional candidates keep the unweighted list-0 planes (slicetype.cpp:3328)
    const PhasePlanes ppB = (BIDIR && pr.ref_bi[0]) ? PhasePlanes{ (const uint8_t*)pr.ref_bi[0] - kBias, (const uint8_t*)pr.ref_bi[1] - kBias, (const uint8_t*)pr.ref_bi[2] - kBias, (const uint8_t*)pr.ref_bi[3] - kBias } : pp0;
    L.c.strideB = g.strideB; L.c.depth = g.depth; L.c.cost = g.cost;
    L.c.have[0] = true;
    for (int t = tBegin; t < tEnd; t++)
    {
        // quad q walks rows q, q + Q, q + 2Q ... (counted from the bottom); W <= 2Q keeps at most one of them active per step
        // (SPLIT: quad q owns row row0 + q of its band)
        int ry = -1, cuX = 0;
        if (SPLIT)
        {
            const int r = row0 + q, dx = t - 2 * r;
            if (q < rowsHere && dx >= 0 && dx < W) { ry = r; cuX = W - 1 - dx; }
        }
        else if (t >= 2 * q)
        {
            const int k = (t - 2 * q) / (2 * Q), r = q + k * Q, dx = t - 2 * r;
            if (r < H && dx < W) { ry = r; cuX = W - 1 - dx; }
        }
        if (ry >= 0)
        {
            const int cuY = H - 1 - ry, cuXY = cuX + cuY * W;
            const uint32_t pel = (uint32_t)((cuY * 8 + ty * 4) * g.strideB + (cuX * 8 + tx * 4) * BPP);
            L.c.refOrg[0] = kBias + pel;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int d = 0; d < BPP; d++) L.c.src[0][r][d] = ld_u32(cur + (pel + (uint32_t)(r * g.strideB + 4 * d)));
            L.c.mvmin.x = -cuX * 8 - 8; L.c.mvmin.y = -cuY * 8 - 8;
            L.c.mvmax.x = (W - cuX - 1) * 8 + 8; L.c.mvmax.y = (H - cuY - 1) * 8 + 8;
            int bcost = 1 << 28, listused = 0;                                     // MotionEstimate::COST_MAX
            int lmx[2] = { 0, 0 }, lmy[2] = { 0, 0 };
#pragma unroll
            for (int lk = 0; lk < NL; lk++)
            {
                const int li = SO ? soList : lk;
                const PhasePlanes pp = li ? pp1 : pp0;
                unsigned long long* mvA = li ? mvsL[1] : mvsL[0];
                int32_t* mcA = li ? mvCostsL[1] : mvCostsL[0];
                int fencCost, qx, qy;
                if (!SO && !pr.do_search[li])
                {
                    // estimateFrameCost's bDoSearch == false: this list was searched by an earlier estimate, its results stand
                    fencCost = mcA[cuXY];
                    const unsigned long long v = mvA[cuXY];
                    qx = (int)(uint32_t)v; qy = (int)(uint32_t)(v >> 32);
                }
                else
                {
                    L.c.base = pp.p0;
                    // reverse-order mv prediction (:3266-3301): right, below, below-left, below-right; the cheapest SATD wins
                    // (strict <).  The (up to) four finished neighbours are fetched and scored together; invalid ones score a
                    // dummy position that the selection below skips.
                    const bool vR = cuX < W - 1, vB = ry > 0, vBL = vB && cuX > 0, vBR = vB && cuX < W - 1;
                    auto finished = [&](bool valid, int idx, int& mx, int& my)
                    {
                        if (SPLIT && q == 0 && wk > 0)
                        {
                            // the row below is the top row of the band below: wait for its published word (qx, qy in 24 bits each under the tag)
                            mx = my = 0;
                            if (valid)
                            {
                                const unsigned long long* w = syncP + ((size_t)(li * g.K + wk - 1) * W + (idx - (cuXY + W) + cuX));
                                unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                while ((unsigned)(v >> 48) != kSplitTag)
                                {
                                    __builtin_amdgcn_s_sleep(1);
                                    v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                mx = ((int)((uint32_t)v << 8)) >> 8; my = ((int)((uint32_t)(v >> 24) << 8)) >> 8;
                            }
                            return;
                        }
                        const unsigned long long v = __hip_atomic_load(&mvA[valid ? idx : cuXY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        mx = valid ? (int)(uint32_t)v : 0; my = valid ? (int)(uint32_t)(v >> 32) : 0;
                    };
                    int cx[4], cy[4], cc[4];
                    cx[0] = vR ? rightx[SO ? 0 : li] : 0; cy[0] = vR ? righty[SO ? 0 : li] : 0;
                    finished(vB, cuXY + W, cx[1], cy[1]);
                    finished(vBL, cuXY + W - 1, cx[2], cy[2]);
                    finished(vBR, cuXY + W + 1, cx[3], cy[3]);
#pragma unroll
                    for (int i = 0; i < 4; i++) cc[i] = L.qpel_cost(pp, cx[i], cy[i], true);
#pragma unroll
                    for (int i = 0; i < 4; i++) pin_value(cc[i]);
                    const bool cv[4] = { vR, vB, vBL, vBR };
                    int mvpx = 0, mvpy = 0, mvpcost = 1 << 28, skipCost = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (cv[i])
                        {
                            if (cc[i] < mvpcost) { mvpcost = cc[i]; mvpx = cx[i]; mvpy = cy[i]; }
                            // :3297-3299: while the best predictor so far is the zero mv, remember the cost just measured
                            if (BIDIR && !(mvpx | mvpy)) skipCost = cc[i];
                        }
                    L.c.mvpx = mvpx; L.c.mvpy = mvpy;
                    fencCost = lowres_motion_estimate<Px>(L, pp, qx, qy);
                    if (BIDIR && skipCost < 64 && skipCost < fencCost) { fencCost = skipCost; qx = qy = 0; }       // :3311-3315
                    rightx[SO ? 0 : li] = qx; righty[SO ? 0 : li] = qy;
                    if (l == 0)
                    {
                        __hip_atomic_store(&mvA[cuXY], (unsigned long long)(uint32_t)qx | ((unsigned long long)(uint32_t)qy << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        mcA[cuXY] = fencCost;
                        if (SPLIT && q == rowsHere - 1 && wk < g.K - 1)                      // the band above reads this row through L2
                            __hip_atomic_store(syncP + ((size_t)(li * g.K + wk) * W + cuX),
                                               ((unsigned long long)kSplitTag << 48) | ((unsigned long long)((uint32_t)qy & 0xffffffu) << 24) | (unsigned long long)((uint32_t)qx & 0xffffffu),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                lmx[SO ? 0 : li] = qx; lmy[SO ? 0 : li] = qy;
                if (fencCost < bcost) { bcost = fencCost; listused = li + 1; }
            }
            if (SO) { }                                                            // the candidates and the sums are the flat kernel's
            else if (BIDIR)
            {
                // avg(l0-mv, l1-mv), then the co-located average (:3322-3343)
                int p0[4][4], p1[4][4];
                L.predict(ppB, lmx[0], lmy[0], p0);
                L.predict(pp1, lmx[1], lmy[1], p1);
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) p0[y][x] = (p0[y][x] + p1[y][x] + 1) >> 1;
                int bicost = L.score(p0, true);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                L.predict(ppB, 0, 0, p0);
                L.predict(pp1, 0, 0, p1);
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) p0[y][x] = (p0[y][x] + p1[y][x] + 1) >> 1;
                bicost = L.score(p0, true);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                bcost += 4;                                                        // lowresPenalty
            }
            else
            {
                bcost += 4;
                const int ic = pr.intra_cost[cuXY];
                if (ic < bcost) { bcost = ic; listused = 0; }
            }
            if (!SO)
            {
            const bool scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
            const int bcostAq = (scored && pr.inv_qscale) ? ((bcost * pr.inv_qscale[cuXY] + 128) >> 8) : bcost;
            if (scored) { costEst += bcost; costEstAq += bcostAq; intraMbs += (!BIDIR && !listused); }
            if (cuX == W - 1) rowSatd = 0;
            rowSatd += bcostAq;
            if (l == 0)
            {
                pr.lowres_costs[cuXY] = (uint16_t)((bcost < 0x3fff ? bcost : 0x3fff) | (listused << 14));
                if (cuX == 0) pr.row_satds[cuY] = rowSatd;
            }
            }
        }
        // the exchange stays inside one workgroup (one CU, one L1 / L2): the barrier's workgroup-scope fence is all it needs - an
        // agent-scope fence would write the L2 back every step
        __syncthreads();
    }
    if (SO) return;
    if (l == 0)
    {
        atomicAdd((unsigned long long*)&sFrame[0], (unsigned long long)costEst);
        atomicAdd((unsigned long long*)&sFrame[1], (unsigned long long)costEstAq);
        atomicAdd((unsigned long long*)&sFrame[2], (unsigned long long)intraMbs);
    }
    __syncthreads();
    if (SPLIT)
    {
        // every band adds its sums; the band that arrives last holds all of them and writes the picture's totals
        if (threadIdx.x == 0)
        {
            unsigned long long* acc = syncP + (size_t)2 * g.K * W;
            for (int i = 0; i < 3; i++) __hip_atomic_fetch_add(&acc[i], (unsigned long long)sFrame[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long before = __hip_atomic_fetch_add(&acc[3], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before == (unsigned long long)(g.K - 1))
            {
                long long f[3];
                for (int i = 0; i < 3; i++) f[i] = (long long)__hip_atomic_load(&acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int i = 0; i < 3; i++) pr.frame[i] = f[i];
                pr.frame[3] = BIDIR ? f[0] * 100 / (130 + g.bFrameBias) : f[0];
            }
        }
        return;
    }
    if (threadIdx.x < 3) pr.frame[threadIdx.x] = sFrame[threadIdx.x];
    if (threadIdx.x == 0) pr.frame[3] = BIDIR ? sFrame[0] * 100 / (130 + g.bFrameBias) : sFrame[0];      // estimateFrameCost's score (:3201-3204)
}

// FLAT (round 6).  An estimate whose lists were BOTH searched by earlier estimates (estimateFrameCost's bDoSearch false for every list it reads, slicetype.cpp:3126-3127:
// a third of the triples the slice-type decision scores - every (p0, b, p1) after the first with the same (b, p0) and (b, p1)) has no dependency between blocks at all:
// the vectors are inputs, what is left per block is the bi-directional candidates / the intra comparison, the AQ weighting and the sums.  The dependent walk still took its
// W + 2 H lock-steps for it (4K: 510 steps, 5 ms on one estimate the lookahead WAITS for - the real encode's fps follows that latency, profiles/r06_lookahead_bound.txt).
// Here a wavefront owns a block row, a quad a segment of ceil(W / 16) consecutive blocks; row sums meet by shuffles, frame sums in the split form's accumulators.
template <typename Px, bool BIDIR>
__global__ void __launch_bounds__(256) lowres_cost_flat_kernel(LowresCostArgs g)
{
    const int pairIdx = (int)blockIdx.y;
    const x265hip_lowres_cost_pair pr = g.pairs[pairIdx];
    const uint8_t* cur = (const uint8_t*)pr.cur;
    const unsigned long long* mvsL[2] = { (const unsigned long long*)pr.mvs, (const unsigned long long*)pr.mvs1 };
    const int32_t* mvCostsL[2] = { pr.mv_costs, pr.mv_costs1 };
    constexpr int BPP = sizeof(Px);
    constexpr uint32_t kBias = 1u << 30;
    constexpr int NL = BIDIR ? 2 : 1;
    const int lane = threadIdx.x & 63, q = lane >> 2, l = lane & 3;
    const int tx = l & 1, ty = l >> 1;
    const int W = g.W, H = g.H;
    const int cuY = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int seg = (W + 15) >> 4, x0 = q * seg, x1 = min(W, x0 + seg);
    unsigned long long* const acc = g.sync + (size_t)pairIdx * 4;
    __shared__ long long sFrame[3];
    if (threadIdx.x < 3) sFrame[threadIdx.x] = 0;
    __syncthreads();
    long long costEst = 0, costEstAq = 0;
    int intraMbs = 0, rowSatd = 0;
    LowresPu<Px> L;
    const PhasePlanes pp0 = { (const uint8_t*)pr.ref[0] - kBias, (const uint8_t*)pr.ref[1] - kBias, (const uint8_t*)pr.ref[2] - kBias, (const uint8_t*)pr.ref[3] - kBias };
    const PhasePlanes pp1 = BIDIR ? PhasePlanes{ (const uint8_t*)pr.ref1[0] - kBias, (const uint8_t*)pr.ref1[1] - kBias, (const uint8_t*)pr.ref1[2] - kBias, (const uint8_t*)pr.ref1[3] - kBias } : pp0;
    const PhasePlanes ppB = (BIDIR && pr.ref_bi[0]) ? PhasePlanes{ (const uint8_t*)pr.ref_bi[0] - kBias, (const uint8_t*)pr.ref_bi[1] - kBias, (const uint8_t*)pr.ref_bi[2] - kBias, (const uint8_t*)pr.ref_bi[3] - kBias } : pp0;
    L.c.strideB = g.strideB; L.c.depth = g.depth; L.c.cost = g.cost;
    L.c.have[0] = true;
    if (cuY < H)
        for (int cuX = x1 - 1; cuX >= x0; cuX--)
        {
            const int cuXY = cuX + cuY * W;
            const uint32_t pel = (uint32_t)((cuY * 8 + ty * 4) * g.strideB + (cuX * 8 + tx * 4) * BPP);
            L.c.refOrg[0] = kBias + pel;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int d = 0; d < BPP; d++) L.c.src[0][r][d] = ld_u32(cur + (pel + (uint32_t)(r * g.strideB + 4 * d)));
            L.c.mvmin.x = -cuX * 8 - 8; L.c.mvmin.y = -cuY * 8 - 8;
            L.c.mvmax.x = (W - cuX - 1) * 8 + 8; L.c.mvmax.y = (H - cuY - 1) * 8 + 8;
            int bcost = 1 << 28, listused = 0;
            int lmx[2] = { 0, 0 }, lmy[2] = { 0, 0 };
#pragma unroll
            for (int li = 0; li < NL; li++)
            {
                const int fencCost = mvCostsL[li][cuXY];
                const unsigned long long v = mvsL[li][cuXY];
                lmx[li] = (int)(uint32_t)v; lmy[li] = (int)(uint32_t)(v >> 32);
                if (fencCost < bcost) { bcost = fencCost; listused = li + 1; }
            }
            if (BIDIR)
            {
                int p0[4][4], p1[4][4];
                L.predict(ppB, lmx[0], lmy[0], p0);
                L.predict(pp1, lmx[1], lmy[1], p1);
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) p0[y][x] = (p0[y][x] + p1[y][x] + 1) >> 1;
                int bicost = L.score(p0, true);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                L.predict(ppB, 0, 0, p0);
                L.predict(pp1, 0, 0, p1);
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) p0[y][x] = (p0[y][x] + p1[y][x] + 1) >> 1;
                bicost = L.score(p0, true);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                bcost += 4;
            }
            else
            {
                bcost += 4;
                const int ic = pr.intra_cost[cuXY];
                if (ic < bcost) { bcost = ic; listused = 0; }
            }
            const bool scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
            const int bcostAq = (scored && pr.inv_qscale) ? ((bcost * pr.inv_qscale[cuXY] + 128) >> 8) : bcost;
            if (scored) { costEst += bcost; costEstAq += bcostAq; intraMbs += (!BIDIR && !listused); }
            rowSatd += bcostAq;
            if (l == 0) pr.lowres_costs[cuXY] = (uint16_t)((bcost < 0x3fff ? bcost : 0x3fff) | (listused << 14));
        }
    // the row's sum over its 16 quads (every lane of a quad holds the quad's value), then the frame sums of the workgroup's four rows
    for (int o = 4; o < 64; o <<= 1) rowSatd += __shfl_xor(rowSatd, o);
    if (lane == 0 && cuY < H) pr.row_satds[cuY] = rowSatd;
    auto xor64 = [](long long v, int o) { return (long long)(((unsigned long long)(uint32_t)__shfl_xor((int)(v >> 32), o) << 32) | (uint32_t)__shfl_xor((int)v, o)); };
    for (int o = 4; o < 64; o <<= 1) { costEst += xor64(costEst, o); costEstAq += xor64(costEstAq, o); intraMbs += __shfl_xor(intraMbs, o); }
    if (lane == 0)
    {
        atomicAdd((unsigned long long*)&sFrame[0], (unsigned long long)costEst);
        atomicAdd((unsigned long long*)&sFrame[1], (unsigned long long)costEstAq);
        atomicAdd((unsigned long long*)&sFrame[2], (unsigned long long)intraMbs);
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int i = 0; i < 3; i++) __hip_atomic_fetch_add(&acc[i], (unsigned long long)sFrame[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long before = __hip_atomic_fetch_add(&acc[3], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == (unsigned long long)(gridDim.x - 1))
        {
            long long f[3];
            for (int i = 0; i < 3; i++) f[i] = (long long)__hip_atomic_load(&acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < 3; i++) pr.frame[i] = f[i];
            pr.frame[3] = BIDIR ? f[0] * 100 / (130 + g.bFrameBias) : f[0];
        }
    }
}

} // namespace x265hip

using namespace x265hip;

static std::atomic<uint64_t> g_lrcFlat{0}, g_lrcWalk{0}, g_lrcSplit{0}, g_lrcSo{0};          // (g_lrcSo: the split launches that walked their lists side by side)
// the A/B switches, read once when the library is loaded; x265hip_lowres_cost_env_refresh() reads them again (tests and A/B tools that change them in a running process)
static std::atomic<int> g_lrcEnvSplit{0}, g_lrcEnvSoOff{0};
extern "C" void x265hip_lowres_cost_env_refresh(void)
{
    const char* e = getenv("X265HIP_LOWRES_COST_SPLIT");
    int k = e ? atoi(e) : 0;
    g_lrcEnvSplit.store(e ? (k < 1 ? 1 : k) : 0, std::memory_order_relaxed);
    g_lrcEnvSoOff.store(getenv("X265HIP_LOWRES_COST_SO_OFF") ? 1 : 0, std::memory_order_relaxed);
}
namespace { struct LrcEnvInit { LrcEnvInit() { x265hip_lowres_cost_env_refresh(); } } g_lrcEnvInit; }
extern "C" void x265hip_lowres_cost_launch_counts(uint64_t out[3])
{
    out[0] = g_lrcFlat.load(); out[1] = g_lrcWalk.load(); out[2] = g_lrcSplit.load();
}

extern "C" int x265hip_lowres_cost(const x265hip_lowres_cost_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->cost_q || !p->pairs) { set_error("lowres_cost: NULL operand"); return X265HIP_EINVAL; }
    if (p->npairs < 0) { set_error("lowres_cost: npairs %d", p->npairs); return X265HIP_EINVAL; }
    if (p->npairs == 0) return 0;
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("lowres_cost: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width_in_cu <= 0 || p->height_in_cu <= 0) { set_error("lowres_cost: empty picture"); return X265HIP_EINVAL; }
    bool bidir = false;
    if (p->pairs_on_device) bidir = (p->pairs_on_device & 3) == 2;
    else
    for (int i = 0; i < p->npairs; i++)
    {
        const x265hip_lowres_cost_pair& q = p->pairs[i];
        if (!q.cur || !q.ref[0] || !q.ref[1] || !q.ref[2] || !q.ref[3] || !q.intra_cost || !q.mvs || !q.mv_costs || !q.lowres_costs || !q.row_satds || !q.frame)
        { set_error("lowres_cost: NULL operand in pair %d", i); return X265HIP_EINVAL; }
        const bool b = q.ref1[0] != nullptr;
        if (i == 0) bidir = b;
        if (b != bidir) { set_error("lowres_cost: pair %d mixes P and B pictures in one call", i); return X265HIP_EINVAL; }
        if (b && (!q.ref1[1] || !q.ref1[2] || !q.ref1[3] || !q.mvs1 || !q.mv_costs1)) { set_error("lowres_cost: NULL list-1 operand in pair %d", i); return X265HIP_EINVAL; }
        if (q.ref_bi[0] && (!b || !q.ref_bi[1] || !q.ref_bi[2] || !q.ref_bi[3])) { set_error("lowres_cost: ref_bi needs a B picture and four planes (pair %d)", i); return X265HIP_EINVAL; }
        if ((((uintptr_t)q.mvs) & 7) || (b && (((uintptr_t)q.mvs1) & 7))) { set_error("lowres_cost: mvs of pair %d must be 8-byte aligned", i); return X265HIP_EINVAL; }
    }
    // FLAT: no list of any pair needs a search - no dependency, no lock-step walk.  Host pair tables are read here; a DEVICE table says so itself (pairs_on_device | 4)
    bool flat = getenv("X265HIP_LOWRES_COST_FLAT_OFF") == nullptr;
    if (flat && p->pairs_on_device) flat = (p->pairs_on_device & 4) != 0;
    else if (flat)
        for (int i = 0; i < p->npairs; i++)
            flat &= !p->pairs[i].do_search[0] && (!bidir || !p->pairs[i].do_search[1]);
    if (flat)
    {
        const int bppF = p->depth == 8 ? 1 : 2;
        hipStream_t sF = (hipStream_t)stream;
        x265hip_lowres_cost_pair* dp = nullptr;
        const size_t nb = sizeof(x265hip_lowres_cost_pair) * (size_t)p->npairs;
        if (p->pairs_on_device) dp = const_cast<x265hip_lowres_cost_pair*>(p->pairs);
        else
        {
            X265HIP_TRY(hipMallocAsync((void**)&dp, nb, sF));
            X265HIP_TRY(hipMemcpyAsync(dp, p->pairs, nb, hipMemcpyHostToDevice, sF));
        }
        LowresCostArgs a;
        a.pairs = dp; a.strideB = (int)(p->stride * bppF); a.W = p->width_in_cu; a.H = p->height_in_cu; a.depth = p->depth;
        a.cost = p->cost_q + p->qoff; a.bFrameBias = p->bframe_bias; a.K = 1; a.R = p->height_in_cu;
        {
            std::unique_lock<std::mutex> seqF = stream_sequence_lock(sF);          // clear + launch are one sequence per stream
            const size_t sb = 4 * 8 * (size_t)p->npairs;
            a.sync = (unsigned long long*)stream_scratch(sF, 1, sb);
            if (!a.sync) return X265HIP_ENODEV;
            X265HIP_TRY(hipMemsetAsync(a.sync, 0, sb, sF));
            const dim3 gridF((p->height_in_cu + 3) / 4, p->npairs), blockF(256);
            if (bppF == 1 && !bidir) hipLaunchKernelGGL((lowres_cost_flat_kernel<uint8_t, false>), gridF, blockF, 0, sF, a);
            else if (bppF == 1) hipLaunchKernelGGL((lowres_cost_flat_kernel<uint8_t, true>), gridF, blockF, 0, sF, a);
            else if (!bidir) hipLaunchKernelGGL((lowres_cost_flat_kernel<uint16_t, false>), gridF, blockF, 0, sF, a);
            else hipLaunchKernelGGL((lowres_cost_flat_kernel<uint16_t, true>), gridF, blockF, 0, sF, a);
        }
        X265HIP_TRY(hipGetLastError());
        if (!p->pairs_on_device) X265HIP_TRY(hipFreeAsync(dp, sF));
        g_lrcFlat.fetch_add(1, std::memory_order_relaxed);
        return 0;
    }
    // One estimate on its own (the lookahead seam) is split into bands of ~8 block rows (at most 16 bands: 4K 8.9 -> 5.3 ms with 16, 5.8 with 9; 1080p 3.27 ->
    // 2.78 ms with 8, 2.88 with 5 - profiles/r03_lowres_cost_counters.txt), a workgroup = a compute unit each; batches keep one
    // workgroup per picture (the chip is full anyway).  X265HIP_LOWRES_COST_SPLIT=<bands> forces the number (1 = never split; tests, A/B runs).
    int K = 1;
    // A split launch's bands SPIN on the band below them (tagged words through L2); forward progress rests on the lower bands being
    // dispatched first, which an ordinary launch does not promise once the chip is full of other spinners.  At most kMaxSplitInFlight split
    // launches are in flight per process (counted at enqueue, released by a host callback behind the launch); above that an estimate runs
    // as one workgroup per picture - slower, never stuck (round-3 advisor).  The A/B switch is read once.
    const int splitEnvBands = g_lrcEnvSplit.load(std::memory_order_relaxed);          // 0: not set
    const bool splitEnv = splitEnvBands != 0;
    static std::atomic<int> splitInFlight{0};
    constexpr int kMaxSplitInFlight = 8;                                // 8 x 16 bands = half the CUs at most wait on a neighbour
    bool counted = false;
    {
        // a launch recorded into a HIP graph is replayed without passing here again: the release callback would run once per replay against ONE
        // increment and the cap below would stop meaning anything (round-4 advisor) - under capture the estimate is not split
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (stream) (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
        if (splitEnv) K = splitEnvBands;
        else if (cap == hipStreamCaptureStatusActive) K = 1;
        else if (p->height_in_cu >= 32 && p->npairs <= 4)
        {
            if (splitInFlight.fetch_add(1) < kMaxSplitInFlight) { K = (p->height_in_cu + 7) / 8; counted = true; }
            else splitInFlight.fetch_sub(1);
        }
        if (K > (splitEnv ? 64 : 16)) K = splitEnv ? 64 : 16;          // (the switch may ask for more bands than the library's own choice ever makes: A/B runs)
        if (K > p->height_in_cu) K = p->height_in_cu;
        if (K < 1) K = 1;
        if (K > 1 && (p->height_in_cu + K - 1) / K > 128) K = (p->height_in_cu + 127) / 128;      // a band is at most 128 rows = 512 threads
    }
    const int R = (p->height_in_cu + K - 1) / K;
    K = (p->height_in_cu + R - 1) / R;                                 // no empty band
    const bool split = K > 1;
    int quads = ((split ? R : p->height_in_cu) + 15) & ~15;
    if (quads > 256) quads = 256;
    if (!split && p->width_in_cu > 2 * quads) { set_error("lowres_cost: %d blocks per row need more than %d rows in flight", p->width_in_cu, quads); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    hipStream_t s = (hipStream_t)stream;
    // the pair table travels in stream order: allocated, filled, used and released on `s`
    x265hip_lowres_cost_pair* dpairs = nullptr;
    const size_t bytes = sizeof(x265hip_lowres_cost_pair) * (size_t)p->npairs;
    if (p->pairs_on_device) dpairs = const_cast<x265hip_lowres_cost_pair*>(p->pairs);
    else
    {
        X265HIP_TRY(hipMallocAsync((void**)&dpairs, bytes, s));
        X265HIP_TRY(hipMemcpyAsync(dpairs, p->pairs, bytes, hipMemcpyHostToDevice, s));
    }
    LowresCostArgs a;
    a.pairs = dpairs;
    a.strideB = (int)(p->stride * bpp);
    a.W = p->width_in_cu; a.H = p->height_in_cu; a.depth = p->depth;
    a.cost = p->cost_q + p->qoff;
    a.bFrameBias = p->bframe_bias;
    a.K = K; a.R = R; a.sync = nullptr;
    // SO: a split B estimate walks its searched lists side by side (one search per lock-step instead of two searches + the bi-directional candidates) and
    // leaves the rest to the dependency-free kernel.  The lists an estimate searches: bits 8 / 16 of pairs_on_device, or the host table itself (all pairs alike).
    int soLists = 0;
    const bool soOff = g_lrcEnvSoOff.load(std::memory_order_relaxed) != 0;
    if (split && bidir && !soOff)
    {
        if (p->pairs_on_device) soLists = (p->pairs_on_device >> 3) & 3;
        else
        {
            soLists = (p->pairs[0].do_search[0] ? 1 : 0) | (p->pairs[0].do_search[1] ? 2 : 0);
            for (int i = 1; i < p->npairs; i++)
                if (soLists != ((p->pairs[i].do_search[0] ? 1 : 0) | (p->pairs[i].do_search[1] ? 2 : 0))) soLists = 0;
        }
    }
    std::unique_lock<std::mutex> seq;                 // split: clear + launch are one sequence per stream (round-3 advisor)
    unsigned long long* flatAcc = nullptr;
    if (split)
    {
        seq = stream_sequence_lock(s);
        const size_t words = split_sync_words(K, p->width_in_cu) * (size_t)p->npairs;
        const size_t sb = (words + (soLists ? 4 * (size_t)p->npairs : 0)) * 8;
        a.sync = (unsigned long long*)stream_scratch(s, 1, sb);
        if (!a.sync) return X265HIP_ENODEV;
        X265HIP_TRY(hipMemsetAsync(a.sync, 0, sb, s));
        flatAcc = a.sync + words;
    }
    (split ? g_lrcSplit : g_lrcWalk).fetch_add(1, std::memory_order_relaxed);
    const dim3 grid(p->npairs * K), block(quads * 4);
    if (soLists)
    {
        const dim3 gridS(p->npairs * K, soLists == 3 ? 2 : 1);
        if (bpp == 1) hipLaunchKernelGGL((lowres_cost_kernel<uint8_t, true, true, true>), gridS, block, 0, s, a);
        else hipLaunchKernelGGL((lowres_cost_kernel<uint16_t, true, true, true>), gridS, block, 0, s, a);
        LowresCostArgs f = a;
        f.K = 1; f.R = p->height_in_cu; f.sync = flatAcc;
        const dim3 gridF((p->height_in_cu + 3) / 4, p->npairs), blockF(256);
        if (bpp == 1) hipLaunchKernelGGL((lowres_cost_flat_kernel<uint8_t, true>), gridF, blockF, 0, s, f);
        else hipLaunchKernelGGL((lowres_cost_flat_kernel<uint16_t, true>), gridF, blockF, 0, s, f);
        g_lrcSo.fetch_add(1, std::memory_order_relaxed);
    }
    else
    {
#define LRC_GO(PX, BI) do { if (split) hipLaunchKernelGGL((lowres_cost_kernel<PX, BI, true>), grid, block, 0, s, a); \
                            else hipLaunchKernelGGL((lowres_cost_kernel<PX, BI, false>), grid, block, 0, s, a); } while (0)
    if (bpp == 1 && !bidir) LRC_GO(uint8_t, false);
    else if (bpp == 1) LRC_GO(uint8_t, true);
    else if (!bidir) LRC_GO(uint16_t, false);
    else LRC_GO(uint16_t, true);
#undef LRC_GO
    }
    if (counted)
    {
        if (split && hipLaunchHostFunc(s, [](void* c) { static_cast<std::atomic<int>*>(c)->fetch_sub(1); }, &splitInFlight) == hipSuccess) counted = false;
        if (counted) splitInFlight.fetch_sub(1);                       // not split after all (or no callback could be queued): nothing to wait for
    }
    X265HIP_TRY(hipGetLastError());
    if (!p->pairs_on_device) X265HIP_TRY(hipFreeAsync(dpairs, s));
    return 0;
}
