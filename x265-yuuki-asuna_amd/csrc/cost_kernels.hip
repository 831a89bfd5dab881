// cost_kernels.hip - SUB-SAMPLE COST TABLES: the values MotionEstimate::motionEstimate's refinement asks subpelCompare for
// (motion.cpp:1456-1561 -> :1571-1664), computed for every PU of a CTU around the integer vectors its search can end on.
//
//   x265hip_cost_candidates   SAD rasters of the 85 square PUs (x265hip_me_fullsearch, X265HIP_SURF_I32) -> per PU SHAPE of the list
//                             (squares, 2NxN / Nx2N, the AMP parts made of 8x8 blocks: primitives.h:41-55) the 1 or 2 displacements of
//                             smallest SAD.  SAD is additive, so a rectangle's raster is the sum of its squares' rasters (what the
//                             reference's pu[LUMA_64x32].sad returns on the same samples).  One wavefront per PU walks the raster,
//                             64 displacements per step; the two smallest (sad << 32 | raster index) keys survive a shuffle reduction:
//                             ties resolve to the smaller raster index, the reference's scan order with its strict '<'.
//   x265hip_cost_tables       (cost_tables_kernel; the shared-tile kernel below runs first) per (CTU, PU, candidate) one wavefront: for every position of the refinement's position set the SATD of the
//                             source block against the block of the reference's fractional-phase plane the position selects
//                             (x265hip_phase_planes holds exactly the samples luma_hpp / luma_vpp / luma_hvpp and the chroma filter_hpp /
//                             filter_vpp / filter_hps + filter_vsp calls of subpelCompare would write), luma or luma + Cb + Cr.
//                             SATD is evaluated 4x4 tile by 4x4 tile - satd_4x4 (pixel.cpp:210-236): sum |H d H'| >> 1; every larger
//                             size the reference uses is a sum of 4x4 / 8x4 tiles (:239-297, which tiling per size :1131-1155) and a
//                             tile's absolute sum is even, so any tiling gives the same integer (checked against
//                             the reference's own functions in tests/).  A lane owns one (position, tile) item; items of one position are
//                             summed across the wavefront when a position fills whole wavefronts, through LDS atomics otherwise.
// Both are HBM/L2-bound integer work: a table launch re-reads the CTU's neighbourhood of the phase planes once per PU that covers it
// (10 covering shapes with the rectangles), served by L2; the algorithmic bytes are the planes' CTU neighbourhoods + the records.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace x265hip {

enum { COST_MAX_PU = 209, COST_MAX_POS = 169, COST_MAX_PARTS = 8, COST_WAVES_PER_CTU = 16 };

struct CostPu { uint8_t x, y, w, h; uint8_t nparts; uint8_t part[COST_MAX_PARTS]; uint8_t pad[3]; };   // samples inside the CTU (w, h <= 64); parts = indices into the 85 squares

// ---------------------------------------------------------------- host: the PU list and the position sets
namespace {

int zindex(int ux, int uy)
{
    int z = 0;
    for (int b = 0; b < 3; b++) z |= ((ux >> b) & 1) << (2 * b) | ((uy >> b) & 1) << (2 * b + 1);
    return z;
}

// squares of the surfaces that tile rectangle (px, py, w, h) (multiples of 8), quadtree order
void decompose(int px, int py, int w, int h, int bx, int by, int size, CostPu& pu)
{
    const int x0 = bx > px ? bx : px, x1 = (bx + size < px + w) ? bx + size : px + w;
    const int y0 = by > py ? by : py, y1 = (by + size < py + h) ? by + size : py + h;
    if (x0 >= x1 || y0 >= y1) return;
    if (x1 - x0 == size && y1 - y0 == size)
    {
        static const int base[4] = { 0, 64, 80, 84 };
        const int level = size == 8 ? 0 : size == 16 ? 1 : size == 32 ? 2 : 3;
        if (pu.nparts < COST_MAX_PARTS) pu.part[pu.nparts] = (uint8_t)(base[level] + zindex(bx / size, by / size));
        pu.nparts++;
        return;
    }
    const int hs = size >> 1;
    decompose(px, py, w, h, bx, by, hs, pu); decompose(px, py, w, h, bx + hs, by, hs, pu);
    decompose(px, py, w, h, bx, by + hs, hs, pu); decompose(px, py, w, h, bx + hs, by + hs, hs, pu);
}

struct PuList { CostPu pu[COST_MAX_PU]; };

const PuList& pu_list()
{
    static PuList L;
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        auto add = [&](int x, int y, int w, int h) { CostPu& p = L.pu[n++]; memset(&p, 0, sizeof(p)); p.x = (uint8_t)x; p.y = (uint8_t)y; p.w = (uint8_t)w; p.h = (uint8_t)h; decompose(x, y, w, h, 0, 0, 64, p); };
        auto unz = [](int z, int& ux, int& uy) { ux = uy = 0; for (int b = 0; b < 3; b++) { ux |= ((z >> (2 * b)) & 1) << b; uy |= ((z >> (2 * b + 1)) & 1) << b; } };
        // [0, 85): the squares in the surfaces' order - 64 8x8, 16 16x16, 4 32x32, the 64x64, z-order per level
        for (int size = 8; size <= 64; size <<= 1)
            for (int z = 0; z < (64 / size) * (64 / size); z++) { int ux, uy; unz(z, ux, uy); add(ux * size, uy * size, size, size); }
        // [85, 169): 2NxN and Nx2N of the 16 / 32 / 64 CUs (the 8x8 CU's 8x4 / 4x8 are not unions of 8x8 blocks)
        for (int size = 16; size <= 64; size <<= 1)
            for (int z = 0; z < (64 / size) * (64 / size); z++)
            {
                int ux, uy; unz(z, ux, uy);
                const int x = ux * size, y = uy * size, hs = size >> 1;
                add(x, y, size, hs); add(x, y + hs, size, hs); add(x, y, hs, size); add(x + hs, y, hs, size);
            }
        // [169, 209): the asymmetric partitions of the 32 / 64 CUs: 2NxnU, 2NxnD, nLx2N, nRx2N (the 16 CU's are 4 samples wide)
        for (int size = 32; size <= 64; size <<= 1)
            for (int z = 0; z < (64 / size) * (64 / size); z++)
            {
                int ux, uy; unz(z, ux, uy);
                const int x = ux * size, y = uy * size, q = size >> 2;
                add(x, y, size, q); add(x, y + q, size, size - q);
                add(x, y, size, size - q); add(x, y + size - q, size, q);
                add(x, y, q, size); add(x + q, y, size - q, size);
                add(x, y, size - q, size); add(x + size - q, y, q, size);
            }
    });
    return L;
}

int pu_count(int shapes) { return shapes <= 0 ? 85 : shapes == 1 ? 169 : 209; }

struct PosSet { int n, radius; int8_t xy[COST_MAX_POS][2]; int16_t map[13 * 13]; };

// the quarter-sample offsets a refinement of SubpelWorkload row `subme` (motion.cpp:48-58) can measure from its start vector: hpel_iters
// rounds of square1[1 .. hpel_dirs] * 2 (each round moves to the best neighbour, :1518-1537), then qpel_iters rounds of square1[1 .. qpel_dirs]
bool positions(int subme, PosSet& P)
{
    static const int wl[8][4] = { { 1, 4, 0, 4 }, { 1, 4, 1, 4 }, { 1, 4, 1, 4 }, { 2, 4, 1, 4 }, { 2, 4, 2, 4 }, { 1, 8, 1, 8 }, { 2, 8, 1, 8 }, { 2, 8, 2, 8 } };
    static const int sq[8][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };      // square1[1 .. 8], motion.cpp:44
    if (subme < 0 || subme > 7) return false;
    bool at[13][13] = {}, seen[13][13] = {};
    at[6][6] = seen[6][6] = true;
    for (int phase = 0; phase < 2; phase++)
    {
        const int iters = wl[subme][phase ? 2 : 0], dirs = wl[subme][phase ? 3 : 1], step = phase ? 1 : 2;
        for (int it = 0; it < iters; it++)
        {
            bool next[13][13];
            memcpy(next, at, sizeof(next));
            for (int y = 0; y < 13; y++)
                for (int x = 0; x < 13; x++)
                    if (at[y][x])
                        for (int d = 0; d < dirs; d++)
                        {
                            const int nx = x + sq[d][0] * step, ny = y + sq[d][1] * step;
                            if (nx < 0 || ny < 0 || nx > 12 || ny > 12) return false;
                            next[ny][nx] = seen[ny][nx] = true;
                        }
            memcpy(at, next, sizeof(at));
        }
    }
    P.n = 0; P.radius = 0;
    for (int y = 0; y < 13; y++)
        for (int x = 0; x < 13; x++)
        {
            P.map[y * 13 + x] = -1;
            if (!seen[y][x]) continue;
            if (P.n >= COST_MAX_POS) return false;
            P.map[y * 13 + x] = (int16_t)P.n;
            P.xy[P.n][0] = (int8_t)(x - 6); P.xy[P.n][1] = (int8_t)(y - 6);
            const int r = abs(x - 6) > abs(y - 6) ? abs(x - 6) : abs(y - 6);
            if (r > P.radius) P.radius = r;
            P.n++;
        }
    return true;
}

} // namespace

// ---------------------------------------------------------------- device
__constant__ CostPu kCostPu[COST_MAX_PU];

struct CandArgs
{
    const int32_t* surf; const int16_t* centres; int16_t* cand; const uint16_t* mvCost;
    int window, npu, K;
};

// one workgroup per CTU, 4 wavefronts, wavefront w takes PUs w, w + 4, ...
__global__ void __launch_bounds__(256) cost_cand_kernel(CandArgs a)
{
    const int ctu = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = 2 * a.window + 1, ng = (nc + 3) >> 2, total = nc * nc;
    const int32_t* base = a.surf + (size_t)ctu * nc * ng * 340;
    const int cx = a.centres ? a.centres[2 * ctu] : 0, cy = a.centres ? a.centres[2 * ctu + 1] : 0;
    for (int pu = wave; pu < a.npu; pu += 4)
    {
        const CostPu& P = kCostPu[pu];
        unsigned long long b1 = ~0ull, b2 = ~0ull;
        for (int d = lane; d < total; d += 64)
        {
            const int row = d / nc, col = d - row * nc;
            const int32_t* rec = base + ((size_t)(row * ng + (col >> 2)) * 85) * 4 + (col & 3);
            uint32_t sad = 0;
            for (int i = 0; i < P.nparts; i++) sad += (uint32_t)rec[P.part[i] * 4];
            if (a.mvCost) sad += (uint32_t)a.mvCost[col] + (uint32_t)a.mvCost[row];
            const unsigned long long key = (unsigned long long)sad << 32 | (uint32_t)d;
            if (key < b1) { b2 = b1; b1 = key; } else if (key < b2) b2 = key;
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1)
        {
            const unsigned long long o1 = __shfl_xor(b1, m, 64), o2 = __shfl_xor(b2, m, 64);
            // the two smallest of { b1 <= b2, o1 <= o2 } (keys are distinct: the raster index is part of them)
            const unsigned long long lo = b1 < o1 ? b1 : o1, hi = b1 < o1 ? o1 : b1;
            const unsigned long long second = (b1 < o1 ? b2 : o2) < hi ? (b1 < o1 ? b2 : o2) : hi;
            b1 = lo; b2 = second;
        }
        if (lane < a.K)
        {
            const unsigned long long key = lane ? b2 : b1;
            int16_t* o = a.cand + ((size_t)(ctu * a.npu + pu) * a.K + lane) * 2;
            if (key == ~0ull) { o[0] = (int16_t)-32768; o[1] = 0; }
            else
            {
                const int d = (int)(uint32_t)key, row = d / nc, col = d - row * nc;
                o[0] = (int16_t)(cx + col - a.window); o[1] = (int16_t)(cy + row - a.window);
            }
        }
    }
}

struct TableArgs
{
    const uint8_t* fenc[3]; const uint8_t* ref[3]; const uint8_t* phases[3];
    size_t planeBytes, planeBytesC;
    long strideB, strideCB;
    int marginX, marginY, marginYC;
    int ctusW, ctuRow0, npu, K, npos, recBytes, maxVal;
    int sadCosts, rec2Off;              // sadCosts: a second { base, delta[] } at rec2Off = the costs of the SAD-typed comparisons (luma SAD + chroma SATD)
    const int16_t* cand;
    uint8_t* tables;
    int8_t pos[COST_MAX_POS][2];
};

// 4x4 SATD of (source tile) - (reference tile): satd_4x4, pixel.cpp:210-236, on plain ints
template <typename Px>
__device__ __forceinline__ int satd_tile(const uint8_t* f, long fStride, const uint8_t* r, long rStride)
{
    int d[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        if (sizeof(Px) == 1)
        {
            const uint32_t a = *reinterpret_cast<const uint32_t*>(f + y * fStride), b = ld_u32(r + y * rStride);
#pragma unroll
            for (int x = 0; x < 4; x++) d[y][x] = (int)((a >> (8 * x)) & 255) - (int)((b >> (8 * x)) & 255);
        }
        else
        {
            const uint2 a = *reinterpret_cast<const uint2*>(f + y * fStride);
            const uint32_t b0 = *reinterpret_cast<const u32_align2*>(r + y * rStride), b1 = *reinterpret_cast<const u32_align2*>(r + y * rStride + 4);
            d[y][0] = (int)(a.x & 0xffff) - (int)(b0 & 0xffff); d[y][1] = (int)(a.x >> 16) - (int)(b0 >> 16);
            d[y][2] = (int)(a.y & 0xffff) - (int)(b1 & 0xffff); d[y][3] = (int)(a.y >> 16) - (int)(b1 >> 16);
        }
    }
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        const int a0 = d[y][0] + d[y][1], a1 = d[y][0] - d[y][1], a2 = d[y][2] + d[y][3], a3 = d[y][2] - d[y][3];
        d[y][0] = a0 + a2; d[y][1] = a1 + a3; d[y][2] = a0 - a2; d[y][3] = a1 - a3;
    }
    int sum = 0;
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        const int a0 = d[0][x] + d[1][x], a1 = d[0][x] - d[1][x], a2 = d[2][x] + d[3][x], a3 = d[2][x] - d[3][x];
        sum += abs(a0 + a2) + abs(a1 + a3) + abs(a0 - a2) + abs(a1 - a3);
    }
    return sum >> 1;
}

// the same tile's SAD (sad<4,4>, pixel.cpp:40-55): what a SAD-typed subpelCompare measures on luma
template <typename Px>
__device__ __forceinline__ int sad_tile(const uint8_t* f, long fStride, const uint8_t* r, long rStride)
{
    uint32_t acc = 0;
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        if (sizeof(Px) == 1) acc = sad4_u8(*reinterpret_cast<const uint32_t*>(f + y * fStride), ld_u32(r + y * rStride), acc);
        else
        {
            const uint2 a = *reinterpret_cast<const uint2*>(f + y * fStride);
            acc = sad2_u16(a.x, *reinterpret_cast<const u32_align2*>(r + y * rStride), acc);
            acc = sad2_u16(a.y, *reinterpret_cast<const u32_align2*>(r + y * rStride + 4), acc);
        }
    }
    return (int)acc;
}

// one wavefront per (CTU of the band, PU, candidate)
template <typename Px, bool CHROMA>
__global__ void __launch_bounds__(64) cost_tables_kernel(TableArgs a, const uint8_t* __restrict__ ctuFlags)
{
    constexpr int BPP = sizeof(Px);
    __shared__ uint32_t acc[COST_MAX_POS];
    __shared__ uint32_t acc2[COST_MAX_POS];                   // sadCosts: luma SAD + chroma SATD per position
    const int lane = threadIdx.x;
    // COST_WAVES_PER_CTU single-wavefront workgroups per CTU, each walks the CTU's (PU, candidate) pairs with that stride: few enough workgroups that a launch in which
    // the shared-tile kernel served (nearly) every CTU costs next to nothing, enough of them to fill the chip when it served none
    const int ctuB = blockIdx.x / COST_WAVES_PER_CTU;
    if (ctuFlags && !ctuFlags[ctuB]) return;                 // the shared-tile kernel wrote this CTU's records
    const int ctuX = ctuB % a.ctusW, ctuY = a.ctuRow0 + ctuB / a.ctusW;
    for (int pc = blockIdx.x % COST_WAVES_PER_CTU; pc < a.npu * a.K; pc += COST_WAVES_PER_CTU)
    {
    const int pu = pc / a.K;
    const size_t recIdx = (size_t)ctuB * a.npu * a.K + pc;
    uint8_t* rec = a.tables + recIdx * a.recBytes;
    const int mvx = a.cand[recIdx * 2], mvy = a.cand[recIdx * 2 + 1];
    if (mvx == -32768)
    {
        // "no record": the vector field says so, the rest of the record is zero
        for (int i = lane; i < a.recBytes / 4; i += 64) reinterpret_cast<uint32_t*>(rec)[i] = i ? 0u : 0x00008000u;
        continue;
    }
    for (int i = lane; i < a.npos; i += 64) { acc[i] = 0; acc2[i] = 0; }
    __syncthreads();
    const CostPu P = kCostPu[pu];
    const int tw = P.w >> 2, L = tw * (P.h >> 2), cw = P.w >> 3, C = CHROMA ? cw * (P.h >> 3) : 0, T = L + 2 * C;
    const int total = a.npos * T;
    const bool whole = (T & 63) == 0;                         // a position fills whole wavefront steps: reduce across the lanes
    const int X0 = ctuX * 64 + P.x, Y0 = ctuY * 64 + P.y;
    for (int i0 = 0; i0 < total; i0 += 64)
    {
        const int i = i0 + lane;
        int v = 0, v2 = 0, pos = 0;
        if (i < total)
        {
            pos = i / T;
            const int t = i - pos * T;
            const int qx = mvx * 4 + a.pos[pos][0], qy = mvy * 4 + a.pos[pos][1];
            if (t < L)
            {
                const int ty = t / tw, tx = t - ty * tw;
                const int X = X0 + tx * 4, Y = Y0 + ty * 4;
                const int ph = (qy & 3) * 4 + (qx & 3);
                const uint8_t* src = ph ? a.phases[0] + (size_t)(ph - 1) * a.planeBytes : a.ref[0];
                const uint8_t* f = a.fenc[0] + (long)(a.marginY + Y) * a.strideB + (long)(a.marginX + X) * BPP;
                const uint8_t* r = src + (long)(a.marginY + Y + (qy >> 2)) * a.strideB + (long)(a.marginX + X + (qx >> 2)) * BPP;
                v = satd_tile<Px>(f, a.strideB, r, a.strideB);
                if (a.sadCosts) v2 = sad_tile<Px>(f, a.strideB, r, a.strideB);
            }
            else if (CHROMA)
            {
                const int tc = t - L, comp = tc >= C ? 2 : 1, u = tc - (comp - 1) * C;
                const int ty = u / cw, tx = u - ty * cw;
                const int X = (X0 >> 1) + tx * 4, Y = (Y0 >> 1) + ty * 4;
                const int ph = (qy & 7) * 8 + (qx & 7);                     // 4:2:0: the quarter-sample luma vector is an eighth-sample chroma vector (motion.cpp:1606-1607)
                const uint8_t* src = ph ? a.phases[comp] + (size_t)(ph - 1) * a.planeBytesC : a.ref[comp];
                const uint8_t* f = a.fenc[comp] + (long)(a.marginYC + Y) * a.strideCB + (long)(a.marginX + X) * BPP;
                const uint8_t* r = src + (long)(a.marginYC + Y + (qy >> 3)) * a.strideCB + (long)(a.marginX + X + (qx >> 3)) * BPP;
                v = satd_tile<Px>(f, a.strideCB, r, a.strideCB);
                v2 = v;                                                     // the chroma part of a SAD-typed comparison is SATD too (motion.cpp:1601-1661 adds chromaSatd whatever cmp is)
            }
        }
        if (whole)
        {
            v = group_sum<64>(v);
            if (a.sadCosts) v2 = group_sum<64>(v2);
            if (lane == 0) { acc[i0 / T] += (uint32_t)v; if (a.sadCosts) acc2[i0 / T] += (uint32_t)v2; }
        }
        else if (i < total)
        {
            atomicAdd(&acc[pos], (uint32_t)v);
            if (a.sadCosts) atomicAdd(&acc2[pos], (uint32_t)v2);
        }
    }
    __syncthreads();
    uint32_t lo = 0xffffffffu;
    for (int i = lane; i < a.npos; i += 64) lo = min(lo, acc[i]);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, m, 64));
    if (lane == 0)
    {
        reinterpret_cast<int16_t*>(rec)[0] = (int16_t)mvx; reinterpret_cast<int16_t*>(rec)[1] = (int16_t)mvy;
        reinterpret_cast<uint32_t*>(rec)[1] = lo;
    }
    uint16_t* delta = reinterpret_cast<uint16_t*>(rec + 8);
    for (int i = lane; i < a.npos; i += 64) delta[i] = (uint16_t)min(acc[i] - lo, 65535u);
    if (a.sadCosts)
    {
        uint32_t lo2 = 0xffffffffu;
        for (int i = lane; i < a.npos; i += 64) lo2 = min(lo2, acc2[i]);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) lo2 = min(lo2, (uint32_t)__shfl_xor((int)lo2, m, 64));
        if (lane == 0) *reinterpret_cast<uint32_t*>(rec + a.rec2Off) = lo2;
        uint16_t* delta2 = reinterpret_cast<uint16_t*>(rec + a.rec2Off + 4);
        for (int i = lane; i < a.npos; i += 64) delta2[i] = (uint16_t)min(acc2[i] - lo2, 65535u);
    }
    __syncthreads();                                         // acc is zeroed again for the next pair
    }
}

// The same records with the tile work SHARED between the PUs of a CTU (round 6, after the first encoder measurements: the service was late for a fifth of the
// comparisons at 4K preset slow - 12 - 21 ms of launches per pair).  Every listed PU is a union of 8x8 luma blocks, and in the content the service is built for most PUs of
// a CTU sit on the same one or two vectors: per DISTINCT candidate vector of the CTU the workgroup fills one map [64 blocks][positions] of block costs (the block's four
// luma tiles + its Cb and Cr tile, one lane per (tile, block, position) item, LDS atomics), then one wavefront per PU sums its blocks position by position - a tile is
// evaluated once per vector instead of once per PU that covers it (10 covering shapes with the rectangles).  CTUs with more distinct vectors than `maxDistinct` are
// flagged and left to cost_tables_kernel (a tile evaluated per PU is then the cheaper way).
enum { COST_MAX_DISTINCT = 40 };
template <typename Px, bool CHROMA>
__global__ void __launch_bounds__(256) cost_tables_shared_kernel(TableArgs a, uint8_t* __restrict__ ctuFlags, int maxDistinct)
{
    constexpr int BPP = sizeof(Px), TPB = CHROMA ? 6 : 4;
    __shared__ int16_t sCand[COST_MAX_PU * 2][2];
    __shared__ int16_t sVec[COST_MAX_DISTINCT + 1][2];
    __shared__ uint16_t sIdx[COST_MAX_PU * 2];
    __shared__ int sNd;
    extern __shared__ uint32_t sMap[];                       // [64 blocks][npos] SATD costs (+ a second map of the SAD-typed costs with sadCosts)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctuB = blockIdx.x, n = a.npu * a.K, npos = a.npos;
    const int ctuX = ctuB % a.ctusW, ctuY = a.ctuRow0 + ctuB / a.ctusW;
    for (int i = tid; i < n; i += 256)
    {
        sCand[i][0] = a.cand[((size_t)ctuB * n + i) * 2];
        sCand[i][1] = a.cand[((size_t)ctuB * n + i) * 2 + 1];
    }
    __syncthreads();
    if (tid == 0)
    {
        int nd = 0;
        for (int i = 0; i < n && nd <= maxDistinct; i++)
        {
            const int x = sCand[i][0], y = sCand[i][1];
            if (x == -32768) { sIdx[i] = 0xffff; continue; }
            int k = 0;
            while (k < nd && (sVec[k][0] != x || sVec[k][1] != y)) k++;
            if (k == nd) { sVec[nd][0] = (int16_t)x; sVec[nd][1] = (int16_t)y; nd++; }
            sIdx[i] = (uint16_t)k;
        }
        sNd = nd;
        ctuFlags[ctuB] = nd > maxDistinct ? 1 : 0;
    }
    __syncthreads();
    const int nd = sNd;
    if (nd > maxDistinct) return;
    for (int i = tid; i < n; i += 256)
        if (sIdx[i] == 0xffff)
        {
            uint32_t* rec = reinterpret_cast<uint32_t*>(a.tables + ((size_t)ctuB * n + i) * a.recBytes);
            for (int w = 0; w < a.recBytes / 4; w++) rec[w] = w ? 0u : 0x00008000u;
        }
    const int X0 = ctuX * 64, Y0 = ctuY * 64;
    for (int d = 0; d < nd; d++)
    {
        uint32_t* sMap2 = sMap + 64 * npos;
        for (int i = tid; i < (a.sadCosts ? 2 : 1) * 64 * npos; i += 256) sMap[i] = 0;
        __syncthreads();
        const int mvx = sVec[d][0], mvy = sVec[d][1];
        const int items = npos * 64 * TPB;
        for (int i = tid; i < items; i += 256)
        {
            const int rest = i / npos, pos = i - rest * npos, block = rest & 63, tile = rest >> 6;
            const int bx = block & 7, by = block >> 3;
            const int qx = mvx * 4 + a.pos[pos][0], qy = mvy * 4 + a.pos[pos][1];
            int v, v2;
            if (tile < 4)
            {
                const int X = X0 + bx * 8 + (tile & 1) * 4, Y = Y0 + by * 8 + (tile >> 1) * 4;
                const int ph = (qy & 3) * 4 + (qx & 3);
                const uint8_t* src = ph ? a.phases[0] + (size_t)(ph - 1) * a.planeBytes : a.ref[0];
                const uint8_t* f = a.fenc[0] + (long)(a.marginY + Y) * a.strideB + (long)(a.marginX + X) * BPP;
                const uint8_t* r = src + (long)(a.marginY + Y + (qy >> 2)) * a.strideB + (long)(a.marginX + X + (qx >> 2)) * BPP;
                v = satd_tile<Px>(f, a.strideB, r, a.strideB);
                v2 = a.sadCosts ? sad_tile<Px>(f, a.strideB, r, a.strideB) : 0;
            }
            else
            {
                const int comp = tile - 3;
                const int X = (X0 >> 1) + bx * 4, Y = (Y0 >> 1) + by * 4;
                const int ph = (qy & 7) * 8 + (qx & 7);
                const uint8_t* src = ph ? a.phases[comp] + (size_t)(ph - 1) * a.planeBytesC : a.ref[comp];
                const uint8_t* f = a.fenc[comp] + (long)(a.marginYC + Y) * a.strideCB + (long)(a.marginX + X) * BPP;
                const uint8_t* r = src + (long)(a.marginYC + Y + (qy >> 3)) * a.strideCB + (long)(a.marginX + X + (qx >> 3)) * BPP;
                v = satd_tile<Px>(f, a.strideCB, r, a.strideCB);
                v2 = v;
            }
            atomicAdd(&sMap[block * npos + pos], (uint32_t)v);
            if (a.sadCosts) atomicAdd(&sMap2[block * npos + pos], (uint32_t)v2);
        }
        __syncthreads();
        for (int pc = wave; pc < n; pc += 4)
        {
            if (sIdx[pc] != d) continue;
            const CostPu& P = kCostPu[pc / a.K];
            const int bx0 = P.x >> 3, by0 = P.y >> 3, bw = P.w >> 3, bh = P.h >> 3;
            uint8_t* rec = a.tables + ((size_t)ctuB * n + pc) * a.recBytes;
            for (int which = 0; which < (a.sadCosts ? 2 : 1); which++)
            {
                const uint32_t* map = which ? sMap2 : sMap;
                uint32_t c[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu };
#pragma unroll
                for (int j = 0; j < 3; j++)
                {
                    const int pos = j * 64 + lane;
                    if (pos >= npos) continue;
                    uint32_t sum = 0;
                    for (int y = 0; y < bh; y++)
                        for (int x = 0; x < bw; x++) sum += map[((by0 + y) * 8 + bx0 + x) * npos + pos];
                    c[j] = sum;
                }
                uint32_t lo = min(c[0], min(c[1], c[2]));
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, m, 64));
                uint8_t* part = which ? rec + a.rec2Off : rec + 4;
                if (lane == 0)
                {
                    if (!which) { reinterpret_cast<int16_t*>(rec)[0] = (int16_t)mvx; reinterpret_cast<int16_t*>(rec)[1] = (int16_t)mvy; }
                    *reinterpret_cast<uint32_t*>(part) = lo;
                }
                uint16_t* delta = reinterpret_cast<uint16_t*>(part + 4);
#pragma unroll
                for (int j = 0; j < 3; j++)
                {
                    const int pos = j * 64 + lane;
                    if (pos < npos) delta[pos] = (uint16_t)min(c[j] - lo, 65535u);
                }
            }
        }
        __syncthreads();
    }
}

namespace {

int upload_pu_list()
{
    static std::mutex mu;
    static uint64_t done = 0;                       // one bit per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 64 && (done >> dev) & 1) return 0;
    X265HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kCostPu), pu_list().pu, sizeof(CostPu) * COST_MAX_PU));
    if (dev < 64) done |= 1ull << dev;
    return 0;
}

} // namespace
} // namespace x265hip

using namespace x265hip;

extern "C" {

int x265hip_cost_pu_count(int shapes) { return (shapes < 0 || shapes > 2) ? 0 : pu_count(shapes); }

int x265hip_cost_pu_rect(int shapes, int pu, int rect[4])
{
    if (shapes < 0 || shapes > 2 || pu < 0 || pu >= pu_count(shapes) || !rect) { set_error("cost_pu_rect: PU %d of shape set %d", pu, shapes); return X265HIP_EINVAL; }
    const CostPu& p = pu_list().pu[pu];
    rect[0] = p.x; rect[1] = p.y; rect[2] = p.w; rect[3] = p.h;
    return 0;
}

int x265hip_cost_positions(int subme, int8_t* xy, int max_positions)
{
    PosSet P;
    if (!positions(subme, P)) { set_error("cost_positions: subme %d", subme); return X265HIP_EINVAL; }
    if (xy)
        for (int i = 0; i < P.n && i < max_positions; i++) { xy[2 * i] = P.xy[i][0]; xy[2 * i + 1] = P.xy[i][1]; }
    return P.n;
}

int x265hip_cost_record_bytes(int subme, int sad_costs)
{
    PosSet P;
    if (!positions(subme, P)) return 0;
    const int one = (8 + 2 * P.n + 3) & ~3;
    return sad_costs ? one + ((4 + 2 * P.n + 3) & ~3) : one;
}

size_t x265hip_cost_ctu_bytes(int subme, int shapes, int candidates, int sad_costs)
{
    if (shapes < 0 || shapes > 2 || candidates < 1 || candidates > 2) return 0;
    return (size_t)x265hip_cost_record_bytes(subme, sad_costs) * pu_count(shapes) * candidates;
}

int x265hip_cost_candidates(const x265hip_cost_candidates_params* p, void* stream)
{
    if (!p || !p->surf || !p->cand) { set_error("cost_candidates: NULL argument"); return X265HIP_EINVAL; }
    if (p->nctu < 1 || p->window < 0 || p->window > 64 || p->shapes < 0 || p->shapes > 2 || p->candidates < 1 || p->candidates > 2)
    { set_error("cost_candidates: %d CTUs, window %d, shape set %d, %d candidates", p->nctu, p->window, p->shapes, p->candidates); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = upload_pu_list())) return rc;
    CandArgs a = { p->surf, p->centres, p->cand, p->mv_cost, p->window, pu_count(p->shapes), p->candidates };
    hipLaunchKernelGGL(cost_cand_kernel, dim3(p->nctu), dim3(256), 0, (hipStream_t)stream, a);
    return check_hip(hipGetLastError(), "cost_candidates launch");
}

int x265hip_cost_tables(const x265hip_cost_tables_params* p, void* stream)
{
    if (!p || !p->fenc[0] || !p->ref[0] || !p->phases[0] || !p->cand || !p->tables) { set_error("cost_tables: NULL argument"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("cost_tables: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->chroma && (!p->fenc[1] || !p->fenc[2] || !p->ref[1] || !p->ref[2] || !p->phases[1] || !p->phases[2] || p->stride_c <= 0))
    { set_error("cost_tables: chroma costs need the chroma planes"); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    if (p->width < 64 || (p->width & 63) || p->ctu_rows < 1 || p->ctu_row0 < 0 || p->stride < p->width + 2 * p->margin_x || ((p->stride * bpp) & 3) || (p->margin_x & 3) ||
        p->shapes < 0 || p->shapes > 2 || p->candidates < 1 || p->candidates > 2)
    { set_error("cost_tables: geometry (width %d, pitch %ld, margin %d, band %d + %d, shape set %d, %d candidates)", p->width, (long)p->stride, p->margin_x, p->ctu_row0, p->ctu_rows,
                p->shapes, p->candidates); return X265HIP_EINVAL; }
    PosSet P;
    if (!positions(p->subme, P)) { set_error("cost_tables: subme %d", p->subme); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = upload_pu_list())) return rc;
    TableArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 3; i++) { a.fenc[i] = (const uint8_t*)p->fenc[i]; a.ref[i] = (const uint8_t*)p->ref[i]; a.phases[i] = (const uint8_t*)p->phases[i]; }
    a.planeBytes = p->plane_bytes; a.planeBytesC = p->plane_bytes_c;
    a.strideB = (long)p->stride * bpp; a.strideCB = (long)p->stride_c * bpp;
    a.marginX = p->margin_x; a.marginY = p->margin_y; a.marginYC = p->margin_y_c;
    a.ctusW = p->width / 64; a.ctuRow0 = p->ctu_row0; a.npu = pu_count(p->shapes); a.K = p->candidates; a.npos = P.n;
    a.sadCosts = p->sad_costs ? 1 : 0; a.rec2Off = (8 + 2 * P.n + 3) & ~3;
    a.recBytes = x265hip_cost_record_bytes(p->subme, a.sadCosts); a.maxVal = (1 << p->depth) - 1;
    a.cand = p->cand; a.tables = (uint8_t*)p->tables;
    memcpy(a.pos, P.xy, sizeof(a.pos));
    const size_t blocks = (size_t)p->ctu_rows * a.ctusW * COST_WAVES_PER_CTU;
    if (blocks > 0x7fffffffull) { set_error("cost_tables: band too large"); return X265HIP_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    // X265HIP_COST_SHARED=0: the per-PU kernel alone (A/B, tests of both routes); otherwise the shared-tile kernel first, then the per-PU kernel for the CTUs it flagged
    static const int shared = getenv("X265HIP_COST_SHARED") ? atoi(getenv("X265HIP_COST_SHARED")) : 1;
    const int nctuBand = p->ctu_rows * a.ctusW;
    uint8_t* flags = nullptr;
    std::unique_lock<std::mutex> seq;
    if (shared)
    {
        seq = stream_sequence_lock(s);                        // the flags live in the stream's scratch: written and read by this launch pair
        flags = (uint8_t*)stream_scratch(s, 5, (size_t)nctuBand);
        if (!flags) { set_error("cost_tables: no scratch for %d CTU flags", nctuBand); return X265HIP_ENODEV; }
        const int maxDistinct = shared > 1 ? shared : (a.K == 1 ? 10 : 20);       // break-even: a tile evaluated once per vector against once per covering PU
        const size_t lds = (size_t)(a.sadCosts ? 2 : 1) * 64 * P.n * sizeof(uint32_t);
        const dim3 g2((unsigned)nctuBand), b2(256);
        const int md = maxDistinct > COST_MAX_DISTINCT ? COST_MAX_DISTINCT : maxDistinct;
        if (lds > 40 * 1024)
        {
            // beyond the default dynamic LDS limit (the statics take 4 KB): gfx950 has 160 KB per CU
            if (bpp == 1) { if (p->chroma) X265HIP_TRY(hipFuncSetAttribute((const void*)cost_tables_shared_kernel<uint8_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); else X265HIP_TRY(hipFuncSetAttribute((const void*)cost_tables_shared_kernel<uint8_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
            else          { if (p->chroma) X265HIP_TRY(hipFuncSetAttribute((const void*)cost_tables_shared_kernel<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); else X265HIP_TRY(hipFuncSetAttribute((const void*)cost_tables_shared_kernel<uint16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
        }
        if (bpp == 1) { if (p->chroma) hipLaunchKernelGGL((cost_tables_shared_kernel<uint8_t, true>), g2, b2, lds, s, a, flags, md); else hipLaunchKernelGGL((cost_tables_shared_kernel<uint8_t, false>), g2, b2, lds, s, a, flags, md); }
        else          { if (p->chroma) hipLaunchKernelGGL((cost_tables_shared_kernel<uint16_t, true>), g2, b2, lds, s, a, flags, md); else hipLaunchKernelGGL((cost_tables_shared_kernel<uint16_t, false>), g2, b2, lds, s, a, flags, md); }
        X265HIP_TRY(hipGetLastError());
    }
    const dim3 grid((unsigned)blocks), block(64);
    if (bpp == 1) { if (p->chroma) hipLaunchKernelGGL((cost_tables_kernel<uint8_t, true>), grid, block, 0, s, a, (const uint8_t*)flags); else hipLaunchKernelGGL((cost_tables_kernel<uint8_t, false>), grid, block, 0, s, a, (const uint8_t*)flags); }
    else          { if (p->chroma) hipLaunchKernelGGL((cost_tables_kernel<uint16_t, true>), grid, block, 0, s, a, (const uint8_t*)flags); else hipLaunchKernelGGL((cost_tables_kernel<uint16_t, false>), grid, block, 0, s, a, (const uint8_t*)flags); }
    return check_hip(hipGetLastError(), "cost_tables launch");
}

} // extern "C"
