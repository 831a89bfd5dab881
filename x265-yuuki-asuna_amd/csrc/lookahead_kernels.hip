// lookahead_kernels.hip - lookahead picture preparation and intra cost estimate on gfx950
// (the intra half).
//
// Reference semantics:
//   Lowres::init (source/common/lowres.cpp:294-306): primitives.frameInitLowres (frame_init_lowres_core,
//     source/common/pixel.cpp:604-629) fills the four half-resolution planes - full-pel, H, V and HV phase, every
//     sample the rounded average of two rounded vertical averages - and extendPicBorder pads each of them;
//   LookaheadTLD::lowresIntraEstimate (source/encoder/slicetype.cpp:696-772): per 8x8 block of lowres plane 0 -
//     neighbours read from the padded plane, intra_filter (intrapred.cpp:31-51), DC / planar / angular predictions
//     (intrapred.cpp:53-204) scored with satd 8x8 (pixel.cpp:239-297), modes scanned 5,10..30 then +-2, +-1 around
//     the best angle, COPY2_IF_LT (strict <) order, + intraPenalty + lowresPenalty.
//
// Mapping: lowres_init - a thread produces 4 horizontally adjacent samples of all four planes from a 3 x 12 source
// patch (dword loads).  lowres_intra - one thread per 4x4 tile, four threads (one DPP quad) per 8x8 block, 64 blocks per
// workgroup; the block's 33 + 33 neighbour samples live in LDS, every candidate mode is predicted sample by sample in
// registers (closed form of the reference's projected reference line), Hadamard-transformed per tile and summed over
// the quad; the data-dependent mode scan runs per quad without divergence because the mode only enters as data.
#include "common.h"
#include "tile_interp.h"

namespace x265hip {

__constant__ int8_t kLaAngle[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int16_t kLaInvAngle[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
__constant__ uint8_t kLaFilterFlags[35] = {
    0x38, 0x00,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38 };

struct LowresInitArgs
{
    const uint8_t* src; long srcStrideB;
    uint8_t* dst[4]; long dstStrideB;
    int width, lines;                 // lowres size
};

__device__ __forceinline__ int la_avg(int a, int b) { return (a + b + 1) >> 1; }

template <typename Px>
__global__ void __launch_bounds__(256) lowres_init_kernel(LowresInitArgs a)
{
    constexpr int BPP = sizeof(Px);
    const int qpr = a.width >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qpr * a.lines) return;
    const int y = q / qpr, x0 = (q - y * qpr) * 4;
    // rows 2y, 2y+1, 2y+2 of the source, samples 2*x0 .. 2*x0 + 8
    int s[3][9];
#pragma unroll
    for (int r = 0; r < 3; r++)
    {
        const uint8_t* rp = a.src + (long)(2 * y + r) * a.srcStrideB + (long)(2 * x0) * BPP;
        if (BPP == 1)
        {
            const uint32_t w0 = ld_u32(rp), w1 = ld_u32(rp + 4), w2 = ld_u32(rp + 8);
#pragma unroll
            for (int k = 0; k < 4; k++) { s[r][k] = (w0 >> (8 * k)) & 0xff; s[r][4 + k] = (w1 >> (8 * k)) & 0xff; }
            s[r][8] = w2 & 0xff;
        }
        else
        {
#pragma unroll
            for (int k = 0; k < 5; k++)
            {
                const uint32_t w = ld_u32(rp + 4 * k);
                s[r][2 * k] = w & 0xffff;
                if (2 * k + 1 < 9) s[r][2 * k + 1] = w >> 16;
            }
        }
    }
    int v01[9], v12[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { v01[k] = la_avg(s[0][k], s[1][k]); v12[k] = la_avg(s[1][k], s[2][k]); }
    int o[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        o[0][k] = la_avg(v01[2 * k], v01[2 * k + 1]);
        o[1][k] = la_avg(v01[2 * k + 1], v01[2 * k + 2]);
        o[2][k] = la_avg(v12[2 * k], v12[2 * k + 1]);
        o[3][k] = la_avg(v12[2 * k + 1], v12[2 * k + 2]);
    }
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
    {
        uint8_t* dp = a.dst[pl] + (long)y * a.dstStrideB + (long)x0 * BPP;
        if (BPP == 1)
            *reinterpret_cast<u32_unaligned*>(dp) = (uint32_t)o[pl][0] | ((uint32_t)o[pl][1] << 8) | ((uint32_t)o[pl][2] << 16) | ((uint32_t)o[pl][3] << 24);
        else
        {
            reinterpret_cast<u32_unaligned*>(dp)[0] = (uint32_t)o[pl][0] | ((uint32_t)o[pl][1] << 16);
            reinterpret_cast<u32_unaligned*>(dp)[1] = (uint32_t)o[pl][2] | ((uint32_t)o[pl][3] << 16);
        }
    }
}

struct LowresIntraArgs
{
    const uint8_t* plane; long strideB;
    int widthInCU, heightInCU, depth, intraPenalty;
    int* intraCost; uint8_t* intraMode; uint16_t* lowresCosts;
};

// One predicted sample of an 8x8 block (n = 8): the reference's per-mode code in closed form.
template <typename Px>
__device__ __forceinline__ int la_sample(const Px* nb, int mode, int dc, int maxVal, int x, int y)
{
    constexpr int n = 8, n2 = 16;
    if (mode == 0)
        return ((n - 1 - x) * nb[n2 + 1 + y] + (n - 1 - y) * nb[1 + x] + (x + 1) * nb[1 + n] + (y + 1) * nb[n2 + 1 + n] + n) >> 4;
    if (mode == 1)       // DC with edge filter (bFilter = cuSize <= 16)
    {
        if (x == 0 && y == 0) return (nb[1] + nb[n2 + 1] + 2 * dc + 2) >> 2;
        if (y == 0) return (nb[1 + x] + 3 * dc + 2) >> 2;
        if (x == 0) return (nb[n2 + 1 + y] + 3 * dc + 2) >> 2;
        return dc;
    }
    const bool hor = mode < 18;
    const int r = hor ? x : y, c = hor ? y : x;                      // horizontal modes predict the transpose
    const int mainBase = hor ? n2 : 0, sideBase = hor ? 0 : n2;
    const int aoff = hor ? 10 - mode : mode - 26;
    const int angle = kLaAngle[8 + aoff];
    if (angle == 0)
    {
        int v = nb[mainBase + 1 + c];
        if (c == 0)                                                  // bFilter
        {
            const int16_t t = (int16_t)(nb[mainBase + 1] + (((int)nb[sideBase + 1 + r] - (int)nb[0]) >> 1));
            v = t < 0 ? 0 : (t > maxVal ? maxVal : t);
        }
        return v;
    }
    const int inv = angle < 0 ? kLaInvAngle[-aoff - 1] : 0;
    auto ref = [&](const int k) -> int
    {
        if (k >= 0) return nb[mainBase + 1 + k];
        if (k == -1) return nb[0];
        return nb[sideBase + ((128 + (-1 - k) * inv) >> 8)];
    };
    const int pos = (r + 1) * angle, off = pos >> 5, frac = pos & 31;
    const int p0 = ref(off + c);
    if (!frac) return p0;
    return ((32 - frac) * p0 + frac * ref(off + c + 1) + 16) >> 5;
}

// LookaheadTLD::lowresIntraEstimate (slicetype.cpp:718-805) for the 8x8 blocks of the half-resolution picture.  The reference walks
// twelve modes one after the other (DC, planar, angular 5 .. 30 in steps of 5, then +-2 and +-1 around the best angular mode); here the
// modes of a block are evaluated SIDE BY SIDE: a block has 32 lanes = 8 mode slots x the 4 tiles (4x4) of the block, and two rounds -
//   round 1: DC, planar and the six coarse angular modes, one per slot;
//   round 2: the six modes a - 3 .. a + 3 (a = the coarse winner) the two refinement steps can ever look at (the +-1 step starts from
//            a - 2, a or a + 2), one per slot, two slots idle;
// the winners are then picked in the reference's order with its strict-less comparisons, so ties fall the same way.  Fourteen mode costs
// instead of twelve, but a dependent chain of two instead of twelve and eight times the lanes: the sequential organisation (one lane
// per tile, 507 workgroups at 4K) ran 110 us with two wavefronts per SIMD waiting on each other's LDS reads
// (profiles/r02_bench_final_pmc.txt), this one is in profiles/r03_*.
template <typename Px>
__global__ void __launch_bounds__(256) lowres_intra_kernel(LowresIntraArgs a)
{
    constexpr int BPP = sizeof(Px);
    __shared__ Px nbS[8][36], nbF[8][36];
    __shared__ int16_t extS[8][8][28];
    const int tid = threadIdx.x;
    const int tile = tid & 3, slot = (tid >> 2) & 7, bw = tid >> 5;      // 4x4 tile within the block, mode slot, block within the workgroup
    const int l32 = tid & 31;
    const int ncu = a.widthInCU * a.heightInCU;
    const int cuXY = blockIdx.x * 8 + bw;
    const bool live = cuXY < ncu;
    const int cuX = live ? cuXY % a.widthInCU : 0, cuY = live ? cuXY / a.widthInCU : 0;
    const Px* pix = reinterpret_cast<const Px*>(a.plane + (long)(8 * cuY) * a.strideB) + 8 * cuX;
    const long st = a.strideB / BPP;
    const int maxVal = (1 << a.depth) - 1;

    // neighbours: [0] corner, [1..16] above + above-right, [17..32] left + below-left (slicetype.cpp:725-729)
    for (int k = l32; k < 33; k += 32)
        nbS[bw][k] = k <= 16 ? pix[-st - 1 + k] : pix[-1 + (long)(k - 17) * st];
    __syncthreads();
    for (int i = l32; i < 33; i += 32)                              // intra_filter, intrapred.cpp:31-51
    {
        const Px* s = nbS[bw];
        int v;
        if (i == 0) v = (2 * s[0] + s[1] + s[17] + 2) >> 2;
        else if (i == 16 || i == 32) v = s[i];
        else if (i == 17) v = (2 * s[17] + s[0] + s[18] + 2) >> 2;
        else if (i == 1) v = (2 * s[1] + s[0] + s[2] + 2) >> 2;
        else v = (2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2;
        nbF[bw][i] = (Px)v;
    }
    __syncthreads();

    // this thread's 4x4 source tile
    const int tx = (tile & 1) * 4, ty = (tile >> 1) * 4;
    int src[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        if (BPP == 1)
        {
            const uint32_t w = *reinterpret_cast<const u32_unaligned*>(pix + (long)(ty + y) * st + tx);
            src[y][0] = w & 255; src[y][1] = (w >> 8) & 255; src[y][2] = (w >> 16) & 255; src[y][3] = w >> 24;
        }
        else
        {
            const uint32_t w0 = reinterpret_cast<const u32_unaligned*>(pix + (long)(ty + y) * st + tx)[0], w1 = reinterpret_cast<const u32_unaligned*>(pix + (long)(ty + y) * st + tx)[1];
            src[y][0] = w0 & 0xffff; src[y][1] = w0 >> 16; src[y][2] = w1 & 0xffff; src[y][3] = w1 >> 16;
        }
    }
    int dcSum = 8;
    for (int i = 0; i < 8; i++) dcSum += nbS[bw][1 + i] + nbS[bw][17 + i];
    const int dc = dcSum / 16;

    // One mode's SATD of this lane's 4x4 tile.  The angular modes (intrapred.cpp:104-214 in closed form, as la_sample states it per
    // sample) are evaluated the way the reference builds them: the 25 reference samples ref(-8) .. ref(16) of the mode - the main side, the
    // corner, and for negative angles the side samples projected through the inverse angle - are laid out once per (block, mode slot) in
    // LDS by the slot's four lanes; a line of the tile (a row of a vertical mode, a column of a horizontal one - the transposed tile has
    // the same SATD) then has ONE position / fraction and reads five neighbouring samples for its four predictions.  ~190 instructions
    // per tile and mode instead of ~130 per SAMPLE of the generic per-sample function (what the first mode-parallel version still did:
    // 103 us at 4K, issue-bound, profiles/r03_tail_kernels.txt).
    int16_t* ext = extS[bw][slot];
    auto satd4 = [&](const int (&d)[4][4]) -> int
    {
        int t4[4][4], acc = 0;
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            const int s0 = d[y][0] + d[y][1], s1 = d[y][0] - d[y][1], s2 = d[y][2] + d[y][3], s3 = d[y][2] - d[y][3];
            t4[y][0] = s0 + s2; t4[y][1] = s1 + s3; t4[y][2] = s0 - s2; t4[y][3] = s1 - s3;
        }
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            const int s0 = t4[0][x] + t4[1][x], s1 = t4[0][x] - t4[1][x], s2 = t4[2][x] + t4[3][x], s3 = t4[2][x] - t4[3][x];
            acc += abs(s0 + s2) + abs(s1 + s3) + abs(s0 - s2) + abs(s1 - s3);
        }
        return quad_sum(acc >> 1);                                   // every 4x4 abs-sum is even; 8x8 satd = sum of its four tiles
    };
    auto mode_cost = [&](const int mode) -> int
    {
        int d[4][4];
        if (mode < 2)
        {   // DC (unfiltered neighbours, edge filter: bFilter = cuSize <= 16) and planar (filtered neighbours)
            const Px* nb = mode ? nbS[bw] : nbF[bw];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = src[y][x] - la_sample<Px>(nb, mode, dc, maxVal, tx + x, ty + y);
            return satd4(d);
        }
        const Px* nb = (kLaFilterFlags[mode] & 8) ? nbF[bw] : nbS[bw];          // modes 2, 18, 34 of an 8x8 block read the filtered set
        const bool hor = mode < 18;
        const int mainBase = hor ? 16 : 0, sideBase = hor ? 0 : 16;
        const int aoff = hor ? 10 - mode : mode - 26;
        const int angle = kLaAngle[8 + aoff];
        const int inv = angle < 0 ? kLaInvAngle[-aoff - 1] : 0;
        for (int e = tile; e < 25; e += 4)
        {
            const int k = e - 8;
            int idx = (128 + (-1 - k) * inv) >> 8;                   // (entries no line of this mode reaches hold a clamped, unused sample)
            idx = idx > 16 ? 16 : idx;
            ext[e] = (int16_t)(k >= 0 ? nb[mainBase + 1 + k] : (k == -1 ? nb[0] : nb[sideBase + idx]));
        }
        __builtin_amdgcn_wave_barrier();                              // the slot's four lanes sit in one wavefront: LDS operations are in order
        const int r0 = hor ? tx : ty, c0 = hor ? ty : tx;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int pos = (r0 + i + 1) * angle, off = pos >> 5, frac = pos & 31;
            const int16_t* q = ext + 8 + off + c0;
            int e5[5];
#pragma unroll
            for (int j = 0; j < 5; j++) e5[j] = q[j];
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int pr = ((32 - frac) * e5[j] + frac * e5[j + 1] + 16) >> 5;       // frac == 0: exactly e5[j]
                d[i][j] = (hor ? src[j][i] : src[i][j]) - pr;
            }
        }
        if (angle == 0 && c0 == 0)
        {   // the pure horizontal / vertical mode's edge filter on the first line across the prediction direction (intrapred.cpp:190-203)
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int16_t t = (int16_t)((int)nb[mainBase + 1] + (((int)nb[sideBase + 1 + r0 + i] - (int)nb[0]) >> 1));
                const int v = t < 0 ? 0 : (t > maxVal ? maxVal : t);
                d[i][0] = (hor ? src[0][i] : src[i][0]) - v;
            }
        }
        __builtin_amdgcn_wave_barrier();                              // round 2 rewrites the slot's line after every lane has read it
        return satd4(d);
    };
    // the cost slot `sl` of this lane's block holds (the block's 32 lanes sit in one half of the wavefront)
    const int lane = tid & 63, half = lane & 32;
    auto from_slot = [&](const int v, const int sl) -> int { return __shfl(v, half | (sl << 2) | tile, 64); };

    // round 1: slot 0 DC, slot 1 planar (the filtered neighbours: kLaFilterFlags[0] & 8), slots 2 .. 7 angular 5 .. 30
    const int c1 = mode_cost(slot == 0 ? 1 : (slot == 1 ? 0 : 5 * (slot - 1)));
    int icost = 1 << 28, ilow = 0;
    int cost = from_slot(c1, 0);
    if (cost < icost) { icost = cost; ilow = 1; }
    cost = from_slot(c1, 1);
    if (cost < icost) { icost = cost; ilow = 0; }
    int acost = 1 << 28, alow = 4;
#pragma unroll
    for (int k = 2; k < 8; k++)
    {
        cost = from_slot(c1, k);
        if (cost < acost) { acost = cost; alow = 5 * (k - 1); }
    }
    // round 2: slots 0 .. 5 = a - 3, a - 2, a - 1, a + 1, a + 2, a + 3 (slots 6, 7 repeat a + 3)
    const int a0 = alow;
    const int o2 = slot < 3 ? slot - 3 : (slot < 6 ? slot - 2 : 3);
    const int c2 = mode_cost(a0 + o2);
    auto at = [&](const int off) -> int { return from_slot(c2, off < 0 ? off + 3 : off + 2); };
    const int cm3 = at(-3), cm2 = at(-2), cm1 = at(-1), cp1 = at(1), cp2 = at(2), cp3 = at(3);
    if (cm2 < acost) { acost = cm2; alow = a0 - 2; }
    if (cp2 < acost) { acost = cp2; alow = a0 + 2; }
    {
        const int b = alow - a0;                                     // -2, 0, 2
        const int minus = b < 0 ? cm3 : (b == 0 ? cm1 : cp1), plus = b < 0 ? cm1 : (b == 0 ? cp1 : cp3);
        const int base = alow;
        if (minus < acost) { acost = minus; alow = base - 1; }
        if (plus < acost) { acost = plus; alow = base + 1; }
    }
    if (acost < icost) { icost = acost; ilow = alow; }
    icost += a.intraPenalty + 4;
    if (live && tile == 0 && slot == 0)
    {
        a.intraCost[cuXY] = icost;
        a.intraMode[cuXY] = (uint8_t)ilow;
        a.lowresCosts[cuXY] = (uint16_t)(icost < 16383 ? icost : 16383);
    }
}

// ---- weighted-reference analysis (LookaheadTLD::weightCostLuma / weightsAnalyse, slicetype.cpp:807-957)
struct WeightCostArgs
{
    const uint8_t* fenc; const uint8_t* ref; long strideB;
    int width, lines, depth;
    const int32_t* intraCost;
    int cand[4][4];
    uint32_t* cost;
};

// primitives.weight_pp's arithmetic for one sample (pixel.cpp:535-536)
__device__ __forceinline__ int weight_sample(int v, int scale, int round, int shift, int offset, int correction, int maxVal)
{
    const int val = (int)(int16_t)(v << correction);
    return clip3(0, maxVal, ((scale * val + round) >> shift) + offset);
}

// one thread per 8x8 block, blockIdx.y = candidate weight: weighted reference vs source through four 4x4 Hadamards (satd8 = two
// satd_8x4, each the two half sums of a 4x4 pair, pixel.cpp:239-297), capped by the block's intra cost, summed per candidate
template <typename Px>
__global__ void __launch_bounds__(256) lowres_weight_cost_kernel(WeightCostArgs a)
{
    const int bw = (a.width + 7) >> 3, nblk = bw * ((a.lines + 7) >> 3);
    const int mb = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    int v = 0;
    if (mb < nblk)
    {
        const int by = mb / bw, bx = mb - by * bw;
        const int present = a.cand[c][0], scale = a.cand[c][1], denom = a.cand[c][2];
        const int correction = 14 - a.depth, maxVal = (1 << a.depth) - 1;
        const int offset = a.cand[c][3] << (a.depth - 8), round = (denom ? 1 << (denom - 1) : 0) << correction, shift = denom + correction;
        const Px* f = reinterpret_cast<const Px*>(a.fenc + (long)(by * 8) * a.strideB) + bx * 8;
        const Px* r = reinterpret_cast<const Px*>(a.ref + (long)(by * 8) * a.strideB) + bx * 8;
        const long st = a.strideB / (long)sizeof(Px);
        int satd = 0;
#pragma unroll
        for (int t = 0; t < 4; t++)
        {
            int d[4][4];
            const int ox = (t & 1) * 4, oy = (t >> 1) * 4;
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int rv = (int)r[(oy + y) * st + ox + x];
                    if (present) rv = weight_sample(rv, scale, round, shift, offset, correction, maxVal);
                    d[y][x] = rv - (int)f[(oy + y) * st + ox + x];
                }
            satd += tile_satd4(d);
        }
        v = min(satd, a.intraCost[mb]);
    }
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&a.cost[c], (uint32_t)v);
}

// ---- adaptive-quantisation pass: block energies + picture statistics (acEnergyCu, slicetype.cpp:48-82,256-275)
struct AqArgs
{
    const uint8_t* y; const uint8_t* cb; const uint8_t* cr;
    long strideB, strideCB;
    int width, height, inc;
    uint32_t* energy; unsigned long long* slots;       // [64][6] partial picture totals
};

// 16 lanes per block, one luma row each (the first inc / 2 lanes also take a row of each chroma block); DPP row sums give the
// block's sum / sum of squares per plane, a wavefront sum and six atomics per wavefront the picture totals
template <typename Px>
__global__ void __launch_bounds__(256) aq_energy_kernel(AqArgs a)
{
    const int inc = a.inc, bw = (a.width + inc - 1) / inc, nblk = bw * ((a.height + inc - 1) / inc);
    const int blk = (blockIdx.x * 256 + threadIdx.x) >> 4, row = threadIdx.x & 15;
    uint32_t sum[3] = { 0, 0, 0 }, sqr[3] = { 0, 0, 0 };
    if (blk < nblk)
    {
        const int by = blk / bw, bx = blk - by * bw;
        if (row < inc)
        {
            const Px* p = reinterpret_cast<const Px*>(a.y + (long)(by * inc + row) * a.strideB) + bx * inc;
            for (int x = 0; x < inc; x++) { const uint32_t v = p[x]; sum[0] += v; sqr[0] += v * v; }
        }
        if (a.cb && row < (inc >> 1))
        {
            const int ci = inc >> 1;
            const Px* pb = reinterpret_cast<const Px*>(a.cb + (long)(by * ci + row) * a.strideCB) + bx * ci;
            const Px* pr = reinterpret_cast<const Px*>(a.cr + (long)(by * ci + row) * a.strideCB) + bx * ci;
            for (int x = 0; x < ci; x++)
            {
                const uint32_t u = pb[x], v = pr[x];
                sum[1] += u; sqr[1] += u * u; sum[2] += v; sqr[2] += v * v;
            }
        }
    }
    const int lshift = inc == 8 ? 6 : 8, cshift = inc == 8 ? 4 : 6;
    uint32_t energy = 0;
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
        sum[c] = (uint32_t)group_sum<16>((int)sum[c]);
        sqr[c] = (uint32_t)group_sum<16>((int)sqr[c]);
        if (c == 0 || a.cb) energy += sqr[c] - (uint32_t)(((unsigned long long)sum[c] * sum[c]) >> (c ? cshift : lshift));
    }
    if (blk < nblk && row == 0) a.energy[blk] = energy;
    // picture totals: one lane per block contributes; wavefront sums meet in LDS, the workgroup adds its six totals to one of 64
    // slots (a single set of counters serialises ~50 000 atomics on six addresses: 0.6 ms per 4K picture)
    __shared__ unsigned long long part[4][6];
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
        const long long s = group_sum<64>((long long)(row == 0 ? sum[c] : 0u)), q = group_sum<64>((long long)(row == 0 ? sqr[c] : 0u));
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][c] = (unsigned long long)s; part[threadIdx.x >> 6][3 + c] = (unsigned long long)q; }
    }
    __syncthreads();
    if (threadIdx.x < 6 && (a.cb || threadIdx.x % 3 == 0))
        atomicAdd(&a.slots[(blockIdx.x & 63) * 6 + threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ void aq_totals_kernel(const unsigned long long* slots, unsigned long long* wp)
{
    if (threadIdx.x >= 6) return;
    unsigned long long t = 0;
    for (int i = 0; i < 64; i++) t += slots[i * 6 + threadIdx.x];
    wp[threadIdx.x] = t;
}

// ---- cuTree: one propagation step (Lookahead::estimateCUPropagate, slicetype.cpp:2641-2753; propagateCost, pixel.cpp:914-940)
struct CuTreeArgs
{
    int w, h;
    const uint16_t* propagateIn; const int32_t* intraCost; const uint16_t* lowresCosts; const int32_t* invQscale;
    const int32_t* mvs[2];
    double fps;                                     // fpsFactor / 256
    int bipred[2];
    unsigned long long* acc[2];                     // 64-bit sums per target block and list
    uint16_t* refCost[2];
};

// a thread per source block: the amount in double precision with separate multiply / add / divide roundings (__dmul_rn & co. are
// never contracted into FMAs: the reference's C is not), then up to four atomic adds per list
__global__ void __launch_bounds__(256) cutree_scatter_kernel(CuTreeArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.w * a.h) return;
    const int blocky = i / a.w, blockx = i - blocky * a.w;
    const int intra = a.intraCost[i];
    const int lc = a.lowresCosts[i];
    const int inter = min(intra, lc & 0x3fff);
    const double propagateIntra = (double)(intra * a.invQscale[i]);
    const double amount = __dadd_rn((double)(a.propagateIn ? (int)a.propagateIn[i] : 0), __dmul_rn(propagateIntra, a.fps));
    const double r = __dadd_rn(__ddiv_rn(__dmul_rn(amount, (double)(intra - inter)), (double)intra), 0.5);
    if (!(r >= 1.0) || r >= 2147483648.0) return;   // (int)r <= 0, NaN (intra cost 0) or out of int range: nothing is passed on
    const int amountI = (int)r;
    const int lists = lc >> 14;
#pragma unroll
    for (int list = 0; list < 2; list++)
    {
        if (!((lists >> list) & 1)) continue;
        int listamount = amountI;
        if (lists == 3) listamount = (listamount * a.bipred[list] + 32) >> 6;
        int x = a.mvs[list][2 * i], y = a.mvs[list][2 * i + 1];
        unsigned long long* acc = a.acc[list];
        if (!x && !y) { atomicAdd(&acc[i], (unsigned long long)listamount); continue; }
        const int cux = (x >> 5) + blockx, cuy = (y >> 5) + blocky;
        x &= 31; y &= 31;
        const int wgt[4] = { (32 - y) * (32 - x), (32 - y) * x, y * (32 - x), y * x };
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int tx = cux + (k & 1), ty = cuy + (k >> 1);
            if (tx < 0 || ty < 0 || tx >= a.w || ty >= a.h) continue;
            const int v = (listamount * wgt[k] + 512) >> 10;
            if (v > 0) atomicAdd(&acc[ty * a.w + tx], (unsigned long long)v);
        }
    }
}

__global__ void __launch_bounds__(256) cutree_finish_kernel(CuTreeArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.w * a.h) return;
    uint16_t* rc = a.refCost[blockIdx.y];
    if (!rc) return;
    const unsigned long long t = (unsigned long long)rc[i] + a.acc[blockIdx.y][i];
    rc[i] = (uint16_t)(t < 65535ull ? t : 65535ull);
}

struct WeightApplyArgs
{
    const uint8_t* src[4]; uint8_t* dst[4];
    long n; int depth, scale, denom, offset;
};

template <typename Px>
__global__ void __launch_bounds__(256) lowres_weight_apply_kernel(WeightApplyArgs a)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= a.n) return;
    const int correction = 14 - a.depth, maxVal = (1 << a.depth) - 1;
    const int offset = a.offset << (a.depth - 8), round = (a.denom ? 1 << (a.denom - 1) : 0) << correction, shift = a.denom + correction;
    const Px* s = reinterpret_cast<const Px*>(a.src[blockIdx.y]);
    Px* d = reinterpret_cast<Px*>(a.dst[blockIdx.y]);
    d[i] = (Px)weight_sample((int)s[i], a.scale, round, shift, offset, correction, maxVal);
}

} // namespace x265hip

namespace x265hip { int extend_borders(void* const* pics, int nplanes, intptr_t stride, int width, int height, int margin_x, int margin_y, int depth, hipStream_t s); }

using namespace x265hip;

extern "C" int x265hip_lowres_init(const x265hip_lowres_init_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->src || !p->plane[0] || !p->plane[1] || !p->plane[2] || !p->plane[3]) { set_error("lowres_init: NULL plane"); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->lines <= 0 || (p->width & 7) || (p->lines & 7)) { set_error("lowres_init: lowres size %dx%d must be positive multiples of 8", p->width, p->lines); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("lowres_init: depth %d", p->depth); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    LowresInitArgs a;
    a.src = (const uint8_t*)p->src; a.srcStrideB = (long)p->src_stride * bpp;
    for (int i = 0; i < 4; i++) a.dst[i] = (uint8_t*)p->plane[i];
    a.dstStrideB = (long)p->stride * bpp; a.width = p->width; a.lines = p->lines;
    const int quads = (p->width >> 2) * p->lines;
    hipStream_t s = (hipStream_t)stream;
    if (bpp == 1) hipLaunchKernelGGL(lowres_init_kernel<uint8_t>, dim3((quads + 255) / 256), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lowres_init_kernel<uint16_t>, dim3((quads + 255) / 256), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    return extend_borders(p->plane, 4, p->stride, p->width, p->lines, p->margin_x, p->margin_y, p->depth, s);      // the four planes in one launch
}

extern "C" int x265hip_lowres_intra(const x265hip_lowres_intra_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->plane || !p->intra_cost || !p->intra_mode || !p->lowres_costs) { set_error("lowres_intra: NULL operand"); return X265HIP_EINVAL; }
    if (p->width_in_cu <= 0 || p->height_in_cu <= 0) { set_error("lowres_intra: empty picture"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("lowres_intra: depth %d", p->depth); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    LowresIntraArgs a;
    a.plane = (const uint8_t*)p->plane; a.strideB = (long)p->stride * bpp;
    a.widthInCU = p->width_in_cu; a.heightInCU = p->height_in_cu; a.depth = p->depth; a.intraPenalty = p->intra_penalty;
    a.intraCost = p->intra_cost; a.intraMode = p->intra_mode; a.lowresCosts = p->lowres_costs;
    const int ncu = p->width_in_cu * p->height_in_cu;
    hipStream_t s = (hipStream_t)stream;
    if (bpp == 1) hipLaunchKernelGGL(lowres_intra_kernel<uint8_t>, dim3((ncu + 7) / 8), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lowres_intra_kernel<uint16_t>, dim3((ncu + 7) / 8), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_lowres_weight_cost(const x265hip_lowres_weight_cost_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->ref || !p->intra_cost || !p->cost) { set_error("lowres_weight_cost: NULL operand"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("lowres_weight_cost: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->lines <= 0) { set_error("lowres_weight_cost: empty picture"); return X265HIP_EINVAL; }
    if (p->ncand < 1 || p->ncand > 4) { set_error("lowres_weight_cost: ncand %d out of [1,4]", p->ncand); return X265HIP_EINVAL; }
    for (int i = 0; i < p->ncand; i++)
        if (p->cand[i][0] && (p->cand[i][1] < 0 || p->cand[i][1] > 127 || p->cand[i][2] < 0 || p->cand[i][2] > 7 || p->cand[i][3] < -128 || p->cand[i][3] > 127))
        { set_error("lowres_weight_cost: candidate %d {scale %d, denom %d, offset %d} out of range", i, p->cand[i][1], p->cand[i][2], p->cand[i][3]); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    WeightCostArgs a;
    a.fenc = (const uint8_t*)p->fenc; a.ref = (const uint8_t*)p->ref; a.strideB = (long)p->stride * bpp;
    a.width = p->width; a.lines = p->lines; a.depth = p->depth; a.intraCost = p->intra_cost; a.cost = p->cost;
    for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) a.cand[i][k] = p->cand[i][k];
    hipStream_t s = (hipStream_t)stream;
    X265HIP_TRY(hipMemsetAsync(p->cost, 0, sizeof(uint32_t) * p->ncand, s));
    const int nblk = ((p->width + 7) >> 3) * ((p->lines + 7) >> 3);
    if (bpp == 1) hipLaunchKernelGGL(lowres_weight_cost_kernel<uint8_t>, dim3((nblk + 255) / 256, p->ncand), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lowres_weight_cost_kernel<uint16_t>, dim3((nblk + 255) / 256, p->ncand), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_lowres_weight_apply(const x265hip_lowres_weight_apply_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p) { set_error("lowres_weight_apply: NULL operand"); return X265HIP_EINVAL; }
    for (int i = 0; i < 4; i++) if (!p->src[i] || !p->dst[i]) { set_error("lowres_weight_apply: NULL plane %d", i); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("lowres_weight_apply: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->stride <= 0 || p->rows <= 0) { set_error("lowres_weight_apply: empty plane"); return X265HIP_EINVAL; }
    if (p->scale < 0 || p->scale > 127 || p->denom < 0 || p->denom > 7 || p->offset < -128 || p->offset > 127) { set_error("lowres_weight_apply: weight out of range"); return X265HIP_EINVAL; }
    WeightApplyArgs a;
    for (int i = 0; i < 4; i++) { a.src[i] = (const uint8_t*)p->src[i]; a.dst[i] = (uint8_t*)p->dst[i]; }
    a.n = (long)p->stride * p->rows; a.depth = p->depth; a.scale = p->scale; a.denom = p->denom; a.offset = p->offset;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)((a.n + 255) / 256);
    if (p->depth == 8) hipLaunchKernelGGL(lowres_weight_apply_kernel<uint8_t>, dim3(g, 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lowres_weight_apply_kernel<uint16_t>, dim3(g, 4), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_aq_energy(const x265hip_aq_energy_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->y || !p->energy || !p->wp) { set_error("aq_energy: NULL operand"); return X265HIP_EINVAL; }
    if ((p->cb == NULL) != (p->cr == NULL)) { set_error("aq_energy: cb and cr go together"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("aq_energy: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->qg_size != 16 && p->qg_size != 8) { set_error("aq_energy: qg_size %d (16 or 8; the larger groups read the 16x16 energies)", p->qg_size); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0) { set_error("aq_energy: empty picture"); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    AqArgs a;
    a.y = (const uint8_t*)p->y; a.cb = (const uint8_t*)p->cb; a.cr = (const uint8_t*)p->cr;
    a.strideB = (long)p->stride * bpp; a.strideCB = (long)p->stride_c * bpp;
    a.width = p->width; a.height = p->height; a.inc = p->qg_size; a.energy = p->energy;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* slots = nullptr;
    X265HIP_TRY(hipMallocAsync((void**)&slots, sizeof(unsigned long long) * 64 * 6, s));
    X265HIP_TRY(hipMemsetAsync(slots, 0, sizeof(unsigned long long) * 64 * 6, s));
    a.slots = slots;
    const int nblk = ((p->width + a.inc - 1) / a.inc) * ((p->height + a.inc - 1) / a.inc);
    const unsigned g = (unsigned)(((long)nblk * 16 + 255) / 256);
    if (bpp == 1) hipLaunchKernelGGL(aq_energy_kernel<uint8_t>, dim3(g), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(aq_energy_kernel<uint16_t>, dim3(g), dim3(256), 0, s, a);
    hipLaunchKernelGGL(aq_totals_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long*)slots, (unsigned long long*)p->wp);
    X265HIP_TRY(hipFreeAsync(slots, s));
    X265HIP_TRY(hipGetLastError());
    return 0;
}

// ---- host side of the adaptive-quantisation pass (slicetype.cpp:508-632, common.cpp:96-103)
#include <cmath>
struct AqExp2Lut
{
    uint8_t v[64];
    AqExp2Lut() { for (int i = 0; i < 64; i++) v[i] = (uint8_t)((std::pow(2.0, i / 64.0) - 1.0) * 256.0 + 0.5); }   // x265_exp2_lut
};
static int aq_exp2fix8(double x)
{
    static const AqExp2Lut table;          // C++11 magic static: initialised once, thread-safe (lookahead workers call concurrently)
    const uint8_t* lut = table.v;
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    return (lut[i & 63] + 256) << (i >> 6) >> 8;
}

// Energy -> QP offset, the host half of the adaptive-quantisation pass.  x265 has two families (encoder/slicetype.cpp:540-633):
//   * the fixed curve of --aq-mode 1: a group's offset is proportional to log2 of its AC energy measured from a pivot that depends on the group
//     size and the bit depth (:606-610);
//   * the picture-adaptive curves of --aq-mode 2 / 3: every group gets the tenth root of its depth-normalised energy, the picture's first and
//     second moment of those roots place the curve (:566-583), mode 3 adds a bias towards dark / flat groups (:590-594).
// Each family is a small object with the pass(es) it needs; the arithmetic keeps the reference's types (float constants inside double
// expressions, sums in block order) because the offsets must come out bit-identical.
namespace {

struct FixedLogCurve                    // --aq-mode 1
{
    double gain; float pivot;
    FixedLogCurve(double aqStrength, int qgSize, int depth) : gain(aqStrength * 1.0397f), pivot((qgSize == 8 ? 11.427f : 14.427f) + 2 * (depth - 8)) {}
    double offset(uint32_t energy) const { return gain * (std::log2((double)(energy > 1 ? energy : 1)) - pivot); }
};

struct PictureAdaptiveCurve             // --aq-mode 2 (biased = false) / 3 (biased = true)
{
    double depthScale, gain = 0, centre = 0, bias = 0;
    float knee;
    bool biased;
    PictureAdaptiveCurve(int qgSize, int depth, bool withBias) : depthScale(1.f / (1 << (2 * (depth - 8)))), knee(qgSize == 8 ? 8.f : 11.f), biased(withBias) {}
    // first pass: the roots themselves (left in `root`), their two moments -> where the curve sits for this picture
    void place(const uint32_t* energy, double* root, int n, double aqStrength)
    {
        double sum = 0, sumSq = 0;
        for (int i = 0; i < n; i++)
        {
            const double r = std::pow(energy[i] * depthScale + 1, 0.1);
            root[i] = r;
            sum += r;
            sumSq += r * r;
        }
        const double mean = sum / n, meanSq = sumSq / n;
        gain = aqStrength * mean;
        centre = mean - 0.5f * (meanSq - knee) / mean;
        bias = aqStrength;
    }
    double offset(double r) const { return biased ? gain * (r - centre) + bias * (1.f - knee / (r * r)) : gain * (r - centre); }
};

} // namespace

extern "C" int x265hip_aq_offsets(const x265hip_aq_offsets_params* p)
{
    if (!p || !p->energy || !p->qp_aq_offset || !p->inv_qscale) { set_error("aq_offsets: NULL operand"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("aq_offsets: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->qg_size != 16 && p->qg_size != 8) { set_error("aq_offsets: qg_size %d", p->qg_size); return X265HIP_EINVAL; }
    if (p->aq_mode < 0 || p->aq_mode > 3) { set_error("aq_offsets: aq_mode %d (0..3; the edge mode and hevcAq are not covered)", p->aq_mode); return X265HIP_EINVAL; }
    if (p->nblocks <= 0) { set_error("aq_offsets: nblocks %d", p->nblocks); return X265HIP_EINVAL; }
    const int n = p->nblocks;
    double* out = p->qp_aq_offset;
    if (p->aq_mode == 0 || p->aq_strength == 0)
    {   // adaptive quantisation off: neutral offsets, unit scale factors (:517-537)
        for (int i = 0; i < n; i++) { out[i] = 0; p->inv_qscale[i] = 256; }
        return 0;
    }
    if (p->aq_mode == 1)
    {
        const FixedLogCurve curve(p->aq_strength, p->qg_size, p->depth);
        for (int i = 0; i < n; i++) out[i] = curve.offset(p->energy[i]);
    }
    else
    {
        PictureAdaptiveCurve curve(p->qg_size, p->depth, p->aq_mode == 3);
        curve.place(p->energy, out, n, p->aq_strength);
        for (int i = 0; i < n; i++) out[i] = curve.offset(out[i]);
    }
    for (int i = 0; i < n; i++) p->inv_qscale[i] = aq_exp2fix8(out[i]);
    return 0;
}

// ---------------------------------------------------------------------------------------------- --hevc-aq: quadrant sums
// The integer half of LookaheadTLD::xPreanalyze (slicetype.cpp:329-404): per partition of part x part samples (clipped at the right /
// bottom edge of the picture) the sum and the sum of squares of its four quadrants, split at half the CLIPPED size.  One wavefront per
// partition: lanes walk its samples in raster order (unit-stride loads), four (sum, sum of squares) pairs per lane, DPP reduction.
struct AqHevcArgs
{
    const uint8_t* y; long strideB;
    int width, height, part, partsW, nparts;
    unsigned long long* sums;
};

template <typename Px>
__global__ void __launch_bounds__(256) aq_hevc_quadrants_kernel(AqHevcArgs a)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= a.nparts) return;
    const int px = (wave % a.partsW) * a.part, py = (wave / a.partsW) * a.part;
    const int cw = min(a.part, a.width - px), ch = min(a.part, a.height - py);
    const int hw = cw >> 1, hh = ch >> 1;
    unsigned long long acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const Px* src = reinterpret_cast<const Px*>(a.y + (long)py * a.strideB) + px;
    const long st = a.strideB / (long)sizeof(Px);
    for (int i = lane; i < cw * ch; i += 64)
    {
        const int by = i / cw, bx = i - by * cw;
        const unsigned v = src[by * st + bx];
        const int quad = (by >= hh) * 2 + (bx >= hw);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            acc[2 * k] += quad == k ? v : 0u;
            acc[2 * k + 1] += quad == k ? (unsigned long long)v * v : 0ull;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        unsigned long long t = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        acc[k] = t;
    }
    if (lane < 8)
    {
        unsigned long long out = acc[0];
#pragma unroll
        for (int k = 1; k < 8; k++) out = lane == k ? acc[k] : out;
        a.sums[(size_t)wave * 8 + lane] = out;
    }
}

extern "C" int x265hip_aq_hevc_quadrants(const x265hip_aq_hevc_params* p, void* stream)
{
    if (!p || !p->y || !p->sums) { set_error("aq_hevc_quadrants: NULL operand"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("aq_hevc_quadrants: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->part != 64 && p->part != 32 && p->part != 16 && p->part != 8) { set_error("aq_hevc_quadrants: partition size %d (64, 32, 16, 8)", p->part); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || (p->width & 1) || (p->height & 1)) { set_error("aq_hevc_quadrants: width / height must be positive and even"); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    AqHevcArgs a;
    const int bpp = p->depth == 8 ? 1 : 2;
    a.y = (const uint8_t*)p->y; a.strideB = (long)p->stride * bpp;
    a.width = p->width; a.height = p->height; a.part = p->part;
    a.partsW = (p->width + p->part - 1) / p->part;
    a.nparts = a.partsW * ((p->height + p->part - 1) / p->part);
    a.sums = (unsigned long long*)p->sums;
    const int blocks = (a.nparts + 3) / 4;
    if (bpp == 1) hipLaunchKernelGGL(aq_hevc_quadrants_kernel<uint8_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(aq_hevc_quadrants_kernel<uint16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

// The double-precision half (host pointers, no device work): activity = 1 + the smallest quadrant variance, the layer's average, and
// xPreanalyzeQp's offset (slicetype.cpp:293-327, 389-404) through the C library's pow / log2 like the reference.
extern "C" int x265hip_aq_hevc_offsets(const x265hip_aq_hevc_offsets_params* p)
{
    if (!p || !p->sums || !p->activity || !p->qp_offset) { set_error("aq_hevc_offsets: NULL operand"); return X265HIP_EINVAL; }
    if (p->part != 64 && p->part != 32 && p->part != 16 && p->part != 8) { set_error("aq_hevc_offsets: partition size %d", p->part); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0) { set_error("aq_hevc_offsets: picture size"); return X265HIP_EINVAL; }
    if (p->qp_adaptation_range < 1.0 || p->qp_adaptation_range > 6.0) { set_error("aq_hevc_offsets: qp_adaptation_range %g out of [1, 6] (param.cpp:1700)", p->qp_adaptation_range); return X265HIP_EINVAL; }
    const int part = p->part, pw = (p->width + part - 1) / part, ph = (p->height + part - 1) / part;
    double dSumAct = 0.0;
    for (int py = 0, i = 0; py < p->height; py += part)
        for (int px = 0; px < p->width; px += part, i++)
        {
            const int cw = std::min(part, p->width - px), ch = std::min(part, p->height - py);
            const uint32_t numPix = (uint32_t)(cw >> 1) * (uint32_t)(ch >> 1);
            double dMinVar = 1.7976931348623158e+308;
            if (numPix)
                for (int k = 0; k < 4; k++)
                {
                    const double dAverage = double(p->sums[(size_t)i * 8 + 2 * k]) / numPix;
                    const double dVariance = double(p->sums[(size_t)i * 8 + 2 * k + 1]) / numPix - dAverage * dAverage;
                    dMinVar = std::min(dMinVar, dVariance);
                }
            else
                dMinVar = 0.0;
            p->activity[i] = 1.0 + dMinVar;
            dSumAct += p->activity[i];
        }
    const double dAvgAct = dSumAct / (uint32_t)(pw * ph);
    if (p->avg_activity) *p->avg_activity = dAvgAct;
    for (int i = 0; i < pw * ph; i++)
    {
        // the reference calls pow(2.0, x); a compiler that rewrites that into exp2(x) (LLVM does) is one ulp off for most x - the base is
        // read through a volatile so that the C library's pow is what runs
        static const volatile double kTwo = 2.0;
        const double dMaxQScale = std::pow(kTwo, p->qp_adaptation_range / 6.0);
        const double dNormAct = (dMaxQScale * p->activity[i] + dAvgAct) / (p->activity[i] + dMaxQScale * dAvgAct);
        p->qp_offset[i] = (std::log2(dNormAct) / std::log2(2.0)) * 6.0;
        if (p->inv_qscale) p->inv_qscale[i] = aq_exp2fix8(p->qp_offset[i]);
    }
    return 0;
}

extern "C" int x265hip_cutree_propagate(const x265hip_cutree_propagate_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->intra_cost || !p->lowres_costs || !p->inv_qscale || !p->mvs0 || !p->ref_cost0) { set_error("cutree_propagate: NULL operand"); return X265HIP_EINVAL; }
    if ((p->mvs1 == NULL) != (p->ref_cost1 == NULL)) { set_error("cutree_propagate: mvs1 and ref_cost1 go together"); return X265HIP_EINVAL; }
    if (p->width_in_cu <= 0 || p->height_in_cu <= 0) { set_error("cutree_propagate: empty picture"); return X265HIP_EINVAL; }
    if (p->bipred_weight < 0 || p->bipred_weight > 64) { set_error("cutree_propagate: bipred_weight %d", p->bipred_weight); return X265HIP_EINVAL; }
    const size_t n = (size_t)p->width_in_cu * p->height_in_cu;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* acc = nullptr;
    X265HIP_TRY(hipMallocAsync((void**)&acc, sizeof(unsigned long long) * 2 * n, s));
    X265HIP_TRY(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * 2 * n, s));
    CuTreeArgs a;
    a.w = p->width_in_cu; a.h = p->height_in_cu;
    a.propagateIn = p->propagate_in; a.intraCost = p->intra_cost; a.lowresCosts = p->lowres_costs; a.invQscale = p->inv_qscale;
    a.mvs[0] = p->mvs0; a.mvs[1] = p->mvs1 ? p->mvs1 : p->mvs0;      // a P picture never uses list 1 (lists used is 0 or 1)
    a.fps = p->fps_factor / 256;
    a.bipred[0] = p->bipred_weight; a.bipred[1] = 64 - p->bipred_weight;
    a.acc[0] = acc; a.acc[1] = acc + n;
    a.refCost[0] = p->ref_cost0; a.refCost[1] = p->ref_cost1;
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(cutree_scatter_kernel, dim3(g), dim3(256), 0, s, a);
    hipLaunchKernelGGL(cutree_finish_kernel, dim3(g, 2), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    X265HIP_TRY(hipFreeAsync(acc, s));
    return 0;
}

// ---- host side of the cuTree step (Lookahead::cuTreeFinish, slicetype.cpp:2889-2937)
extern "C" int x265hip_cutree_finish(const x265hip_cutree_finish_params* p)
{
    if (!p || !p->intra_cost || !p->inv_qscale || !p->propagate_cost || !p->qp_aq_offset || !p->qp_cutree_offset) { set_error("cutree_finish: NULL operand"); return X265HIP_EINVAL; }
    if (p->nblocks <= 0) { set_error("cutree_finish: nblocks %d", p->nblocks); return X265HIP_EINVAL; }
    for (int i = 0; i < p->nblocks; i++)
    {
        const int intracost = (p->intra_cost[i] * p->inv_qscale[i] + 128) >> 8;
        if (!intracost) continue;
        const int propagate = (p->propagate_cost[i] * p->fps_factor_q8 + 128) >> 8;
        const double log2_ratio = std::log2((double)(intracost + propagate)) - std::log2((double)intracost) + p->weight_delta;
        p->qp_cutree_offset[i] = p->qp_aq_offset[i] - p->strength * log2_ratio;
    }
    return 0;
}

// --qg-size 8: the offsets live on the full-resolution 8x8 grid (two by two per lowres block, 2 * width_in_cu per row), costs and the
// averaged invQscaleFactor8x8 on the lowres grid; intra and propagated costs enter at a quarter (slicetype.cpp:2903-2921)
extern "C" int x265hip_cutree_finish_qg8(const x265hip_cutree_finish_params* p, int width_in_cu, int height_in_cu)
{
    if (!p || !p->intra_cost || !p->inv_qscale || !p->propagate_cost || !p->qp_aq_offset || !p->qp_cutree_offset) { set_error("cutree_finish_qg8: NULL operand"); return X265HIP_EINVAL; }
    if (width_in_cu <= 0 || height_in_cu <= 0 || p->nblocks != width_in_cu * height_in_cu) { set_error("cutree_finish_qg8: nblocks %d for %d x %d blocks", p->nblocks, width_in_cu, height_in_cu); return X265HIP_EINVAL; }
    const int w = width_in_cu, full = 2 * w;
    for (int cuY = 0; cuY < height_in_cu; cuY++)
        for (int cuX = 0; cuX < w; cuX++)
        {
            const int cuXY = cuX + cuY * w;
            const int intracost = (p->intra_cost[cuXY] / 4 * p->inv_qscale[cuXY] + 128) >> 8;
            if (!intracost) continue;
            const int propagate = (p->propagate_cost[cuXY] / 4 * p->fps_factor_q8 + 128) >> 8;
            const double log2_ratio = std::log2((double)(intracost + propagate)) - std::log2((double)intracost) + p->weight_delta;
            const int at = cuX * 2 + cuY * w * 4;
            const int idx[4] = { at, at + 1, at + full, at + full + 1 };
            for (int k = 0; k < 4; k++) p->qp_cutree_offset[idx[k]] = p->qp_aq_offset[idx[k]] - p->strength * log2_ratio;
        }
    return 0;
}

extern "C" int x265hip_frame_cost_recalculate_qg8(const x265hip_frame_cost_recalculate_params* p)
{
    if (!p || !p->lowres_costs || !p->qp_cutree_offset || !p->row_satds || !p->score) { set_error("frame_cost_recalculate_qg8: NULL operand"); return X265HIP_EINVAL; }
    if (p->width_in_cu <= 0 || p->height_in_cu <= 0) { set_error("frame_cost_recalculate_qg8: empty picture"); return X265HIP_EINVAL; }
    const int w = p->width_in_cu, h = p->height_in_cu, full = 2 * w;
    int64_t score = 0;
    for (int cuy = h - 1; cuy >= 0; cuy--)
    {
        int row = 0;
        for (int cux = w - 1; cux >= 0; cux--)
        {
            const int at = cux * 2 + cuy * w * 4;
            const double qp_adj = (p->qp_cutree_offset[at] + p->qp_cutree_offset[at + 1] + p->qp_cutree_offset[at + full] + p->qp_cutree_offset[at + full + 1]) / 4;
            const int cuCost = ((p->lowres_costs[cux + cuy * w] & 0x3fff) * aq_exp2fix8(qp_adj) + 128) >> 8;
            row += cuCost;
            if ((cuy > 0 && cuy < h - 1 && cux > 0 && cux < w - 1) || w <= 2 || h <= 2) score += cuCost;
        }
        p->row_satds[cuy] = row;
    }
    *p->score = score;
    return 0;
}

// cuTree with --hevc-aq (Lookahead::computeCUTreeQpOffset, slicetype.cpp:2749-2887, quantisation groups of 16 or more): per partition
// of one layer the mean, over the 16x16 blocks it covers, of log2(intra + propagate) - log2(intra) + weight_delta, scaled by the cuTree
// strength and taken off the layer's dQpOffset.  Blocks whose scaled intra cost is 0 are NOT skipped here (the reference does not
// either): their term is +inf or NaN and so is the partition's offset.
extern "C" int x265hip_cutree_finish_hevc_aq(const x265hip_cutree_finish_hevc_params* p)
{
    if (!p || !p->intra_cost || !p->inv_qscale || !p->propagate_cost || !p->qp_offset || !p->cutree_offset) { set_error("cutree_finish_hevc_aq: NULL operand"); return X265HIP_EINVAL; }
    if (p->part != 64 && p->part != 32 && p->part != 16) { set_error("cutree_finish_hevc_aq: partition size %d (64, 32, 16)", p->part); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || p->blocks_in_row <= 0) { set_error("cutree_finish_hevc_aq: picture size"); return X265HIP_EINVAL; }
    const uint32_t part = (uint32_t)p->part, w = (uint32_t)p->width, h = (uint32_t)p->height, loopIncr = 16;
    const uint32_t nw = (w + part - 1) / part, nh = (h + part - 1) / part;
    const double* pcQP = p->qp_offset;
    double* pcCuTree = p->cutree_offset;
    for (uint32_t y = 0; y < nh; y++)
        for (uint32_t x = 0; x < nw; x++, pcQP++, pcCuTree++)
        {
            const uint32_t block_x = x * part, block_y = y * part;
            uint32_t blockXY = 0;
            double log2_ratio = 0;
            for (uint32_t yy = block_y; yy < block_y + part && yy < h; yy += loopIncr)
                for (uint32_t xx = block_x; xx < block_x + part && xx < w; xx += loopIncr)
                {
                    const uint32_t idx = (yy / loopIncr) * (uint32_t)p->blocks_in_row + xx / loopIncr;
                    const int intraCost = (p->intra_cost[idx] * p->inv_qscale[idx] + 128) >> 8;
                    const int propagateCost = (p->propagate_cost[idx] * p->fps_factor_q8 + 128) >> 8;
                    log2_ratio += (std::log2((double)(intraCost + propagateCost)) - std::log2((double)intraCost) + p->weight_delta);
                    blockXY++;
                }
            const double qp_offset = (p->strength * log2_ratio) / blockXY;
            *pcCuTree = *pcQP - qp_offset;
        }
    return 0;
}

extern "C" int x265hip_frame_cost_recalculate(const x265hip_frame_cost_recalculate_params* p)
{
    if (!p || !p->lowres_costs || !p->qp_cutree_offset || !p->row_satds || !p->score) { set_error("frame_cost_recalculate: NULL operand"); return X265HIP_EINVAL; }
    if (p->width_in_cu <= 0 || p->height_in_cu <= 0) { set_error("frame_cost_recalculate: empty picture"); return X265HIP_EINVAL; }
    const int w = p->width_in_cu, h = p->height_in_cu;
    int64_t score = 0;
    for (int cuy = h - 1; cuy >= 0; cuy--)
    {
        int row = 0;
        for (int cux = w - 1; cux >= 0; cux--)
        {
            const int cuxy = cux + cuy * w;
            const int cuCost = ((p->lowres_costs[cuxy] & 0x3fff) * aq_exp2fix8(p->qp_cutree_offset[cuxy]) + 128) >> 8;
            row += cuCost;
            if ((cuy > 0 && cuy < h - 1 && cux > 0 && cux < w - 1) || w <= 2 || h <= 2) score += cuCost;
        }
        p->row_satds[cuy] = row;
    }
    *p->score = score;
    return 0;
}
