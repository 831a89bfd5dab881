// subpel_kernels.hip - sub-pel motion refinement for every PU of every CTU, on gfx950.
//
// Reference semantics: MotionEstimate::motionEstimate after the integer search
// (source/encoder/motion.cpp:1448-1561, non-lowres branch: SubpelWorkload table :48-58, square1 pattern :66,
// half-pel then quarter-pel iterations, COPY2_IF_LT strict-less) and MotionEstimate::subpelCompare
// (:1571-1664, luma: pu[].luma_hpp / luma_vpp / luma_hvpp into a blockwidth-stride buffer, then pu[].sad or
// pu[].satd against the PU source copy).  Interpolation arithmetic: source/common/ipfilter.cpp:79-118,
// :164-203, :120-162 + :319-369 (hps with row extension then vertical sp), SATD source/common/pixel.cpp:210-297.
//
// Mapping: one workgroup per (CTU, PU level).  The CTU source block and, per PU, the reference patch around its
// integer motion vector (+-3 pixels of drift + 8-tap apron) are staged in LDS once; every candidate of an
// iteration (4 or 8 directions) of every PU is evaluated concurrently, one thread per (PU, candidate, 4x4 tile): the thread
// interpolates its 16 samples straight from the LDS patch (hv: 11 horizontally filtered rows kept in
// registers), takes the 4x4 Hadamard (or SAD) in registers and adds its partial cost to an LDS bin.  The
// serial decision loop of the reference then runs in one thread per PU on the reduced costs, so no
// host round trip sits between the candidates - this is SURVEY section 8(f) item 1 for the sub-pel part.
#include "common.h"

namespace x265hip {

struct SubpelArgs
{
    const uint8_t* fenc; long fencStrideB;
    const uint8_t* fref; long frefStrideB;
    int ctusW, range, depth;
    const unsigned long long* bestIn;
    const uint16_t* costQ; int qoff;
    int hpelIters, hpelDirs, qpelIters, qpelDirs, hpelSatd;
    int2* out;                       // {cost, qx | qy << 16} per PU, [ctu][85]
    const uint8_t* planes; long planeBytes;      // phase planes of fref (x265hip_phase_planes), sample (0,0) of phase 1; NULL = interpolate
};

__constant__ int16_t kSpLumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ int kSpSquare1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };

constexpr int SP_MARGIN = 7;                        // patch origin = integer mv - 7: 3 px of drift + 3 px of left apron + 1

__device__ __forceinline__ int sp_clip16(int v, int maxVal)
{
    const int16_t s = (int16_t)v;
    return s < 0 ? 0 : (s > maxVal ? maxVal : s);
}

struct SpShared                                     // per-PU search state of one workgroup
{
    int cost[64][9];
    int best[64], qx[64], qy[64], ix[64], iy[64], state[64];
};

template <typename Px, int LEVEL> struct SpGeom
{
    static constexpr int BPP = sizeof(Px);
    static constexpr int N = 8 << LEVEL, NPU = 64 >> (2 * LEVEL);
    static constexpr int TSHIFT = 2 * LEVEL + 2, NTILES = 1 << TSHIFT, TPR = N >> 2;   // 4x4 tiles per PU; NPU * NTILES = 256
    static constexpr int PW = N + 2 * SP_MARGIN + 1;                                     // patch width = height (pixels)
    static constexpr int PITCH = BPP == 1 ? (PW + 3) & ~3 : (PW + 1) & ~1;               // row pitch in pixels, dword multiple
    static constexpr int DWR = PITCH * BPP / 4;                                          // dwords per patch row
    static constexpr int PSZ = PW * PITCH;
    static constexpr int LBASE = LEVEL == 0 ? 0 : (LEVEL == 1 ? 64 : (LEVEL == 2 ? 80 : 84));
};

// One workgroup = one (CTU, PU level): 256 threads, thread t owns 4x4 tile (t % NTILES) of PU (t / NTILES) for the
// whole search, its 16 source pixels in registers.  Each pass of an evaluation round handles ONE candidate for
// all PUs, so a wavefront mostly runs a single interpolation case.
// PL: the candidates' samples are READ from the reference picture's phase planes (x265hip_phase_planes: the same luma_hpp / luma_vpp /
// luma_hvpp samples, computed once per picture) instead of being interpolated per candidate tile: no LDS patches, four dword loads
// per candidate tile.
template <typename Px, int LEVEL, bool PL>
__device__ __forceinline__ void subpel_level(const SubpelArgs& a, uint8_t* smemRaw, SpShared& sh)
{
    typedef SpGeom<Px, LEVEL> G;
    constexpr int BPP = G::BPP, N = G::N, NPU = G::NPU, PW = G::PW, PITCH = G::PITCH;
    Px* patches = reinterpret_cast<Px*>(smemRaw);

    const int ctu = xcd_swizzle(blockIdx.x, gridDim.x);
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int tid = threadIdx.x;
    const int R = a.range, NC = 2 * R + 1;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;

    if (tid < NPU)
    {
        const unsigned long long key = a.bestIn[(size_t)ctu * 85 + G::LBASE + tid];
        const int idx = (int)(key & 0xffffffffu);
        const int imx = (idx % NC) - R, imy = (idx / NC) - R;
        sh.ix[tid] = imx; sh.iy[tid] = imy;
        sh.qx[tid] = imx * 4; sh.qy[tid] = imy * 4;
        const int c = (int)(key >> 32);
        sh.best[tid] = c;
        sh.state[tid] = c != 0;                            // zero residual: skip refinement (motion.cpp:1464-1469)
    }
    const int pu = tid >> G::TSHIFT, tile = tid & (G::NTILES - 1);
    const int bxz = (pu & 1) | ((pu >> 1) & 2) | ((pu >> 2) & 4), byz = ((pu >> 1) & 1) | ((pu >> 2) & 2) | ((pu >> 3) & 4);
    const int ty = tile / G::TPR, tx = tile % G::TPR;
    int src[4][4];
    {
        const Px* fe = reinterpret_cast<const Px*>(a.fenc + (long)(cy + byz * N + ty * 4) * a.fencStrideB) + (cx + bxz * N + tx * 4);
        const long fst = a.fencStrideB / BPP;
        // the tile's rows as one dword (two for 16-bit samples) each: sixteen single-sample loads were a quarter of the kernel's load instructions
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            const uint8_t* rp = reinterpret_cast<const uint8_t*>(fe + y * fst);
            if (BPP == 1) { const uint32_t w = ld_u32(rp); src[y][0] = w & 0xff; src[y][1] = (w >> 8) & 0xff; src[y][2] = (w >> 16) & 0xff; src[y][3] = w >> 24; }
            else { const uint32_t w0 = ld_u32(rp), w1 = ld_u32(rp + 4); src[y][0] = w0 & 0xffff; src[y][1] = w0 >> 16; src[y][2] = w1 & 0xffff; src[y][3] = w1 >> 16; }
        }
    }
    // The tile's rows as two packed int16 pairs each - columns (0, 2) / (1, 3) for 8-bit samples (what two masks make of a loaded dword),
    // (0, 1) / (2, 3) for 16-bit ones (the dwords as loaded): the difference, the 4x4 Hadamard and the absolute sums below run on
    // packed pairs (v_pk_add_i16 / v_pk_sub_i16 / v_pk_max_i16).  A 4-point Hadamard's absolute sum does not depend on the order of its
    // inputs (its three non-constant rows are the three ways of splitting four items two and two), so any pairing gives the reference's
    // value; every intermediate fits int16 up to 12-bit samples (8 * 4095 = 32760).  The kernel is VALU-bound (profiles/r03_inst_counters.txt).
    typedef short sp_v2s __attribute__((ext_vector_type(2)));
    auto pk = [](const int lo, const int hi) -> sp_v2s { return __builtin_bit_cast(sp_v2s, ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16)); };
    sp_v2s srcP[4], srcQ[4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        srcP[y] = BPP == 1 ? pk(src[y][0], src[y][2]) : pk(src[y][0], src[y][1]);
        srcQ[y] = BPP == 1 ? pk(src[y][1], src[y][3]) : pk(src[y][2], src[y][3]);
    }
    __syncthreads();
    if (!PL)
    {   // stage every PU's reference patch, one (possibly unaligned) dword per item
        constexpr int ITEMS = NPU * PW * G::DWR;
        uint32_t* pdw = reinterpret_cast<uint32_t*>(smemRaw);
#pragma unroll 4
        for (int i = tid; i < ITEMS; i += 256)
        {
            const int q = i / (PW * G::DWR), r = i % (PW * G::DWR), y = r / G::DWR, c = r % G::DWR;
            const int qbx = (q & 1) | ((q >> 1) & 2) | ((q >> 2) & 4), qby = ((q >> 1) & 1) | ((q >> 2) & 2) | ((q >> 3) & 4);
            const uint8_t* rf = a.fref + (long)(cy + qby * N + sh.iy[q] - SP_MARGIN + y) * a.frefStrideB
                                + (long)(cx + qbx * N + sh.ix[q] - SP_MARGIN) * BPP + 4 * c;
            pdw[i] = ld_u32(rf);
        }
    }
    __syncthreads();

    const Px* patch = patches + pu * G::PSZ;
    const int pox = SP_MARGIN + tx * 4 - sh.ix[pu], poy = SP_MARGIN + ty * 4 - sh.iy[pu];

    // Packed interpolation of the thread's 4 x 4 tile out of the LDS patch (same identities as csrc/interp_kernels.hip):
    // horizontal taps by v_dot4_i32_i8 on (pixel - 128) bytes (8-bit) or v_dot2_i32_i16 on pixel pairs (16-bit), vertical
    // taps by v_dot2_i32_i16 on (row r, row r + 1) pairs built with v_perm_b32; all sums are the exact int32 values.
    typedef short v2i16 __attribute__((ext_vector_type(2)));
    auto dot2 = [](const uint32_t x, const uint32_t y, const int c) -> int
    { return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2i16, x), __builtin_bit_cast(v2i16, y), c, false); };
    auto sel3 = [](const int f, const uint32_t a1, const uint32_t a2, const uint32_t a3) -> uint32_t { return f == 1 ? a1 : (f == 2 ? a2 : a3); };
    const uint8_t* pbytes = reinterpret_cast<const uint8_t*>(patch);
    // 4 horizontal sums (no rounding) of the row whose sample (ox - 3) sits at byte pointer rp
    auto hrow = [&](const uint8_t* rp, const int xf, int (&out)[4])
    {
        if (BPP == 1)
        {
            const uint32_t c03 = sel3(xf, 0x3af604ffu, 0x28f504ffu, 0x11fb0100u), c47 = sel3(xf, 0x0001fb11u, 0xff04f528u, 0xff04f63au);
            const uint32_t w0 = ld_u32(rp) ^ 0x80808080u, w1 = ld_u32(rp + 4) ^ 0x80808080u, w2 = ld_u32(rp + 8) ^ 0x80808080u;
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                const uint32_t lo = x ? __builtin_amdgcn_alignbyte(w1, w0, x) : w0, hi = x ? __builtin_amdgcn_alignbyte(w2, w1, x) : w1;
                out[x] = __builtin_amdgcn_sdot4((int)hi, (int)c47, __builtin_amdgcn_sdot4((int)lo, (int)c03, 8192, false), false);
            }
        }
        else
        {
            // taps as int16 pairs (c0,c1) (c2,c3) (c4,c5) (c6,c7)
            const uint32_t cp[4] = { sel3(xf, 0x0004ffffu, 0x0004ffffu, 0x00010000u), sel3(xf, 0x003afff6u, 0x0028fff5u, 0x0011fffbu),
                                     sel3(xf, 0xfffb0011u, 0xfff50028u, 0xfff6003au), sel3(xf, 0x00000001u, 0xffff0004u, 0xffff0004u) };
            uint32_t dd[6];
#pragma unroll
            for (int k = 0; k < 6; k++) dd[k] = ld_u32(rp + 4 * k);
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int sacc = 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const int k = (x >> 1) + j;
                    sacc = dot2((x & 1) ? __builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 2) : dd[k], cp[j], sacc);
                }
                out[x] = sacc;
            }
        }
    };
    // PL: byte address of the tile's sample (0,0) at integer mv (0,0), relative to fref / to a phase plane's sample (0,0)
    const long tileOff = (long)(cy + byz * N + ty * 4) * a.frefStrideB + (long)(cx + bxz * N + tx * 4) * BPP;
    auto tile_cost = [&](const int qx, const int qy, const bool useSatd) -> int
    {
        const int ox = (qx >> 2) + pox, oy = (qy >> 2) + poy;
        const int xf = qx & 3, yf = qy & 3;
        int d[4][4];
        sp_v2s dP[4], dQ[4];                                 // the prediction's rows (then the differences) as packed pairs
        if (PL)
        {
            const int ph = yf * 4 + xf;
            const uint8_t* base = (ph ? a.planes + (long)(ph - 1) * a.planeBytes : a.fref) + tileOff + (long)(qy >> 2) * a.frefStrideB + (long)(qx >> 2) * BPP;
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                const uint8_t* rp = base + (long)y * a.frefStrideB;
                if (BPP == 1)
                {
                    const uint32_t w = ld_u32(rp);
                    dP[y] = __builtin_bit_cast(sp_v2s, w & 0x00ff00ffu);
                    dQ[y] = __builtin_bit_cast(sp_v2s, (w >> 8) & 0x00ff00ffu);
                }
                else
                {
                    dP[y] = __builtin_bit_cast(sp_v2s, ld_u32(rp));
                    dQ[y] = __builtin_bit_cast(sp_v2s, ld_u32(rp + 4));
                }
            }
        }
        else if (!(xf | yf))
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = patch[(oy + y) * PITCH + ox + x];
        }
        else if (!yf)
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                int hs[4];
                hrow(pbytes + ((oy + y) * PITCH + ox - 3) * BPP, xf, hs);
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = sp_clip16((hs[x] + 32) >> 6, maxVal);
            }
        }
        else
        {
            const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
            const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
            const uint32_t cv[4] = { sel3(yf, 0x0004ffffu, 0x0004ffffu, 0x00010000u), sel3(yf, 0x003afff6u, 0x0028fff5u, 0x0011fffbu),
                                     sel3(yf, 0xfffb0011u, 0xfff50028u, 0xfff6003au), sel3(yf, 0x00000001u, 0xffff0004u, 0xffff0004u) };
            uint32_t pairs[10][4];                           // (row r, row r + 1) at the 4 columns, rows oy - 3 .. oy + 7
            if (!xf)
            {
                uint32_t raw[11][2];
#pragma unroll
                for (int t = 0; t < 11; t++)
                {
                    const uint8_t* rp = pbytes + ((oy + t - 3) * PITCH + ox) * BPP;
                    raw[t][0] = ld_u32(rp);
                    raw[t][1] = BPP == 2 ? ld_u32(rp + 4) : 0;
                }
#pragma unroll
                for (int t = 0; t < 10; t++)
#pragma unroll
                    for (int x = 0; x < 4; x++)
                        pairs[t][x] = BPP == 1 ? __builtin_amdgcn_perm(raw[t + 1][0], raw[t][0], 0x0c000c00u | (uint32_t)x | ((uint32_t)(4 + x) << 16))
                                               : __builtin_amdgcn_perm(raw[t + 1][x >> 1], raw[t][x >> 1], (x & 1) ? 0x07060302u : 0x05040100u);
            }
            else
            {
                int im[11][4];
#pragma unroll
                for (int t = 0; t < 11; t++)
                {
                    int hs[4];
                    hrow(pbytes + ((oy + t - 3) * PITCH + ox - 3) * BPP, xf, hs);
#pragma unroll
                    for (int x = 0; x < 4; x++) im[t][x] = (hs[x] + offPS) >> shiftPS;
                }
#pragma unroll
                for (int t = 0; t < 10; t++)
#pragma unroll
                    for (int x = 0; x < 4; x++) pairs[t][x] = __builtin_amdgcn_perm((uint32_t)im[t + 1][x], (uint32_t)im[t][x], 0x05040100u);
            }
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int sum = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) sum = dot2(pairs[y + 2 * j][x], cv[j], sum);
                    d[y][x] = xf ? sp_clip16((sum + offSP) >> shiftSP, maxVal) : sp_clip16((sum + 32) >> 6, maxVal);
                }
        }
        if (!PL)
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                dP[y] = BPP == 1 ? pk(d[y][0], d[y][2]) : pk(d[y][0], d[y][1]);
                dQ[y] = BPP == 1 ? pk(d[y][1], d[y][3]) : pk(d[y][2], d[y][3]);
            }
        }
#pragma unroll
        for (int y = 0; y < 4; y++) { dP[y] = srcP[y] - dP[y]; dQ[y] = srcQ[y] - dQ[y]; }
        // |lo| + |hi| of a packed pair / max(|lo|, |hi|)
        auto pabs = [](const sp_v2s v) -> sp_v2s { return __builtin_elementwise_max(v, (sp_v2s)(-v)); };
        auto lo_plus_hi = [](const sp_v2s v) -> int { const uint32_t u = __builtin_bit_cast(uint32_t, v); return (int)((u & 0xffffu) + (u >> 16)); };
        auto lo_max_hi = [](const sp_v2s v) -> int { const uint32_t u = __builtin_bit_cast(uint32_t, v); const uint32_t l = u & 0xffffu, h = u >> 16; return (int)(l > h ? l : h); };
        int acc = 0;
        if (!useSatd)
        {
#pragma unroll
            for (int y = 0; y < 4; y++) acc += lo_plus_hi(pabs(dP[y])) + lo_plus_hi(pabs(dQ[y]));
            return acc;
        }
        // vertical 4-point Hadamard of both pair columns (packed adds), then per transformed row k: s = P + Q, t = P - Q hold the two
        // half-sums of the horizontal transform and |a + b| + |a - b| = 2 max(|a|, |b|): the row's four absolute values sum to
        // 2 (max(|s.lo|, |s.hi|) + max(|t.lo|, |t.hi|)) - the reference's (sum >> 1) is the sum of those maxima
        sp_v2s pv[4], qv[4];
        {
            const sp_v2s a0 = dP[0] + dP[1], b0 = dP[0] - dP[1], c0 = dP[2] + dP[3], e0 = dP[2] - dP[3];
            pv[0] = a0 + c0; pv[1] = b0 + e0; pv[2] = a0 - c0; pv[3] = b0 - e0;
            const sp_v2s a1 = dQ[0] + dQ[1], b1 = dQ[0] - dQ[1], c1 = dQ[2] + dQ[3], e1 = dQ[2] - dQ[3];
            qv[0] = a1 + c1; qv[1] = b1 + e1; qv[2] = a1 - c1; qv[3] = b1 - e1;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            acc += lo_max_hi(pabs(pv[k] + qv[k])) + lo_max_hi(pabs(pv[k] - qv[k]));
        return acc;
    };

    auto mvcost = [&](const int qx, const int qy) { return (int)a.costQ[qx + a.qoff] + (int)a.costQ[qy + a.qoff]; };

    // PU state: 0 = zero residual at the integer mv, refinement skipped; 1 = searching; 2 = the current phase
    // ended for this PU (the reference's `break`), waiting for the next phase.
    // The reference's sequence [SATD re-measure if hpelSatd] hpel iterations [re-measure otherwise] qpel iterations
    // is run as one loop of rounds so the interpolation code exists once.
    const bool hs = a.hpelSatd != 0;
    const int nH = a.hpelIters, remAt = hs ? 0 : nH, total = 1 + nH + a.qpelIters;
    if (G::NTILES == 256) { if (tid < 9) sh.cost[0][tid] = 0; __syncthreads(); }
    for (int rd = 0; rd < total; rd++)
    {
        const bool isRem = rd == remAt;
        const bool isQ = rd > nH;
        if (rd == nH + (hs ? 1 : 0))
        {
            if (tid < NPU && sh.state[tid] == 2) sh.state[tid] = 1;      // next phase: every refined PU searches again
            __syncthreads();
        }
        const int first = isRem ? 0 : 1, last = isRem ? 0 : (isQ ? a.qpelDirs : a.hpelDirs);
        const int step = isRem ? 0 : (isQ ? 1 : 2);
        const bool useSatd = isRem || isQ || hs;
        // evaluate: for every searching PU, candidates best + square1[first..last] * step -> sh.cost[pu][first..last]
        {
            const bool on = sh.state[pu] == 1;
            const int qx0 = sh.qx[pu], qy0 = sh.qy[pu];
            for (int c = first; c <= last; c++)
            {
                int v = on ? tile_cost(qx0 + kSpSquare1[c][0] * step, qy0 + kSpSquare1[c][1] * step, useSatd) : 0;
                // sum over the PU's NTILES consecutive threads (all lanes take part)
                v = quad_sum(v);
                if (G::NTILES >= 16) v = row_sum_of_quads(v);
                if (G::NTILES >= 64) v = wave_sum_of_rows(v);
                if (G::NTILES == 256) { if ((tid & 63) == 0) atomicAdd(&sh.cost[pu][c], v); }
                else if ((tile & (G::NTILES < 64 ? G::NTILES - 1 : 63)) == 0) sh.cost[pu][c] = v;
            }
            __syncthreads();
        }
        if (tid < NPU && sh.state[tid] == 1)
        {
            const int qx0 = sh.qx[tid], qy0 = sh.qy[tid];
            if (isRem)
                sh.best[tid] = sh.cost[tid][0] + mvcost(qx0, qy0);      // SATD of the current best replaces the SAD-based cost
            else
            {
                // COPY2_IF_LT: strict less, first direction wins (motion.cpp:1520-1531)
                int bcost = sh.best[tid], bdir = 0;
                for (int i = 1; i <= last; i++)
                {
                    const int c = sh.cost[tid][i] + mvcost(qx0 + kSpSquare1[i][0] * step, qy0 + kSpSquare1[i][1] * step);
                    if (c < bcost) { bcost = c; bdir = i; }
                }
                sh.best[tid] = bcost;
                if (bdir) { sh.qx[tid] = qx0 + kSpSquare1[bdir][0] * step; sh.qy[tid] = qy0 + kSpSquare1[bdir][1] * step; }
                else sh.state[tid] = 2;
            }
        }
        if (G::NTILES == 256) { __syncthreads(); if (tid < 9) sh.cost[0][tid] = 0; }     // the 64x64 PU accumulates across wavefronts
        __syncthreads();
    }
    if (tid < NPU)
    {
        int bcost = sh.best[tid];
        if (sh.state[tid] == 0) bcost = mvcost(sh.qx[tid], sh.qy[tid]);     // zero-residual PUs return the mv cost only
        a.out[(size_t)ctu * 85 + G::LBASE + tid] = make_int2(bcost, (sh.qx[tid] & 0xffff) | (sh.qy[tid] << 16));
    }
}

template <typename Px, bool PL = false>
__global__ void __launch_bounds__(256) subpel_refine_kernel(SubpelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemRaw[];
    __shared__ SpShared sh;
    switch (blockIdx.y)                                   // all four PU levels of a CTU run concurrently
    {
    case 0: subpel_level<Px, 0, PL>(a, smemRaw, sh); break;
    case 1: subpel_level<Px, 1, PL>(a, smemRaw, sh); break;
    case 2: subpel_level<Px, 2, PL>(a, smemRaw, sh); break;
    default: subpel_level<Px, 3, PL>(a, smemRaw, sh); break;
    }
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_subpel_refine(const x265hip_subpel_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->fref || !p->best_in || !p->cost_q || !p->out) { set_error("subpel_refine: NULL operand"); return X265HIP_EINVAL; }
    if ((p->width & 63) || (p->height & 63) || p->width <= 0 || p->height <= 0) { set_error("subpel_refine: width/height must be multiples of 64"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("subpel_refine: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->subme < 0 || p->subme > 7) { set_error("subpel_refine: subme %d out of [0,7]", p->subme); return X265HIP_EINVAL; }
    static const int wl[8][5] = { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
                                  { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };   // motion.cpp:48-58
    const int bpp = p->depth == 8 ? 1 : 2;
    SubpelArgs a;
    a.fenc = (const uint8_t*)p->fenc; a.fencStrideB = (long)p->fenc_stride * bpp;
    a.fref = (const uint8_t*)p->fref; a.frefStrideB = (long)p->fref_stride * bpp;
    a.ctusW = p->width / 64; a.range = p->range; a.depth = p->depth;
    a.bestIn = (const unsigned long long*)p->best_in; a.costQ = p->cost_q; a.qoff = p->qoff;
    a.hpelIters = wl[p->subme][0]; a.hpelDirs = wl[p->subme][1]; a.qpelIters = wl[p->subme][2]; a.qpelDirs = wl[p->subme][3]; a.hpelSatd = wl[p->subme][4];
    a.out = (int2*)p->out;
    a.planes = (const uint8_t*)p->phase_planes; a.planeBytes = (long)p->phase_plane_samples * bpp;
    if (a.planes && p->phase_plane_samples <= 0) { set_error("subpel_refine: phase_plane_samples"); return X265HIP_EINVAL; }
    const int nctu = a.ctusW * (p->height / 64);
    hipStream_t s = (hipStream_t)stream;
    // one launch: grid.y = PU level; LDS sized for level 0 (64 patches of 23 rows), the largest
    const size_t lds = (size_t)(bpp == 1 ? SpGeom<uint8_t, 0>::PSZ : SpGeom<uint16_t, 0>::PSZ) * 64 * bpp;
    static_assert(SpGeom<uint8_t, 0>::PSZ * 64 >= SpGeom<uint8_t, 1>::PSZ * 16 && SpGeom<uint8_t, 0>::PSZ * 64 >= SpGeom<uint8_t, 2>::PSZ * 4
                  && SpGeom<uint8_t, 0>::PSZ * 64 >= SpGeom<uint8_t, 3>::PSZ, "level 0 is the largest");
    static_assert(SpGeom<uint16_t, 0>::PSZ * 64 >= SpGeom<uint16_t, 1>::PSZ * 16 && SpGeom<uint16_t, 0>::PSZ * 64 >= SpGeom<uint16_t, 2>::PSZ * 4
                  && SpGeom<uint16_t, 0>::PSZ * 64 >= SpGeom<uint16_t, 3>::PSZ, "level 0 is the largest");
    if (a.planes)
    {
        if (p->depth == 8) hipLaunchKernelGGL((subpel_refine_kernel<uint8_t, true>), dim3(nctu, 4), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((subpel_refine_kernel<uint16_t, true>), dim3(nctu, 4), dim3(256), 0, s, a);
    }
    else if (p->depth == 8)
        hipLaunchKernelGGL(subpel_refine_kernel<uint8_t>, dim3(nctu, 4), dim3(256), lds, s, a);
    else
    {
        static bool attr = false;
        if (!attr) { X265HIP_TRY(hipFuncSetAttribute((const void*)subpel_refine_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
        hipLaunchKernelGGL(subpel_refine_kernel<uint16_t>, dim3(nctu, 4), dim3(256), lds, s, a);
    }
    X265HIP_TRY(hipGetLastError());
    return 0;
}
