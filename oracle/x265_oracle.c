/* oracle/x265_oracle.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker / the
 * reported CPU baseline.  The product (x265-yuuki-asuna_amd/csrc) never links it.
 *
 * A plain-C restatement of the x265 3.5 block-primitive hot path (the C reference
 * behind the EncoderPrimitives table).  Every function cites the reference file:line
 * whose behaviour it follows; the code itself is written from the algorithm (HEVC
 * integer transforms / interpolation / intra / SAO / deblock definitions plus the
 * reference-specific rounding, casts and side effects), not copied.
 *
 * Pinning: the restatement is checked bit-for-bit against the REAL reference compiled
 * from /root/reference by oracle/Makefile (oracle/_ref/libx265ref{8,10}.so) in
 * tests/test_oracle_vs_reference.py, and against the committed golden vectors in
 * tests/golden/ (generated from that reference build by tools/gen_golden.py).
 *
 * Build: compiled once per bit depth with -DX265HIP_DEPTH=8|10|12; all exported names
 * carry a _d<depth> suffix so the three objects link into one libx265oracle.so.
 */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"

#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef x265hip_pixel pixel;
typedef x265hip_sse_t sse_t;

#define DEPTH        X265HIP_DEPTH
#define PIXEL_MAX    ((1 << DEPTH) - 1)
#define FENC_STRIDE  64                 /* common.h:70 */
#define MAX_CU       64                 /* common.h:257 */
#define IF_PREC      14                 /* constants.h:68 IF_INTERNAL_PREC */
#define IF_FPREC     6                  /* constants.h:69 IF_FILTER_PREC */
#define IF_OFFS      (1 << (IF_PREC - 1))

#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

static inline int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline pixel clip_pixel(int v) { return (pixel)clip3i(0, PIXEL_MAX, v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int sgn(int v) { return (v > 0) - (v < 0); }

/* ------------------------------------------------------------------ tables */
/* HEVC core transform matrix.  T32[k][n] = s(m) * c[m'], m = k(2n+1) folded by the
 * cosine symmetries; the 33 magnitudes are the standard's 32-point basis values.
 * The 16/8/4-point matrices are its even-row decimations.  Must equal the reference's
 * g_t4/g_t8/g_t16/g_t32 (constants.cpp:270-344) - checked in tests. */
static const int16_t kBasis[33] = {
    64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
    64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
static int16_t T32[32][32];
static int tablesReady;

static int basis_at(int m)
{
    m &= 127;
    if (m <= 32) return kBasis[m];
    if (m <= 64) return -kBasis[64 - m];
    if (m <= 96) return -kBasis[m - 64];
    return kBasis[128 - m];
}

static void init_tables(void)
{
    if (tablesReady) return;
    for (int k = 0; k < 32; k++)
        for (int n = 0; n < 32; n++)
            T32[k][n] = (int16_t)basis_at(k * (2 * n + 1));
    tablesReady = 1;
}

/* row k, column n of the N-point matrix (N = 4, 8, 16, 32) */
static inline int tcoef(int N, int k, int n) { return T32[k * (32 / N)][n]; }

const int16_t* EXPORT(x265oracle_dct_matrix32)(void) { init_tables(); return &T32[0][0]; }

/* constants.cpp:250-268 - HEVC interpolation taps (spec tables 8-11 / 8-12) */
static const int16_t kLumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int16_t kChromaTaps[8][4] = {
    { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
/* constants.cpp:561 - which TU sizes (bitmask of size) get smoothed neighbours, per mode */
static const uint8_t kIntraFilterFlags[35] = {
    0x38, 0x00,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38 };

/* ================================================================== a1/a2: SAD */
/* pixel.cpp:40-55 */
static inline int sad_wh(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int w, int h)
{
    int acc = 0;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++)
            acc += iabs((int)a[x] - (int)b[x]);
    return acc;
}

/* pixel.cpp:74-119: the encode block always has stride FENC_STRIDE; all refs share one stride */
static inline void sad_xn_wh(const pixel* fenc, const pixel* const* refs, int n, intptr_t rs, int32_t* res, int w, int h)
{
    for (int i = 0; i < n; i++)
        res[i] = sad_wh(fenc, FENC_STRIDE, refs[i], rs, w, h);
}

/* ================================================================== a3: ADS (SEA pre-filter) */
/* pixel.cpp:121-165.  nsum = 4/2/1 chosen per PU size at pixel.cpp:1105-1129; the loop index is an
 * int16_t in the reference (width never exceeds the search width). */
static inline int ads_n(int nsum, int lx, int* encDC, uint32_t* sums, int delta, uint16_t* costMvX,
                        int16_t* mvs, int width, int thresh)
{
    int nmv = 0;
    for (int i = 0; i < width; i++, sums++)
    {
        long a = labs((long)encDC[0] - (long)sums[0]);
        if (nsum == 4)
            a += labs((long)encDC[1] - (long)sums[lx >> 1]) + labs((long)encDC[2] - (long)sums[delta])
               + labs((long)encDC[3] - (long)sums[delta + (lx >> 1)]);
        else if (nsum == 2)
            a += labs((long)encDC[1] - (long)sums[delta]);
        int ads = (int)a + costMvX[i];
        if (ads < thresh)
            mvs[nmv++] = (int16_t)i;
    }
    return nmv;
}

/* ================================================================== a4/a5: SATD / SA8D */
/* 4x4 Hadamard cost: sum |H d H^T| >> 1 (pixel.cpp:210-236).  The reference packs two 16/32-bit
 * lanes into one word (SWAR); lane overflow cannot happen for valid pixel ranges, so the plain
 * integer transform is bit-identical (re-checked in tests). */
static int hadamard4x4_abs(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int d[4][4], t[4][4];
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
            d[y][x] = (int)a[y * sa + x] - (int)b[y * sb + x];
    for (int y = 0; y < 4; y++)
    {
        int s0 = d[y][0] + d[y][1], s1 = d[y][0] - d[y][1], s2 = d[y][2] + d[y][3], s3 = d[y][2] - d[y][3];
        t[y][0] = s0 + s2; t[y][1] = s1 + s3; t[y][2] = s0 - s2; t[y][3] = s1 - s3;
    }
    int acc = 0;
    for (int x = 0; x < 4; x++)
    {
        int s0 = t[0][x] + t[1][x], s1 = t[0][x] - t[1][x], s2 = t[2][x] + t[3][x], s3 = t[2][x] - t[3][x];
        acc += iabs(s0 + s2) + iabs(s1 + s3) + iabs(s0 - s2) + iabs(s1 - s3);
    }
    return acc;   /* always even */
}

/* pixel.cpp:263-297 + size map :1131-1155: tiles of 4x4 (or 8x4 pairs), one >>1 per tile (pair) */
static inline int satd_wh(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int w, int h)
{
    int acc = 0;
    for (int y = 0; y < h; y += 4)
        for (int x = 0; x < w; x += 4)
            acc += hadamard4x4_abs(a + y * sa + x, sa, b + y * sb + x, sb) >> 1;
    return acc;
}

/* un-normalised 8x8 Hadamard abs-sum (pixel.cpp:299-334) */
static int hadamard8x8_abs(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int m[8][8];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            m[y][x] = (int)a[y * sa + x] - (int)b[y * sb + x];
    for (int pass = 0; pass < 2; pass++)
    {
        for (int r = 0; r < 8; r++)
        {
            int v[8];
            for (int i = 0; i < 8; i++) v[i] = pass ? m[i][r] : m[r][i];
            for (int step = 1; step < 8; step <<= 1)
                for (int i = 0; i < 8; i += step << 1)
                    for (int j = i; j < i + step; j++)
                    {
                        int p = v[j], q = v[j + step];
                        v[j] = p + q; v[j + step] = p - q;
                    }
            for (int i = 0; i < 8; i++) { if (pass) m[i][r] = v[i]; else m[r][i] = v[i]; }
        }
    }
    int acc = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            acc += iabs(m[y][x]);
    return acc;
}

/* pixel.cpp:336-377: 8x8 -> (s+2)>>2; 16x16 -> ONE rounding over the four 8x8 sums; larger
 * blocks are sums of 16x16 units (sa8d16) or of 8x8 units (sa8d8, used for chroma 8-wide CUs). */
static inline int sa8d_8x8(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    return (hadamard8x8_abs(a, sa, b, sb) + 2) >> 2;
}
static inline int sa8d_16x16(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int s = hadamard8x8_abs(a, sa, b, sb) + hadamard8x8_abs(a + 8, sa, b + 8, sb)
          + hadamard8x8_abs(a + 8 * sa, sa, b + 8 * sb, sb) + hadamard8x8_abs(a + 8 * sa + 8, sa, b + 8 * sb + 8, sb);
    return (s + 2) >> 2;
}
static inline int sa8d_units8(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int w, int h)
{
    int acc = 0;
    for (int y = 0; y < h; y += 8)
        for (int x = 0; x < w; x += 8)
            acc += sa8d_8x8(a + y * sa + x, sa, b + y * sb + x, sb);
    return acc;
}
static inline int sa8d_units16(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int w, int h)
{
    int acc = 0;
    for (int y = 0; y < h; y += 16)
        for (int x = 0; x < w; x += 16)
            acc += sa8d_16x16(a + y * sa + x, sa, b + y * sb + x, sb);
    return acc;
}

/* ================================================================== a6: distortion metrics */
/* pixel.cpp:167-186: int product, accumulated in sse_t */
static inline sse_t sse_pp_wh(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int w, int h)
{
    sse_t acc = 0;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++)
        {
            int d = (int)a[x] - (int)b[x];
            acc += d * d;
        }
    return acc;
}
static inline sse_t sse_ss_wh(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb, int w, int h)
{
    sse_t acc = 0;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++)
        {
            int d = (int)a[x] - (int)b[x];
            acc += d * d;
        }
    return acc;
}
/* pixel.cpp:379-391 */
static inline sse_t ssd_s_n(const int16_t* a, intptr_t sa, int n)
{
    sse_t acc = 0;
    for (int y = 0; y < n; y++, a += sa)
        for (int x = 0; x < n; x++)
            acc += a[x] * a[x];
    return acc;
}
/* pixel.cpp:703-720: low word = sum, high word = sum of squares, both 32-bit accumulators */
static inline uint64_t var_n(const pixel* p, intptr_t s, int n)
{
    uint32_t sum = 0, sqr = 0;
    for (int y = 0; y < n; y++, p += s)
        for (int x = 0; x < n; x++)
        {
            sum += p[x];
            sqr += (uint32_t)p[x] * p[x];
        }
    return sum + ((uint64_t)sqr << 32);
}
/* pixel.cpp:726-757: |AC energy(source) - AC energy(recon)| per 8x8 (4x4 for the smallest CU),
 * AC energy = Hadamard cost against a zero block minus (pixel sum >> 2). */
static inline int psy_cost_n(const pixel* src, intptr_t ss, const pixel* rec, intptr_t rs, int n)
{
    static const pixel zeros[8] = { 0 };
    if (n == 4)
    {
        int es = (hadamard4x4_abs(src, ss, zeros, 0) >> 1) - (sad_wh(src, ss, zeros, 0, 4, 4) >> 2);
        int er = (hadamard4x4_abs(rec, rs, zeros, 0) >> 1) - (sad_wh(rec, rs, zeros, 0, 4, 4) >> 2);
        return iabs(es - er);
    }
    uint32_t tot = 0;
    for (int y = 0; y < n; y += 8)
        for (int x = 0; x < n; x += 8)
        {
            int es = sa8d_8x8(src + y * ss + x, ss, zeros, 0) - (sad_wh(src + y * ss + x, ss, zeros, 0, 8, 8) >> 2);
            int er = sa8d_8x8(rec + y * rs + x, rs, zeros, 0) - (sad_wh(rec + y * rs + x, rs, zeros, 0, 8, 8) >> 2);
            tot += (uint32_t)iabs(es - er);
        }
    return (int)tot;
}
/* pixel.cpp:958-994 */
static inline void ssim_dist_n(const pixel* fenc, uint32_t fs, const pixel* rec, intptr_t rs, uint64_t* ssBlock,
                               int shift, uint64_t* ac_k, int n)
{
    uint64_t ss = 0, ac = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
        {
            int d = (int)fenc[y * fs + x] - (int)rec[y * rs + x];
            ss += d * d;
            uint32_t t = (uint32_t)fenc[y * fs + x] >> shift;
            ac += t * t;
        }
    *ssBlock = ss;
    *ac_k = ac;
}
static void norm_fact(const pixel* src, uint32_t blockSize, int shift, uint64_t* z_k)
{
    uint64_t z = 0;
    for (uint32_t y = 0; y < blockSize; y++)
        for (uint32_t x = 0; x < blockSize; x++)
        {
            uint32_t t = (uint32_t)src[y * blockSize + x] >> shift;
            z += t * t;
        }
    *z_k = z;
}

/* ================================================================== a10: block glue ops */
/* pixel.cpp:759-862, 393-491, 545-557 */
static inline void copy_pp_wh(pixel* d, intptr_t ds, const pixel* s, intptr_t ss, int w, int h)
{ for (int y = 0; y < h; y++, d += ds, s += ss) for (int x = 0; x < w; x++) d[x] = s[x]; }
static inline void copy_ss_wh(int16_t* d, intptr_t ds, const int16_t* s, intptr_t ss, int w, int h)
{ for (int y = 0; y < h; y++, d += ds, s += ss) for (int x = 0; x < w; x++) d[x] = s[x]; }
static inline void copy_sp_wh(pixel* d, intptr_t ds, const int16_t* s, intptr_t ss, int w, int h)
{ for (int y = 0; y < h; y++, d += ds, s += ss) for (int x = 0; x < w; x++) d[x] = (pixel)s[x]; }
static inline void copy_ps_wh(int16_t* d, intptr_t ds, const pixel* s, intptr_t ss, int w, int h)
{ for (int y = 0; y < h; y++, d += ds, s += ss) for (int x = 0; x < w; x++) d[x] = (int16_t)s[x]; }
static inline void sub_ps_wh(int16_t* d, intptr_t ds, const pixel* a, const pixel* b, intptr_t sa, intptr_t sb, int w, int h)
{ for (int y = 0; y < h; y++, d += ds, a += sa, b += sb) for (int x = 0; x < w; x++) d[x] = (int16_t)((int)a[x] - (int)b[x]); }
static inline void add_ps_wh(pixel* d, intptr_t ds, const pixel* a, const int16_t* r, intptr_t sa, intptr_t sr, int w, int h)
{ for (int y = 0; y < h; y++, d += ds, a += sa, r += sr) for (int x = 0; x < w; x++) d[x] = clip_pixel((int)a[x] + (int)r[x]); }
/* pixel.cpp:471-483: one shared stride for fenc, pred and residual */
static inline void calcresidual_n(const pixel* fenc, const pixel* pred, int16_t* resi, intptr_t stride, int n)
{ sub_ps_wh(resi, stride, fenc, pred, stride, stride, n, n); }
/* pixel.cpp:842-862: bi-pred average of two 14-bit intermediates */
static inline void addavg_wh(const int16_t* a, const int16_t* b, pixel* d, intptr_t sa, intptr_t sb, intptr_t ds, int w, int h)
{
    const int shift = IF_PREC + 1 - DEPTH;
    const int offset = (1 << (shift - 1)) + 2 * IF_OFFS;
    for (int y = 0; y < h; y++, a += sa, b += sb, d += ds)
        for (int x = 0; x < w; x++)
            d[x] = clip_pixel(((int)a[x] + (int)b[x] + offset) >> shift);
}
static inline void pixelavg_wh(pixel* d, intptr_t ds, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int w, int h)
{ for (int y = 0; y < h; y++, d += ds, a += sa, b += sb) for (int x = 0; x < w; x++) d[x] = (pixel)(((int)a[x] + (int)b[x] + 1) >> 1); }
static inline void blockfill_n(int16_t* d, intptr_t ds, int16_t v, int n)
{ for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) d[y * ds + x] = v; }
static inline void transpose_n(pixel* d, const pixel* s, intptr_t ss, int n)
{ for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) d[y * n + x] = s[x * ss + y]; }
/* pixel.cpp:401-469: shifts happen in int, the store truncates to int16; the shr rounding
 * constant is itself an int16_t in the reference */
static inline void cpy2dto1d_shl_n(int16_t* d, const int16_t* s, intptr_t ss, int shift, int n)
{ for (int y = 0; y < n; y++, s += ss, d += n) for (int x = 0; x < n; x++) d[x] = (int16_t)((int)s[x] << shift); }
static inline void cpy2dto1d_shr_n(int16_t* d, const int16_t* s, intptr_t ss, int shift, int n)
{ int16_t r = (int16_t)(1 << (shift - 1)); for (int y = 0; y < n; y++, s += ss, d += n) for (int x = 0; x < n; x++) d[x] = (int16_t)(((int)s[x] + r) >> shift); }
static inline void cpy1dto2d_shl_n(int16_t* d, const int16_t* s, intptr_t ds, int shift, int n)
{ for (int y = 0; y < n; y++, s += n, d += ds) for (int x = 0; x < n; x++) d[x] = (int16_t)((int)s[x] << shift); }
static inline void cpy1dto2d_shr_n(int16_t* d, const int16_t* s, intptr_t ds, int shift, int n)
{ int16_t r = (int16_t)(1 << (shift - 1)); for (int y = 0; y < n; y++, s += n, d += ds) for (int x = 0; x < n; x++) d[x] = (int16_t)(((int)s[x] + r) >> shift); }
/* dct.cpp:728-742 */
static inline uint32_t copy_cnt_n(int16_t* coeff, const int16_t* resi, intptr_t rs, int n)
{
    uint32_t nz = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
        {
            int16_t v = resi[y * rs + x];
            coeff[y * n + x] = v;
            nz += (v != 0);
        }
    return nz;
}
/* dct.cpp:714-726 */
static inline int count_nonzero_n(const int16_t* q, int n)
{ int c = 0; for (int i = 0; i < n * n; i++) c += (q[i] != 0); return c; }

/* pixel.cpp:493-543: explicit weighted prediction */
static void weight_sp(const int16_t* src, pixel* dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift, int offset)
{
    for (int y = 0; y < height; y++, src += ss, dst += ds)
        for (int x = 0; x < width; x++)
            dst[x] = clip_pixel(((w0 * ((int)src[x] + IF_OFFS) + round) >> shift) + offset);
}
static void weight_pp(const pixel* src, pixel* dst, intptr_t stride, int width, int height, int w0, int round, int shift, int offset)
{
    const int corr = IF_PREC - DEPTH;
    for (int y = 0; y < height; y++, src += stride, dst += stride)
        for (int x = 0; x < width; x++)
        {
            int16_t v = (int16_t)((int)src[x] << corr);
            dst[x] = clip_pixel(((w0 * (int)v + round) >> shift) + offset);
        }
}
/* pixel.cpp:559-602: intra 64x64 neighbour / block decimation */
static void scale1d_128to64(pixel* dst, const pixel* src)
{
    for (int x = 0; x < 64; x++)
    {
        dst[x] = (pixel)(((int)src[2 * x] + src[2 * x + 1] + 1) >> 1);
        dst[64 + x] = (pixel)(((int)src[128 + 2 * x] + src[128 + 2 * x + 1] + 1) >> 1);
    }
}
static void scale2d_64to32(pixel* dst, const pixel* src, intptr_t stride)
{
    for (int y = 0; y < 32; y++)
        for (int x = 0; x < 32; x++)
        {
            const pixel* p = src + 2 * y * stride + 2 * x;
            dst[y * 32 + x] = (pixel)(((int)p[0] + p[1] + p[stride] + p[stride + 1] + 2) >> 2);
        }
}

/* ================================================================== a7: transforms */
/* Forward: dst[k][j] = (int16)((sum_i M[k][i] * src[j][i] + add) >> shift), i.e. row transform with
 * transposed store, applied twice (dct.cpp:83-240, 418-440, 459-525).  The reference's partial
 * butterflies are an exact refactoring of this product in wrap-around int32 arithmetic, and the
 * (int16_t) store truncates WITHOUT clipping. */
static void fwd_stage(const int16_t* src, int16_t* dst, int N, int shift)
{
    const int add = 1 << (shift - 1);
    for (int j = 0; j < N; j++)
        for (int k = 0; k < N; k++)
        {
            int32_t acc = 0;
            for (int i = 0; i < N; i++)
                acc += tcoef(N, k, i) * (int32_t)src[j * N + i];
            dst[k * N + j] = (int16_t)((acc + add) >> shift);
        }
}
static void dct_n(const int16_t* src, int16_t* dst, intptr_t srcStride, int N, int log2N)
{
    int16_t blk[32 * 32], tmp[32 * 32];
    init_tables();
    for (int y = 0; y < N; y++)
        memcpy(blk + y * N, src + y * srcStride, N * sizeof(int16_t));
    fwd_stage(blk, tmp, N, log2N - 1 + DEPTH - 8);   /* shift_1st, dct.cpp:461,478,495,512 */
    fwd_stage(tmp, dst, N, log2N + 6);               /* shift_2nd */
}
/* Inverse: dst[j][k] = clip16((sum_i M[i][k] * src[i][j] + add) >> shift) (dct.cpp:242-416,
 * 544-610): first shift 7, second 12 - (depth - 8), every stage clipped to int16. */
static void inv_stage(const int16_t* src, int16_t* dst, int N, int shift)
{
    const int add = 1 << (shift - 1);
    for (int j = 0; j < N; j++)
        for (int k = 0; k < N; k++)
        {
            int32_t acc = 0;
            for (int i = 0; i < N; i++)
                acc += tcoef(N, i, k) * (int32_t)src[i * N + j];
            dst[j * N + k] = (int16_t)clip3i(-32768, 32767, (acc + add) >> shift);
        }
}
static void idct_n(const int16_t* src, int16_t* dst, intptr_t dstStride, int N)
{
    int16_t tmp[32 * 32], blk[32 * 32];
    init_tables();
    inv_stage(src, tmp, N, 7);
    inv_stage(tmp, blk, N, 12 - (DEPTH - 8));
    for (int y = 0; y < N; y++)
        memcpy(dst + y * dstStride, blk + y * N, N * sizeof(int16_t));
}
/* 4x4 DST-VII for intra luma (dct.cpp:43-81, 442-457, 527-542).  Matrix form of the fast
 * algorithm: rows {29,55,74,84}, {74,74,0,-74}, {84,-29,-74,55}, {55,-84,74,-29}. */
static const int kDst[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
static void dst4_stage(const int16_t* in, int16_t* out, int shift)
{
    const int add = 1 << (shift - 1);
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 4; k++)
        {
            int acc = 0;
            for (int i = 0; i < 4; i++) acc += kDst[k][i] * (int)in[4 * j + i];
            out[4 * k + j] = (int16_t)((acc + add) >> shift);
        }
}
static void idst4_stage(const int16_t* in, int16_t* out, int shift)
{
    const int add = 1 << (shift - 1);
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 4; k++)
        {
            int acc = 0;
            for (int i = 0; i < 4; i++) acc += kDst[i][k] * (int)in[4 * i + j];
            out[4 * j + k] = (int16_t)clip3i(-32768, 32767, (acc + add) >> shift);
        }
}
static void dst4(const int16_t* src, int16_t* dst, intptr_t srcStride)
{
    int16_t blk[16], tmp[16];
    for (int y = 0; y < 4; y++) memcpy(blk + 4 * y, src + y * srcStride, 4 * sizeof(int16_t));
    dst4_stage(blk, tmp, 1 + DEPTH - 8);
    dst4_stage(tmp, dst, 8);
}
static void idst4(const int16_t* src, int16_t* dst, intptr_t dstStride)
{
    int16_t tmp[16], blk[16];
    idst4_stage(src, tmp, 7);
    idst4_stage(tmp, blk, 12 - (DEPTH - 8));
    for (int y = 0; y < 4; y++) memcpy(dst + y * dstStride, blk + 4 * y, 4 * sizeof(int16_t));
}

/* lowpassdct.cpp:34-113 (--lowpass-dct): 2x2-average the residual, transform at half size into the
 * top-left quadrant, zero the rest, and overwrite DC with a scaled block sum.  The int16_t
 * truncations of the running sums are part of the behaviour. */
static void lowpass_dct_n(const int16_t* src, int16_t* dst, intptr_t ss, int N, int log2N)
{
    const int H = N / 2;
    int16_t avg[16 * 16], coef[16 * 16];
    int32_t total32 = 0;
    int16_t total16 = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < H; j++)
        {
            int16_t s4 = (int16_t)((int)src[2 * i * ss + 2 * j] + src[2 * i * ss + 2 * j + 1]
                                 + src[(2 * i + 1) * ss + 2 * j] + src[(2 * i + 1) * ss + 2 * j + 1]);
            avg[i * H + j] = (int16_t)(s4 >> 2);
            total32 += s4;
            total16 = (int16_t)(total16 + s4);
        }
    dct_n(avg, coef, H, H, log2N - 1);
    memset(dst, 0, (size_t)N * N * sizeof(int16_t));
    for (int i = 0; i < H; i++)
        memcpy(dst + i * N, coef + i * H, H * sizeof(int16_t));
    if (N == 8) dst[0] = (int16_t)((int)total16 << 1);
    else if (N == 16) dst[0] = (int16_t)(total32 >> 1);
    else dst[0] = (int16_t)(total32 >> 3);
}
static void lowpass_8(const int16_t* s, int16_t* d, intptr_t ss) { lowpass_dct_n(s, d, ss, 8, 3); }
static void lowpass_16(const int16_t* s, int16_t* d, intptr_t ss) { lowpass_dct_n(s, d, ss, 16, 4); }
static void lowpass_32(const int16_t* s, int16_t* d, intptr_t ss) { lowpass_dct_n(s, d, ss, 32, 5); }

/* ================================================================== a8: quantisation */
/* dct.cpp:664-686 */
static uint32_t quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    uint32_t numSig = 0;
    for (int i = 0; i < numCoeff; i++)
    {
        int c = coef[i];
        int t = iabs(c) * quantCoeff[i];
        int level = (t + add) >> qBits;
        deltaU[i] = (t - (level << qBits)) >> (qBits - 8);
        numSig += (level != 0);
        if (c < 0) level = -level;
        qCoef[i] = (int16_t)clip3i(-32768, 32767, level);
    }
    return numSig;
}
/* dct.cpp:688-713: magnitude-only variant used by RDOQ */
static uint32_t nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    uint32_t numSig = 0;
    for (int i = 0; i < numCoeff; i++)
    {
        int c = coef[i];
        int level = (iabs(c) * quantCoeff[i] + add) >> qBits;
        numSig += (level != 0);
        if (c < 0) level = -level;
        qCoef[i] = (int16_t)iabs(clip3i(-32768, 32767, level));
    }
    return numSig;
}
/* dct.cpp:612-634 */
static void dequant_normal(const int16_t* q, int16_t* coef, int num, int scale, int shift)
{
    const int add = 1 << (shift - 1);
    for (int i = 0; i < num; i++)
        coef[i] = (int16_t)clip3i(-32768, 32767, ((int)q[i] * scale + add) >> shift);
}
/* dct.cpp:636-662 */
static void dequant_scaling(const int16_t* q, const int32_t* dq, int16_t* coef, int num, int per, int shift)
{
    shift += 4;
    if (shift > per)
    {
        const int add = 1 << (shift - per - 1);
        for (int i = 0; i < num; i++)
            coef[i] = (int16_t)clip3i(-32768, 32767, ((int)q[i] * dq[i] + add) >> (shift - per));
    }
    else
    {
        for (int i = 0; i < num; i++)
        {
            int v = clip3i(-32768, 32767, (int)q[i] * dq[i]);
            coef[i] = (int16_t)clip3i(-32768, 32767, v << (per - shift));
        }
    }
}
/* dct.cpp:744-755 */
static void denoise_dct(int16_t* dctCoef, uint32_t* resSum, const uint16_t* offset, int numCoeff)
{
    for (int i = 0; i < numCoeff; i++)
    {
        int level = dctCoef[i];
        int neg = level < 0;
        int mag = neg ? -level : level;
        resSum[i] += (uint32_t)mag;
        mag -= offset[i];
        dctCoef[i] = (int16_t)(mag < 0 ? 0 : (neg ? -mag : mag));
    }
}

/* ================================================================== a11: interpolation */
/* ipfilter.cpp:79-317.  taps = 8 (luma, coeffIdx 0..3) or 4 (chroma, 0..7).  Every variant forms
 * `(int16_t)((sum + offset) >> shift)` BEFORE any clipping. */
static inline const int16_t* taps_for(int N, int idx) { return N == 8 ? kLumaTaps[idx] : kChromaTaps[idx]; }

static inline int fir(const pixel* p, intptr_t step, const int16_t* c, int N)
{ int s = 0; for (int t = 0; t < N; t++) s += (int)p[t * step] * c[t]; return s; }
static inline int fir_s(const int16_t* p, intptr_t step, const int16_t* c, int N)
{ int s = 0; for (int t = 0; t < N; t++) s += (int)p[t * step] * c[t]; return s; }
static inline pixel clip_val16(int16_t v) { return (pixel)(v < 0 ? 0 : (v > PIXEL_MAX ? PIXEL_MAX : v)); }

static inline void interp_hpp(const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idx, int N, int w, int h)
{
    const int16_t* c = taps_for(N, idx);
    src -= N / 2 - 1;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
            dst[x] = clip_val16((int16_t)((fir(src + x, 1, c, N) + (1 << (IF_FPREC - 1))) >> IF_FPREC));
}
static inline void interp_hps(const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx, int isRowExt, int N, int w, int h)
{
    const int16_t* c = taps_for(N, idx);
    const int shift = IF_FPREC - (IF_PREC - DEPTH);
    const int offset = -(IF_OFFS << shift);
    src -= N / 2 - 1;
    if (isRowExt) { src -= (N / 2 - 1) * ss; h += N - 1; }
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
            dst[x] = (int16_t)((fir(src + x, 1, c, N) + offset) >> shift);
}
static inline void interp_vpp(const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idx, int N, int w, int h)
{
    const int16_t* c = taps_for(N, idx);
    src -= (N / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
            dst[x] = clip_val16((int16_t)((fir(src + x, ss, c, N) + (1 << (IF_FPREC - 1))) >> IF_FPREC));
}
static inline void interp_vps(const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx, int N, int w, int h)
{
    const int16_t* c = taps_for(N, idx);
    const int shift = IF_FPREC - (IF_PREC - DEPTH);
    const int offset = -(IF_OFFS << shift);
    src -= (N / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
            dst[x] = (int16_t)((fir(src + x, ss, c, N) + offset) >> shift);
}
static inline void interp_vsp(const int16_t* src, intptr_t ss, pixel* dst, intptr_t ds, int idx, int N, int w, int h)
{
    const int16_t* c = taps_for(N, idx);
    const int shift = IF_FPREC + (IF_PREC - DEPTH);
    const int offset = (1 << (shift - 1)) + (IF_OFFS << IF_FPREC);
    src -= (N / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
            dst[x] = clip_val16((int16_t)((fir_s(src + x, ss, c, N) + offset) >> shift));
}
static inline void interp_vss(const int16_t* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx, int N, int w, int h)
{
    const int16_t* c = taps_for(N, idx);
    src -= (N / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
            dst[x] = (int16_t)(fir_s(src + x, ss, c, N) >> IF_FPREC);
}
/* ipfilter.cpp:362-369: horizontal pass with N-1 extra rows into a w-stride scratch, then vertical */
static inline void interp_hvpp(const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idxX, int idxY, int N, int w, int h)
{
    int16_t immed[64 * (64 + 7)];
    interp_hps(src, ss, immed, w, idxX, 1, N, w, h);
    interp_vsp(immed + (N / 2 - 1) * w, w, dst, ds, idxY, N, w, h);
}
/* ipfilter.cpp:40-57 */
static inline void p2s_wh(const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int w, int h)
{
    const int shift = IF_PREC - DEPTH;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int16_t v = (int16_t)((int)src[x] << shift);
            dst[x] = (int16_t)(v - (int16_t)IF_OFFS);
        }
}
/* ipfilter.cpp:59-77 */
static void extend_row_border(pixel* txt, intptr_t stride, int width, int height, int marginX)
{
    for (int y = 0; y < height; y++, txt += stride)
        for (int x = 0; x < marginX; x++)
        {
            txt[-marginX + x] = txt[0];
            txt[width + x] = txt[width - 1];
        }
}

/* ================================================================== a12: intra prediction */
/* Neighbour buffer layout (intrapred.cpp:36-50,92-93): [0] top-left, [1..2N] above + above-right,
 * [2N+1..4N] left + below-left. */
/* intrapred.cpp:31-51: [1 2 1] smoothing, corner joins the two arms, far ends are kept */
static inline void intra_filter_n(const pixel* s, pixel* f, int n)
{
    const int n2 = 2 * n;
    f[0] = (pixel)((2 * (int)s[0] + s[1] + s[n2 + 1] + 2) >> 2);
    for (int i = 1; i < n2; i++)
        f[i] = (pixel)((2 * (int)s[i] + s[i - 1] + s[i + 1] + 2) >> 2);
    f[n2] = s[n2];
    f[n2 + 1] = (pixel)((2 * (int)s[n2 + 1] + s[0] + s[n2 + 2] + 2) >> 2);
    for (int i = n2 + 2; i < 2 * n2; i++)
        f[i] = (pixel)((2 * (int)s[i] + s[i - 1] + s[i + 1] + 2) >> 2);
    f[2 * n2] = s[2 * n2];
}
/* intrapred.cpp:53-85 */
static inline void intra_dc_n(pixel* dst, intptr_t ds, const pixel* nb, int dirMode, int bFilter, int n)
{
    (void)dirMode;
    const pixel* above = nb + 1;
    const pixel* left = nb + 2 * n + 1;
    int sum = n;
    for (int i = 0; i < n; i++) sum += above[i] + left[i];
    const int dc = sum / (2 * n);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * ds + x] = (pixel)dc;
    if (bFilter)
    {
        dst[0] = (pixel)((above[0] + left[0] + 2 * dc + 2) >> 2);
        for (int x = 1; x < n; x++) dst[x] = (pixel)((above[x] + 3 * dc + 2) >> 2);
        for (int y = 1; y < n; y++) dst[y * ds] = (pixel)((left[y] + 3 * dc + 2) >> 2);
    }
}
/* intrapred.cpp:87-100 */
static inline void intra_planar_n(pixel* dst, intptr_t ds, const pixel* nb, int dirMode, int bFilter, int n, int log2n)
{
    (void)dirMode; (void)bFilter;
    const pixel* above = nb + 1;
    const pixel* left = nb + 2 * n + 1;
    const int tr = above[n], bl = left[n];
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * ds + x] = (pixel)(((n - 1 - x) * left[y] + (n - 1 - y) * above[x] + (x + 1) * tr + (y + 1) * bl + n) >> (log2n + 1));
}
/* intrapred.cpp:102-204.  Modes 2..17 are predicted as their vertical mirror (roles of the two
 * neighbour arms swapped) and transposed at the end unless `keepTransposed` (all-angs packing,
 * intrapred.cpp:206-234). */
static const int8_t kAngle[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int16_t kInvAngle[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };

static void intra_ang_core(pixel* dst, intptr_t ds, const pixel* nb0, int mode, int bFilter, int n, int keepTransposed)
{
    const int n2 = 2 * n;
    const int hor = mode < 18;
    pixel swapped[129];
    const pixel* nb = nb0;
    if (hor)
    {
        swapped[0] = nb0[0];
        for (int i = 0; i < n2; i++)
        {
            swapped[1 + i] = nb0[n2 + 1 + i];
            swapped[n2 + 1 + i] = nb0[1 + i];
        }
        nb = swapped;
    }
    const int aoff = hor ? 10 - mode : mode - 26;
    const int angle = kAngle[8 + aoff];

    if (angle == 0)
    {
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                dst[y * ds + x] = nb[1 + x];
        if (bFilter)
        {
            const int tl = nb[0], top = nb[1];
            for (int y = 0; y < n; y++)
                dst[y * ds] = clip_pixel((int16_t)(top + (((int)nb[n2 + 1 + y] - tl) >> 1)));
        }
    }
    else
    {
        pixel line[64 + 1 + 64];
        const pixel* ref;
        if (angle < 0)
        {
            /* extend the main arm to the left with projected side-arm samples */
            const int nproj = -((n * angle) >> 5) - 1;
            pixel* base = line + nproj + 1;          /* base[-1] = top-left, base[0..] = main arm */
            int acc = 128;
            for (int i = 0; i < nproj; i++)
            {
                acc += kInvAngle[-aoff - 1];
                base[-2 - i] = nb[n2 + (acc >> 8)];
            }
            for (int i = 0; i <= n; i++)
                base[-1 + i] = nb[i];
            ref = base;
        }
        else
            ref = nb + 1;

        int pos = 0;
        for (int y = 0; y < n; y++)
        {
            pos += angle;
            const int off = pos >> 5, frac = pos & 31;
            if (frac)
                for (int x = 0; x < n; x++)
                    dst[y * ds + x] = (pixel)(((32 - frac) * ref[off + x] + frac * ref[off + x + 1] + 16) >> 5);
            else
                for (int x = 0; x < n; x++)
                    dst[y * ds + x] = ref[off + x];
        }
    }

    if (hor && !keepTransposed)
        for (int y = 0; y < n - 1; y++)
            for (int x = y + 1; x < n; x++)
            {
                pixel t = dst[y * ds + x];
                dst[y * ds + x] = dst[x * ds + y];
                dst[x * ds + y] = t;
            }
}
static inline void intra_ang_n(pixel* dst, intptr_t ds, const pixel* nb, int mode, int bFilter, int n)
{ intra_ang_core(dst, ds, nb, mode, bFilter, n, 0); }
/* intrapred.cpp:206-234: modes 2..34 packed at dest + (mode-2)*N*N, horizontal modes stored
 * un-flipped (i.e. transposed); filtered neighbours chosen per mode by kIntraFilterFlags & N */
static inline void intra_allangs_n(pixel* dest, pixel* refPix, pixel* filtPix, int bLuma, int n)
{
    for (int mode = 2; mode <= 34; mode++)
    {
        const pixel* nb = (kIntraFilterFlags[mode] & n) ? filtPix : refPix;
        intra_ang_core(dest + (mode - 2) * n * n, n, nb, mode, bLuma, n, 1);
    }
}

/* ================================================================== a13: SAO */
/* loopfilter.cpp:39-138 - apply.  Sign carry buffers are part of the contract. */
static void sao_sign(int8_t* dst, const pixel* a, const pixel* b, const int endX)
{ for (int x = 0; x < endX; x++) dst[x] = (int8_t)sgn((int)a[x] - (int)b[x]); }

static void sao_e0(pixel* rec, int8_t* offsetEo, int width, int8_t* signLeft, intptr_t stride)
{
    for (int y = 0; y < 2; y++, rec += stride)
    {
        int sl = signLeft[y];
        for (int x = 0; x < width; x++)
        {
            int sr = sgn((int)rec[x] - (int)rec[x + 1]);
            int cls = sr + sl + 2;
            sl = -sr;
            rec[x] = clip_pixel(rec[x] + offsetEo[cls]);
        }
    }
}
static inline void sao_e1_rows(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int width, int rows)
{
    for (int y = 0; y < rows; y++, rec += stride)
        for (int x = 0; x < width; x++)
        {
            int sd = sgn((int)rec[x] - (int)rec[x + stride]);
            int cls = sd + upBuff1[x] + 2;
            upBuff1[x] = (int8_t)(-sd);
            rec[x] = clip_pixel(rec[x] + offsetEo[cls]);
        }
}
static void sao_e1(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int width) { sao_e1_rows(rec, upBuff1, offsetEo, stride, width, 1); }
static void sao_e1_2rows(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int width) { sao_e1_rows(rec, upBuff1, offsetEo, stride, width, 2); }
static void sao_e2(pixel* rec, int8_t* bufft, int8_t* buff1, int8_t* offsetEo, int width, intptr_t stride)
{
    for (int x = 0; x < width; x++)
    {
        int sd = sgn((int)rec[x] - (int)rec[x + stride + 1]);
        int cls = sd + buff1[x] + 2;
        bufft[x + 1] = (int8_t)(-sd);
        rec[x] = clip_pixel(rec[x] + offsetEo[cls]);
    }
}
static void sao_e3(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int startX, int endX)
{
    for (int x = startX + 1; x < endX; x++)
    {
        int sd = sgn((int)rec[x] - (int)rec[x + stride]);
        int cls = (int8_t)(sd + upBuff1[x] + 2);
        upBuff1[x - 1] = (int8_t)(-sd);
        rec[x] = clip_pixel(rec[x] + offsetEo[cls]);
    }
}
static void sao_b0(pixel* rec, const int8_t* offset, int ctuWidth, int ctuHeight, intptr_t stride)
{
    const int boShift = DEPTH - 5;
    for (int y = 0; y < ctuHeight; y++, rec += stride)
        for (int x = 0; x < ctuWidth; x++)
            rec[x] = clip_pixel(rec[x] + offset[rec[x] >> boShift]);
}
/* sao.cpp:1762-1925 - statistics; diff has fixed stride MAX_CU (64); edge classes are folded
 * through s_eoTable = {1,2,0,3,4} (sao.cpp:65-72) and ACCUMULATED into stats/count. */
static const int kEoTable[5] = { 1, 2, 0, 3, 4 };
static void sao_stats_bo(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count)
{
    const int boShift = DEPTH - 5;
    for (int y = 0; y < endY; y++, diff += MAX_CU, rec += stride)
        for (int x = 0; x < endX; x++)
        {
            int cls = rec[x] >> boShift;
            stats[cls] += diff[x];
            count[cls]++;
        }
}
static inline void eo_fold(const int32_t* ts, const int32_t* tc, int32_t* stats, int32_t* count)
{ for (int i = 0; i < 5; i++) { stats[kEoTable[i]] += ts[i]; count[kEoTable[i]] += tc[i]; } }
static void sao_stats_e0(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count)
{
    int32_t ts[5] = { 0 }, tc[5] = { 0 };
    for (int y = 0; y < endY; y++, diff += MAX_CU, rec += stride)
    {
        int sl = sgn((int)rec[0] - (int)rec[-1]);
        for (int x = 0; x < endX; x++)
        {
            int sr = sgn((int)rec[x] - (int)rec[x + 1]);
            int cls = sr + sl + 2;
            sl = -sr;
            ts[cls] += diff[x]; tc[cls]++;
        }
    }
    eo_fold(ts, tc, stats, count);
}
static void sao_stats_e1(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* upBuff1, int endX, int endY, int32_t* stats, int32_t* count)
{
    int32_t ts[5] = { 0 }, tc[5] = { 0 };
    for (int y = 0; y < endY; y++, diff += MAX_CU, rec += stride)
        for (int x = 0; x < endX; x++)
        {
            int sd = sgn((int)rec[x] - (int)rec[x + stride]);
            int cls = sd + upBuff1[x] + 2;
            upBuff1[x] = (int8_t)(-sd);
            ts[cls] += diff[x]; tc[cls]++;
        }
    eo_fold(ts, tc, stats, count);
}
static void sao_stats_e2(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* upBuff1, int8_t* upBufft, int endX, int endY, int32_t* stats, int32_t* count)
{
    int32_t ts[5] = { 0 }, tc[5] = { 0 };
    for (int y = 0; y < endY; y++, diff += MAX_CU, rec += stride)
    {
        upBufft[0] = (int8_t)sgn((int)rec[stride] - (int)rec[-1]);
        for (int x = 0; x < endX; x++)
        {
            int sd = sgn((int)rec[x] - (int)rec[x + stride + 1]);
            int cls = sd + upBuff1[x] + 2;
            upBufft[x + 1] = (int8_t)(-sd);
            ts[cls] += diff[x]; tc[cls]++;
        }
        int8_t* t = upBuff1; upBuff1 = upBufft; upBufft = t;    /* the two row buffers trade places */
    }
    eo_fold(ts, tc, stats, count);
}
static void sao_stats_e3(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* upBuff1, int endX, int endY, int32_t* stats, int32_t* count)
{
    int32_t ts[5] = { 0 }, tc[5] = { 0 };
    for (int y = 0; y < endY; y++, diff += MAX_CU, rec += stride)
    {
        for (int x = 0; x < endX; x++)
        {
            int sd = sgn((int)rec[x] - (int)rec[x + stride - 1]);
            int cls = sd + upBuff1[x] + 2;
            upBuff1[x - 1] = (int8_t)(-sd);
            ts[cls] += diff[x]; tc[cls]++;
        }
        upBuff1[endX - 1] = (int8_t)sgn((int)rec[endX - 1 + stride] - (int)rec[endX]);
    }
    eo_fold(ts, tc, stats, count);
}

/* ================================================================== a14: deblocking edge filters */
/* loopfilter.cpp:140-180: 4 lines per call; `offset` walks across the edge, `srcStep` along it */
static void deblock_luma_strong(pixel* src, intptr_t srcStep, intptr_t offset, int32_t tcP, int32_t tcQ)
{
    for (int i = 0; i < 4; i++, src += srcStep)
    {
        const int p3 = src[-offset * 4], p2 = src[-offset * 3], p1 = src[-offset * 2], p0 = src[-offset];
        const int q0 = src[0], q1 = src[offset], q2 = src[offset * 2], q3 = src[offset * 3];
        src[-offset * 3] = (pixel)(clip3i(-tcP, tcP, ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2) + p2);
        src[-offset * 2] = (pixel)(clip3i(-tcP, tcP, ((p2 + p1 + p0 + q0 + 2) >> 2) - p1) + p1);
        src[-offset]     = (pixel)(clip3i(-tcP, tcP, ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0) + p0);
        src[0]           = (pixel)(clip3i(-tcQ, tcQ, ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0) + q0);
        src[offset]      = (pixel)(clip3i(-tcQ, tcQ, ((p0 + q0 + q1 + q2 + 2) >> 2) - q1) + q1);
        src[offset * 2]  = (pixel)(clip3i(-tcQ, tcQ, ((p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3) - q2) + q2);
    }
}
static void deblock_chroma(pixel* src, intptr_t srcStep, intptr_t offset, int32_t tc, int32_t maskP, int32_t maskQ)
{
    for (int i = 0; i < 4; i++, src += srcStep)
    {
        const int p1 = src[-offset * 2], p0 = src[-offset], q0 = src[0], q1 = src[offset];
        const int delta = clip3i(-tc, tc, (((q0 - p0) * 4) + p1 - q1 + 4) >> 3);
        src[-offset] = clip_pixel(p0 + (delta & maskP));
        src[0] = clip_pixel(q0 - (delta & maskQ));
    }
}

/* ================================================================== a15: SEA integral images */
/* framefilter.cpp:39-140: horizontal = running N-wide row sum added to the row above;
 * vertical = difference of rows N apart, in place. */
static inline void integral_h_n(uint32_t* sum, pixel* pix, intptr_t stride, int n)
{
    int32_t v = 0;
    for (int i = 0; i < n; i++) v += pix[i];
    for (int x = 0; x < stride - n; x++)
    {
        sum[x] = (uint32_t)v + sum[x - stride];
        v += (int)pix[x + n] - (int)pix[x];
    }
}
static inline void integral_v_n(uint32_t* sum, intptr_t stride, int n)
{ for (int x = 0; x < stride; x++) sum[x] = sum[x + n * stride] - sum[x]; }

/* ================================================================== per-size thunks */
#define PU_LIST(X) X(4,4) X(8,8) X(16,16) X(32,32) X(64,64) X(8,4) X(4,8) X(16,8) X(8,16) X(32,16) X(16,32) \
    X(64,32) X(32,64) X(16,12) X(12,16) X(16,4) X(4,16) X(32,24) X(24,32) X(32,8) X(8,32) X(64,48) X(48,64) X(64,16) X(16,64)
#define CU_LIST(X) X(4, 2) X(8, 3) X(16, 4) X(32, 5) X(64, 6)
#define TU_LIST(X) X(4, 2) X(8, 3) X(16, 4) X(32, 5)

/* ADS variant per PU (pixel.cpp:1105-1129) */
#define ADS_N_4x4 1
#define ADS_N_8x8 1
#define ADS_N_8x4 2
#define ADS_N_4x8 2
#define ADS_N_16x16 4
#define ADS_N_16x8 2
#define ADS_N_8x16 2
#define ADS_N_16x12 1
#define ADS_N_12x16 1
#define ADS_N_16x4 1
#define ADS_N_4x16 1
#define ADS_N_32x32 4
#define ADS_N_32x16 2
#define ADS_N_16x32 2
#define ADS_N_32x24 4
#define ADS_N_24x32 4
#define ADS_N_32x8 4
#define ADS_N_8x32 4
#define ADS_N_64x64 4
#define ADS_N_64x32 2
#define ADS_N_32x64 2
#define ADS_N_64x48 4
#define ADS_N_48x64 4
#define ADS_N_64x16 4
#define ADS_N_16x64 4

/* luma / generic (W,H) thunks; CW/CH variants are instantiated below for chroma dims too */
#define DEF_BLOCK(W, H) \
static int sad_##W##x##H(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sad_wh(a, sa, b, sb, W, H); } \
static void sadx3_##W##x##H(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res) \
{ const pixel* r[3] = { r0, r1, r2 }; sad_xn_wh(f, r, 3, rs, res, W, H); } \
static void sadx4_##W##x##H(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res) \
{ const pixel* r[4] = { r0, r1, r2, r3 }; sad_xn_wh(f, r, 4, rs, res, W, H); } \
static int satd_##W##x##H(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return satd_wh(a, sa, b, sb, W, H); } \
static void pixelavg_##W##x##H(pixel* d, intptr_t ds, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int wgt) { (void)wgt; pixelavg_wh(d, ds, a, sa, b, sb, W, H); } \
static void addavg_##W##x##H(const int16_t* a, const int16_t* b, pixel* d, intptr_t sa, intptr_t sb, intptr_t ds) { addavg_wh(a, b, d, sa, sb, ds, W, H); } \
static void copypp_##W##x##H(pixel* d, intptr_t ds, const pixel* s, intptr_t ss) { copy_pp_wh(d, ds, s, ss, W, H); } \
static void p2s_##W##x##H(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds) { p2s_wh(s, ss, d, ds, W, H); } \
static void copysp_##W##x##H(pixel* d, intptr_t ds, const int16_t* s, intptr_t ss) { copy_sp_wh(d, ds, s, ss, W, H); } \
static void copyps_##W##x##H(int16_t* d, intptr_t ds, const pixel* s, intptr_t ss) { copy_ps_wh(d, ds, s, ss, W, H); } \
static void copyss_##W##x##H(int16_t* d, intptr_t ds, const int16_t* s, intptr_t ss) { copy_ss_wh(d, ds, s, ss, W, H); } \
static void subps_##W##x##H(int16_t* d, intptr_t ds, const pixel* a, const pixel* b, intptr_t sa, intptr_t sb) { sub_ps_wh(d, ds, a, b, sa, sb, W, H); } \
static void addps_##W##x##H(pixel* d, intptr_t ds, const pixel* a, const int16_t* r, intptr_t sa, intptr_t sr) { add_ps_wh(d, ds, a, r, sa, sr, W, H); } \
static sse_t ssepp_##W##x##H(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sse_pp_wh(a, sa, b, sb, W, H); }

#define DEF_FILTERS(N, TAG, W, H) \
static void TAG##hpp_##W##x##H(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx) { interp_hpp(s, ss, d, ds, idx, N, W, H); } \
static void TAG##hps_##W##x##H(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx, int ext) { interp_hps(s, ss, d, ds, idx, ext, N, W, H); } \
static void TAG##vpp_##W##x##H(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx) { interp_vpp(s, ss, d, ds, idx, N, W, H); } \
static void TAG##vps_##W##x##H(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx) { interp_vps(s, ss, d, ds, idx, N, W, H); } \
static void TAG##vsp_##W##x##H(const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int idx) { interp_vsp(s, ss, d, ds, idx, N, W, H); } \
static void TAG##vss_##W##x##H(const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int idx) { interp_vss(s, ss, d, ds, idx, N, W, H); }

/* every distinct (W,H) that appears as a luma PU or as a 4:2:0 / 4:2:2 chroma PU / CU */
#define ALL_DIMS(X) PU_LIST(X) \
    X(2,2) X(4,2) X(2,4) X(8,6) X(6,8) X(8,2) X(2,8) X(2,16) X(6,16) X(8,12) X(4,32) X(12,32) X(16,24) X(8,64) \
    X(24,64) X(32,48)
ALL_DIMS(DEF_BLOCK)

#define DEF_LUMA_FILTERS(W, H) DEF_FILTERS(8, l, W, H) \
static void lhvpp_##W##x##H(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int ix, int iy) { interp_hvpp(s, ss, d, ds, ix, iy, 8, W, H); } \
static int ads_##W##x##H(int* encDC, uint32_t* sums, int delta, uint16_t* costMvX, int16_t* mvs, int width, int thresh) \
{ return ads_n(ADS_N_##W##x##H, W, encDC, sums, delta, costMvX, mvs, width, thresh); }
PU_LIST(DEF_LUMA_FILTERS)
#define DEF_CHROMA_FILTERS(W, H) DEF_FILTERS(4, c, W, H)
ALL_DIMS(DEF_CHROMA_FILTERS)

#define DEF_CU(N, L2) \
static void dct_##N(const int16_t* s, int16_t* d, intptr_t ss) { dct_n(s, d, ss, N, L2); } \
static void idct_##N(const int16_t* s, int16_t* d, intptr_t ds) { idct_n(s, d, ds, N); } \
static void calcres_##N(const pixel* f, const pixel* p, int16_t* r, intptr_t st) { calcresidual_n(f, p, r, st, N); } \
static void blockfill_##N(int16_t* d, intptr_t ds, int16_t v) { blockfill_n(d, ds, v, N); } \
static uint32_t copycnt_##N(int16_t* c, const int16_t* r, intptr_t rs) { return copy_cnt_n(c, r, rs, N); } \
static int cntnz_##N(const int16_t* q) { return count_nonzero_n(q, N); } \
static void c2d1d_shl_##N(int16_t* d, const int16_t* s, intptr_t ss, int sh) { cpy2dto1d_shl_n(d, s, ss, sh, N); } \
static void c2d1d_shr_##N(int16_t* d, const int16_t* s, intptr_t ss, int sh) { cpy2dto1d_shr_n(d, s, ss, sh, N); } \
static void c1d2d_shl_##N(int16_t* d, const int16_t* s, intptr_t ds, int sh) { cpy1dto2d_shl_n(d, s, ds, sh, N); } \
static void c1d2d_shr_##N(int16_t* d, const int16_t* s, intptr_t ds, int sh) { cpy1dto2d_shr_n(d, s, ds, sh, N); } \
static uint64_t var_##N(const pixel* p, intptr_t s) { return var_n(p, s, N); } \
static sse_t ssess_##N(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb) { return sse_ss_wh(a, sa, b, sb, N, N); } \
static int psy_##N(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return psy_cost_n(a, sa, b, sb, N); } \
static sse_t ssds_##N(const int16_t* a, intptr_t sa) { return ssd_s_n(a, sa, N); } \
static void transp_##N(pixel* d, const pixel* s, intptr_t ss) { transpose_n(d, s, ss, N); } \
static void ssimdist_##N(const pixel* f, uint32_t fs, const pixel* r, intptr_t rs, uint64_t* ssb, int sh, uint64_t* ac) { ssim_dist_n(f, fs, r, rs, ssb, sh, ac, N); }
CU_LIST(DEF_CU)

#define DEF_TU(N, L2) \
static void ifilt_##N(const pixel* s, pixel* f) { intra_filter_n(s, f, N); } \
static void idc_##N(pixel* d, intptr_t ds, const pixel* nb, int m, int bf) { intra_dc_n(d, ds, nb, m, bf, N); } \
static void iplanar_##N(pixel* d, intptr_t ds, const pixel* nb, int m, int bf) { intra_planar_n(d, ds, nb, m, bf, N, L2); } \
static void iang_##N(pixel* d, intptr_t ds, const pixel* nb, int m, int bf) { intra_ang_n(d, ds, nb, m, bf, N); } \
static void iall_##N(pixel* d, pixel* r, pixel* f, int bl) { intra_allangs_n(d, r, f, bl, N); }
TU_LIST(DEF_TU)

static int sa8d_4(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return satd_wh(a, sa, b, sb, 4, 4); }
static int sa8d_8(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sa8d_8x8(a, sa, b, sb); }
static int sa8d_16(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sa8d_16x16(a, sa, b, sb); }
static int sa8d_32(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sa8d_units16(a, sa, b, sb, 32, 32); }
static int sa8d_64(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sa8d_units16(a, sa, b, sb, 64, 64); }
/* 4:2:2 chroma CU costs (pixel.cpp:1322-1325) */
static int sa8d_8x16(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sa8d_units8(a, sa, b, sb, 8, 16); }
static int sa8d_16x32(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sa8d_units16(a, sa, b, sb, 16, 32); }
static int sa8d_32x64(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return sa8d_units16(a, sa, b, sb, 32, 64); }

#define DEF_INTEGRAL(N) \
static void integh_##N(uint32_t* sum, pixel* pix, intptr_t stride) { integral_h_n(sum, pix, stride, N); } \
static void integv_##N(uint32_t* sum, intptr_t stride) { integral_v_n(sum, stride, N); }
DEF_INTEGRAL(4) DEF_INTEGRAL(8) DEF_INTEGRAL(12) DEF_INTEGRAL(16) DEF_INTEGRAL(24) DEF_INTEGRAL(32)

/* ================================================================== table filler */
/* Produces the same slot population as the reference TestBench's `cprim`
 * (setupCPrimitives + setupAliasPrimitives, primitives.cpp:63-73,88-209) for every slot this
 * restatement covers; rows a9/a16 stay NULL here. */
#define CHROMA_OK(W, H) (((W) % 4 == 0) && ((H) % 4 == 0))

void EXPORT(x265oracle_setup_primitives)(x265hip_EncoderPrimitives* p)
{
    init_tables();
    memset(p, 0, sizeof(*p));

#define SET_PU(W, H) { struct x265hip_PU* u = &p->pu[X265HIP_LUMA_##W##x##H]; \
    u->sad = sad_##W##x##H; u->sad_x3 = sadx3_##W##x##H; u->sad_x4 = sadx4_##W##x##H; u->ads = ads_##W##x##H; \
    u->satd = satd_##W##x##H; u->luma_hpp = lhpp_##W##x##H; u->luma_hps = lhps_##W##x##H; u->luma_vpp = lvpp_##W##x##H; \
    u->luma_vps = lvps_##W##x##H; u->luma_vsp = lvsp_##W##x##H; u->luma_vss = lvss_##W##x##H; u->luma_hvpp = lhvpp_##W##x##H; \
    u->pixelavg_pp[0] = u->pixelavg_pp[1] = pixelavg_##W##x##H; u->addAvg[0] = u->addAvg[1] = addavg_##W##x##H; \
    u->copy_pp = copypp_##W##x##H; u->convert_p2s[0] = u->convert_p2s[1] = p2s_##W##x##H; }
    PU_LIST(SET_PU)

#define SET_CU(N, L2) { struct x265hip_CU* c = &p->cu[L2 - 2]; \
    c->calcresidual[0] = c->calcresidual[1] = calcres_##N; c->sub_ps = subps_##N##x##N; \
    c->add_ps[0] = c->add_ps[1] = addps_##N##x##N; c->blockfill_s[0] = c->blockfill_s[1] = blockfill_##N; \
    c->cpy2Dto1D_shl = c2d1d_shl_##N; c->cpy2Dto1D_shr = c2d1d_shr_##N; \
    c->cpy1Dto2D_shl[0] = c->cpy1Dto2D_shl[1] = c1d2d_shl_##N; c->cpy1Dto2D_shr = c1d2d_shr_##N; \
    c->copy_sp = copysp_##N##x##N; c->copy_ps = copyps_##N##x##N; c->copy_ss = copyss_##N##x##N; c->copy_pp = copypp_##N##x##N; \
    c->var = var_##N; c->sse_pp = ssepp_##N##x##N; c->sse_ss = ssess_##N; c->psy_cost_pp = psy_##N; \
    c->ssd_s[0] = c->ssd_s[1] = ssds_##N; c->sa8d = sa8d_##N; c->transpose = transp_##N; c->ssimDist = ssimdist_##N; \
    if (N > 4) c->normFact = norm_fact; }
    CU_LIST(SET_CU)

    /* TU-only slots exist for 4..32 (dct.cpp:1073-1120, intrapred.cpp:240-269) */
#define SET_TU(N, L2) { struct x265hip_CU* c = &p->cu[L2 - 2]; \
    c->dct = dct_##N; c->idct = idct_##N; c->standard_dct = dct_##N; \
    c->copy_cnt = copycnt_##N; c->count_nonzero = cntnz_##N; \
    c->intra_filter = ifilt_##N; c->intra_pred_allangs = iall_##N; \
    c->intra_pred[0] = iplanar_##N; c->intra_pred[1] = idc_##N; \
    for (int m = 2; m < 35; m++) c->intra_pred[m] = iang_##N; }
    TU_LIST(SET_TU)

    p->cu[1].lowpass_dct = lowpass_8; p->cu[2].lowpass_dct = lowpass_16; p->cu[3].lowpass_dct = lowpass_32;
    p->dst4x4 = dst4; p->idst4x4 = idst4;
    p->quant = quant; p->nquant = nquant; p->dequant_scaling = dequant_scaling; p->dequant_normal = dequant_normal;
    p->denoiseDct = denoise_dct;
    p->scale1D_128to64[0] = p->scale1D_128to64[1] = scale1d_128to64; p->scale2D_64to32 = scale2d_64to32;
    p->sign = sao_sign; p->saoCuOrgE0 = sao_e0; p->saoCuOrgE1 = sao_e1; p->saoCuOrgE1_2Rows = sao_e1_2rows;
    p->saoCuOrgE2[0] = p->saoCuOrgE2[1] = sao_e2; p->saoCuOrgE3[0] = p->saoCuOrgE3[1] = sao_e3; p->saoCuOrgB0 = sao_b0;
    p->saoCuStatsBO = sao_stats_bo; p->saoCuStatsE0 = sao_stats_e0; p->saoCuStatsE1 = sao_stats_e1;
    p->saoCuStatsE2 = sao_stats_e2; p->saoCuStatsE3 = sao_stats_e3;
    p->extendRowBorder = extend_row_border;
    p->weight_sp = weight_sp; p->weight_pp = weight_pp;
    p->pelFilterLumaStrong[0] = p->pelFilterLumaStrong[1] = deblock_luma_strong;
    p->pelFilterChroma[0] = p->pelFilterChroma[1] = deblock_chroma;
#define SET_INTEG(I, N) p->integral_initv[I] = integv_##N; p->integral_inith[I] = integh_##N;
    SET_INTEG(0, 4) SET_INTEG(1, 8) SET_INTEG(2, 12) SET_INTEG(3, 16) SET_INTEG(4, 24) SET_INTEG(5, 32)

    /* ---- chroma tables, indexed by the LUMA enum (primitives.h:77-79,393-428) ---- */
#define SET_CHROMA_PU(CSP, W, H, CW, CH) { struct x265hip_PUChroma* u = &p->chroma[CSP].pu[X265HIP_LUMA_##W##x##H]; \
    u->satd = CHROMA_OK(CW, CH) ? satd_##CW##x##CH : NULL; \
    u->filter_vpp = cvpp_##CW##x##CH; u->filter_vps = cvps_##CW##x##CH; u->filter_vsp = cvsp_##CW##x##CH; \
    u->filter_vss = cvss_##CW##x##CH; u->filter_hpp = chpp_##CW##x##CH; u->filter_hps = chps_##CW##x##CH; \
    u->addAvg[0] = u->addAvg[1] = addavg_##CW##x##CH; u->copy_pp = copypp_##CW##x##CH; u->p2s[0] = u->p2s[1] = p2s_##CW##x##CH; }
    /* 4:2:0 : half width, half height */
    SET_CHROMA_PU(1, 4,4, 2,2) SET_CHROMA_PU(1, 8,8, 4,4) SET_CHROMA_PU(1, 16,16, 8,8) SET_CHROMA_PU(1, 32,32, 16,16) SET_CHROMA_PU(1, 64,64, 32,32)
    SET_CHROMA_PU(1, 8,4, 4,2) SET_CHROMA_PU(1, 4,8, 2,4) SET_CHROMA_PU(1, 16,8, 8,4) SET_CHROMA_PU(1, 8,16, 4,8)
    SET_CHROMA_PU(1, 32,16, 16,8) SET_CHROMA_PU(1, 16,32, 8,16) SET_CHROMA_PU(1, 64,32, 32,16) SET_CHROMA_PU(1, 32,64, 16,32)
    SET_CHROMA_PU(1, 16,12, 8,6) SET_CHROMA_PU(1, 12,16, 6,8) SET_CHROMA_PU(1, 16,4, 8,2) SET_CHROMA_PU(1, 4,16, 2,8)
    SET_CHROMA_PU(1, 32,24, 16,12) SET_CHROMA_PU(1, 24,32, 12,16) SET_CHROMA_PU(1, 32,8, 16,4) SET_CHROMA_PU(1, 8,32, 4,16)
    SET_CHROMA_PU(1, 64,48, 32,24) SET_CHROMA_PU(1, 48,64, 24,32) SET_CHROMA_PU(1, 64,16, 32,8) SET_CHROMA_PU(1, 16,64, 8,32)
    /* the reference never instantiates 2x2 interpolation / p2s (ipfilter.cpp:416-521 has no CHROMA_420(2, 2)) */
    { struct x265hip_PUChroma* u = &p->chroma[1].pu[X265HIP_LUMA_4x4];
      u->filter_vpp = NULL; u->filter_vps = NULL; u->filter_vsp = NULL; u->filter_vss = NULL;
      u->filter_hpp = NULL; u->filter_hps = NULL; u->p2s[0] = u->p2s[1] = NULL; }
    /* 4:2:2 : half width, full height */
    SET_CHROMA_PU(2, 4,4, 2,4) SET_CHROMA_PU(2, 8,8, 4,8) SET_CHROMA_PU(2, 16,16, 8,16) SET_CHROMA_PU(2, 32,32, 16,32) SET_CHROMA_PU(2, 64,64, 32,64)
    SET_CHROMA_PU(2, 8,4, 4,4) SET_CHROMA_PU(2, 4,8, 2,8) SET_CHROMA_PU(2, 16,8, 8,8) SET_CHROMA_PU(2, 8,16, 4,16)
    SET_CHROMA_PU(2, 32,16, 16,16) SET_CHROMA_PU(2, 16,32, 8,32) SET_CHROMA_PU(2, 64,32, 32,32) SET_CHROMA_PU(2, 32,64, 16,64)
    SET_CHROMA_PU(2, 16,12, 8,12) SET_CHROMA_PU(2, 12,16, 6,16) SET_CHROMA_PU(2, 16,4, 8,4) SET_CHROMA_PU(2, 4,16, 2,16)
    SET_CHROMA_PU(2, 32,24, 16,24) SET_CHROMA_PU(2, 24,32, 12,32) SET_CHROMA_PU(2, 32,8, 16,8) SET_CHROMA_PU(2, 8,32, 4,32)
    SET_CHROMA_PU(2, 64,48, 32,48) SET_CHROMA_PU(2, 48,64, 24,64) SET_CHROMA_PU(2, 64,16, 32,16) SET_CHROMA_PU(2, 16,64, 8,64)
    /* 4:4:4 : luma sizes, 4-tap filters; everything else aliases luma (primitives.cpp:110-134) */
#define SET_CHROMA_PU444(W, H) { struct x265hip_PUChroma* u = &p->chroma[3].pu[X265HIP_LUMA_##W##x##H]; \
    u->satd = satd_##W##x##H; u->filter_vpp = cvpp_##W##x##H; u->filter_vps = cvps_##W##x##H; u->filter_vsp = cvsp_##W##x##H; \
    u->filter_vss = cvss_##W##x##H; u->filter_hpp = chpp_##W##x##H; u->filter_hps = chps_##W##x##H; \
    u->addAvg[0] = u->addAvg[1] = addavg_##W##x##H; u->copy_pp = copypp_##W##x##H; u->p2s[0] = u->p2s[1] = p2s_##W##x##H; }
    PU_LIST(SET_CHROMA_PU444)

#define SET_CHROMA_CU(CSP, IDX, CW, CH) { struct x265hip_CUChroma* c = &p->chroma[CSP].cu[IDX]; \
    c->sse_pp = ssepp_##CW##x##CH; c->sub_ps = subps_##CW##x##CH; c->add_ps[0] = c->add_ps[1] = addps_##CW##x##CH; \
    c->copy_ps = copyps_##CW##x##CH; c->copy_sp = copysp_##CW##x##CH; c->copy_ss = copyss_##CW##x##CH; c->copy_pp = copypp_##CW##x##CH; }
    SET_CHROMA_CU(1, 0, 2,2) SET_CHROMA_CU(1, 1, 4,4) SET_CHROMA_CU(1, 2, 8,8) SET_CHROMA_CU(1, 3, 16,16) SET_CHROMA_CU(1, 4, 32,32)
    SET_CHROMA_CU(2, 0, 2,4) SET_CHROMA_CU(2, 1, 4,8) SET_CHROMA_CU(2, 2, 8,16) SET_CHROMA_CU(2, 3, 16,32) SET_CHROMA_CU(2, 4, 32,64)
    SET_CHROMA_CU(3, 0, 4,4) SET_CHROMA_CU(3, 1, 8,8) SET_CHROMA_CU(3, 2, 16,16) SET_CHROMA_CU(3, 3, 32,32) SET_CHROMA_CU(3, 4, 64,64)
    /* sub-4x4 chroma CUs have no sse/sa8d (primitives.cpp:184-208) */
    p->chroma[1].cu[0].sse_pp = NULL; p->chroma[2].cu[0].sse_pp = NULL;
    p->chroma[1].cu[0].sa8d = NULL; p->chroma[1].cu[1].sa8d = sa8d_4; p->chroma[1].cu[2].sa8d = sa8d_8;
    p->chroma[1].cu[3].sa8d = sa8d_16; p->chroma[1].cu[4].sa8d = sa8d_32;
    p->chroma[2].cu[0].sa8d = NULL; p->chroma[2].cu[1].sa8d = satd_4x8; p->chroma[2].cu[2].sa8d = sa8d_8x16;
    p->chroma[2].cu[3].sa8d = sa8d_16x32; p->chroma[2].cu[4].sa8d = sa8d_32x64;
    p->chroma[3].cu[0].sa8d = sa8d_4; p->chroma[3].cu[1].sa8d = sa8d_8; p->chroma[3].cu[2].sa8d = sa8d_16;
    p->chroma[3].cu[3].sa8d = sa8d_32; p->chroma[3].cu[4].sa8d = sa8d_64;
}

int EXPORT(x265oracle_depth)(void) { return DEPTH; }


/* Thread-safe lazy fill of a stage's private table: the stage restatements are called from several threads at once by the seam tests
 * (an unguarded `if (!ready) { setup; ready = 1; }` lets a late thread re-fill the table under an early thread's feet).
 * state: 0 = empty, 1 = being filled, 2 = ready. */
void EXPORT(x265oracle_prims_once)(x265hip_EncoderPrimitives* p, int* state)
{
    if (__atomic_load_n(state, __ATOMIC_ACQUIRE) == 2) return;
    int expected = 0;
    if (__atomic_compare_exchange_n(state, &expected, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
    {
        EXPORT(x265oracle_setup_primitives)(p);
        __atomic_store_n(state, 2, __ATOMIC_RELEASE);
    }
    else
        while (__atomic_load_n(state, __ATOMIC_ACQUIRE) != 2) { }
}
