/* oracle/x265_oracle_bench.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.  CPU timing harness for tools/bench_prims.py: runs one table slot of the
 * oracle (or of the real reference build, oracle/_ref) over a job list of the same shape the HIP batch layer takes
 * (x265hip_job: 4 element offsets + 4 int args into operand planes), OpenMP-parallel over jobs, and returns the
 * wall time per pass.  This is the "CPU path timed beside it" of SURVEY.md section 8(d) at primitive level.
 */
#include <stdint.h>
#include <stddef.h>
#include <time.h>
#include <omp.h>

typedef struct { void* base; intptr_t stride; int elem; } bplane;       /* elem = bytes per element */
typedef struct { int64_t off[4]; int32_t arg[4]; } bjob;

enum {
    SIG_PIXELCMP = 0,   /* int f(a, sa, b, sb)                          p0 = a, p1 = b                      */
    SIG_SAD_X4,         /* void f(fenc, r0, r1, r2, r3, stride, res[4]) p0 = fenc, p1 = ref (4 columns)     */
    SIG_FILTER,         /* void f(src, ss, dst, ds, idx)                p0 = src, p1 = dst, arg0 = idx      */
    SIG_FILTER_HPS,     /* void f(src, ss, dst, ds, idx, rowExt)                                            */
    SIG_FILTER_HV,      /* void f(src, ss, dst, ds, idxX, idxY)                                             */
    SIG_P2S,            /* void f(src, ss, dst, ds)                                                         */
    SIG_DCT,            /* void f(src, dst, stride): stride = p0 (dct) or p1 (idct) stride, arg3 selects   */
    SIG_QUANT,          /* u32 f(coef, quantCoeff, deltaU, qCoef, qBits, add, n)   p0..p3                   */
    SIG_NQUANT,         /* u32 f(coef, quantCoeff, qCoef, qBits, add, n)           p0, p1, p3               */
    SIG_DEQUANT_NORMAL, /* void f(quantCoef, coef, n, scale, shift)                p0, p3                   */
    SIG_INTRA_PRED,     /* void f(dst, ds, srcPix, mode, filter)                   p0 = srcPix, p1 = dst    */
    SIG_INTRA_ALLANGS,  /* void f(dst, refPix, filtPix, luma)   p0 = refPix(off0)/filtPix(off2), p1 = dst   */
    SIG_COPY,           /* void f(dst, ds, src, ss)                                p0 = dst, p1 = src       */
    SIG_SUB_PS,         /* void f(dst, ds, a, b, sa, sb)                           p0 = dst, p1 = a, p2 = b */
    SIG_ADD_PS,         /* void f(dst, ds, a, r, sa, sr)                                                    */
    SIG_ADDAVG,         /* void f(a, b, dst, sa, sb, ds)                           p0 = dst, p1 = a, p2 = b */
    SIG_SAO_E0,         /* void f(rec, offsetEo, width, signLeft, stride)  p0 = rec, p1 = offsets(5 x int8), p2 = signLeft */
    SIG_SAO_B0,         /* void f(rec, offsetBo, w, h, stride)             p0 = rec, p1 = offsets (32 x int8) */
    SIG_DEBLOCK_LUMA,   /* void f(src, srcStep, offset, tcP, tcQ)          p0 = src; arg = {srcStep, offset, tcP, tcQ} */
    SIG_VAR,            /* u64 f(pix, stride)                                                               */
    SIG_CALCRES,        /* void f(fenc, pred, resi, stride)   p0 = fenc, p1 = pred, p2 = resi (same stride) */
};

#define AT(k) ((char*)pl[k].base + jb->off[k] * (int64_t)pl[k].elem)

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* returns seconds per pass (mean of `reps` passes after one warm-up); *sink receives a checksum so nothing is elided */
double x265oracle_time_jobs(void* fn, int sig, const bplane* pl, const bjob* jobs, int njobs, int reps, int threads, uint64_t* sink)
{
    if (threads > 0) omp_set_num_threads(threads);
    uint64_t acc = 0;
    double t0 = 0.0;
    for (int rep = -1; rep < reps; rep++)
    {
        if (rep == 0) t0 = now();
#pragma omp parallel for schedule(static) reduction(+ : acc)
        for (int j = 0; j < njobs; j++)
        {
            const bjob* jb = &jobs[j];
            switch (sig)
            {
            case SIG_PIXELCMP:
                acc += (uint64_t)((int (*)(const void*, intptr_t, const void*, intptr_t))fn)(AT(0), pl[0].stride, AT(1), pl[1].stride); break;
            case SIG_SAD_X4:
            {
                int32_t res[4];
                const char* r = AT(1);
                ((void (*)(const void*, const void*, const void*, const void*, const void*, intptr_t, int32_t*))fn)
                    (AT(0), r, r + pl[1].elem, r + 2 * pl[1].elem, r + 3 * pl[1].elem, pl[1].stride, res);
                acc += (uint64_t)(res[0] + res[1] + res[2] + res[3]);
                break;
            }
            case SIG_FILTER:
                ((void (*)(const void*, intptr_t, void*, intptr_t, int))fn)(AT(0), pl[0].stride, AT(1), pl[1].stride, jb->arg[0]); break;
            case SIG_FILTER_HPS:
                ((void (*)(const void*, intptr_t, void*, intptr_t, int, int))fn)(AT(0), pl[0].stride, AT(1), pl[1].stride, jb->arg[0], jb->arg[1]); break;
            case SIG_FILTER_HV:
                ((void (*)(const void*, intptr_t, void*, intptr_t, int, int))fn)(AT(0), pl[0].stride, AT(1), pl[1].stride, jb->arg[0], jb->arg[1]); break;
            case SIG_P2S:
                ((void (*)(const void*, intptr_t, void*, intptr_t))fn)(AT(0), pl[0].stride, AT(1), pl[1].stride); break;
            case SIG_DCT:
                ((void (*)(const void*, void*, intptr_t))fn)(AT(0), AT(1), jb->arg[3] ? pl[1].stride : pl[0].stride); break;
            case SIG_QUANT:
                acc += ((uint32_t (*)(const void*, const void*, void*, void*, int, int, int))fn)(AT(0), AT(1), AT(2), AT(3), jb->arg[0], jb->arg[1], jb->arg[2]); break;
            case SIG_NQUANT:
                acc += ((uint32_t (*)(const void*, const void*, void*, int, int, int))fn)(AT(0), AT(1), AT(3), jb->arg[0], jb->arg[1], jb->arg[2]); break;
            case SIG_DEQUANT_NORMAL:
                ((void (*)(const void*, void*, int, int, int))fn)(AT(0), AT(3), jb->arg[0], jb->arg[1], jb->arg[2]); break;
            case SIG_INTRA_PRED:
                ((void (*)(void*, intptr_t, const void*, int, int))fn)(AT(1), pl[1].stride, AT(0), jb->arg[0], jb->arg[1]); break;
            case SIG_INTRA_ALLANGS:
                ((void (*)(void*, void*, void*, int))fn)(AT(1), AT(0), (char*)pl[0].base + jb->off[2] * (int64_t)pl[0].elem, jb->arg[0]); break;
            case SIG_COPY:
                ((void (*)(void*, intptr_t, const void*, intptr_t))fn)(AT(0), pl[0].stride, AT(1), pl[1].stride); break;
            case SIG_SUB_PS:
            case SIG_ADD_PS:
                ((void (*)(void*, intptr_t, const void*, const void*, intptr_t, intptr_t))fn)(AT(0), pl[0].stride, AT(1), AT(2), pl[1].stride, pl[2].stride); break;
            case SIG_ADDAVG:
                ((void (*)(const void*, const void*, void*, intptr_t, intptr_t, intptr_t))fn)(AT(1), AT(2), AT(0), pl[1].stride, pl[2].stride, pl[0].stride); break;
            case SIG_SAO_E0:
                ((void (*)(void*, void*, int, void*, intptr_t))fn)(AT(0), AT(1), jb->arg[0], AT(2), pl[0].stride); break;
            case SIG_SAO_B0:
                ((void (*)(void*, const void*, int, int, intptr_t))fn)(AT(0), AT(1), jb->arg[0], jb->arg[1], pl[0].stride); break;
            case SIG_DEBLOCK_LUMA:
                ((void (*)(void*, intptr_t, intptr_t, int32_t, int32_t))fn)(AT(0), jb->arg[0], jb->arg[1], jb->arg[2], jb->arg[3]); break;
            case SIG_VAR:
                acc += ((uint64_t (*)(const void*, intptr_t))fn)(AT(0), pl[0].stride); break;
            case SIG_CALCRES:
                ((void (*)(const void*, const void*, void*, intptr_t))fn)(AT(0), AT(1), AT(2), pl[0].stride); break;
            default: break;
            }
        }
    }
    const double dt = (now() - t0) / (reps > 0 ? reps : 1);
    if (sink) *sink = acc;
    return dt;
}

int x265oracle_max_threads(void) { return omp_get_max_threads(); }

/* ---- the RDOQ helper slots (round 6): the same harness for x265hip_coeff_job lists (5 element offsets + 5 ints into five buffers, include/x265hip.h).
 * kind = x265hip_coeff_kind; every job owns its context bytes, so passes may run in any order (the contexts drift from pass to pass - timing only). */
typedef struct { int64_t off[5]; int32_t arg[5]; int32_t reserved; } cjob;

double x265oracle_time_coeff_jobs(void* fn, int kind, void* const* bufs, const cjob* jobs, int njobs, int reps, int threads, uint64_t* sink)
{
    if (threads > 0) omp_set_num_threads(threads);
    uint64_t acc = 0;
    double t0 = 0.0;
    for (int rep = -1; rep < reps; rep++)
    {
        if (rep == 0) t0 = now();
#pragma omp parallel for schedule(static) reduction(+ : acc)
        for (int j = 0; j < njobs; j++)
        {
            const cjob* jb = &jobs[j];
            uint16_t* b0 = (uint16_t*)bufs[0] + jb->off[0]; int16_t* b1 = (int16_t*)bufs[1] + jb->off[1];
            switch (kind)
            {
            case 0:   /* scanPosLast(scan, coeff, coeffSign, coeffFlag, coeffNum, numSig, scanCG4x4, trSize) */
                acc += (uint64_t)((int (*)(const uint16_t*, const int16_t*, uint16_t*, uint16_t*, uint8_t*, int, const uint16_t*, int))fn)(
                    b0, b1, (uint16_t*)bufs[2] + jb->off[2], (uint16_t*)bufs[3] + jb->off[3], (uint8_t*)bufs[4] + jb->off[4], jb->arg[0], b0, jb->arg[1]);
                break;
            case 1:   /* findPosFirstLast(dstCoeff, trSize, scanTbl) */
                acc += ((uint32_t (*)(const int16_t*, intptr_t, const uint16_t*))fn)(b1, jb->arg[0], b0);
                break;
            case 2:   /* costCoeffNxN(scan, coeff, trSize, absCoeff, tabSigCtx, scanFlagMask, baseCtx, offset, scanPosSigOff, subPosBase) */
                acc += ((uint32_t (*)(const uint16_t*, const int16_t*, intptr_t, uint16_t*, const uint8_t*, uint32_t, uint8_t*, int, int, int))fn)(
                    b0, b1, jb->arg[0], (uint16_t*)bufs[2] + jb->off[2], (const uint8_t*)bufs[3] + jb->off[3], (uint32_t)jb->arg[1], (uint8_t*)bufs[4] + jb->off[4],
                    jb->arg[2], jb->arg[3], jb->arg[4]);
                break;
            case 3:   /* costCoeffRemain(absCoeff, numNonZero, idx) */
                acc += ((uint32_t (*)(uint16_t*, int, int))fn)((uint16_t*)bufs[2] + jb->off[2], jb->arg[0], jb->arg[1]);
                break;
            case 4:   /* costC1C2Flag(absCoeff, numC1Flag, baseCtxMod, ctxOffset) */
                acc += ((uint32_t (*)(uint16_t*, intptr_t, uint8_t*, intptr_t))fn)((uint16_t*)bufs[2] + jb->off[2], jb->arg[0], (uint8_t*)bufs[4] + jb->off[4], jb->arg[1]);
                break;
            case 5: case 7:   /* nonPsyRdoQuant / psyRdoQuant_1p (resi, costUncoded, totalUncoded, totalRd, blkPos) */
                ((void (*)(int16_t*, int64_t*, int64_t*, int64_t*, uint32_t))fn)(b1, (int64_t*)bufs[2] + jb->off[2], (int64_t*)bufs[3] + jb->off[3], (int64_t*)bufs[3] + jb->off[3] + 1,
                                                                               (uint32_t)jb->arg[0]);
                break;
            default:  /* psyRdoQuant / psyRdoQuant_2p (resi, fenc, costUncoded, totalUncoded, totalRd, psyScale, blkPos) */
                ((void (*)(int16_t*, int16_t*, int64_t*, int64_t*, int64_t*, int64_t*, uint32_t))fn)(b1, (int16_t*)b0, (int64_t*)bufs[2] + jb->off[2], (int64_t*)bufs[3] + jb->off[3],
                                                                                                   (int64_t*)bufs[3] + jb->off[3] + 1, (int64_t*)bufs[4] + jb->off[4], (uint32_t)jb->arg[0]);
                break;
            }
        }
    }
    if (sink) *sink = acc;
    return (now() - t0) / (reps > 0 ? reps : 1);
}
