// intra_sample.h - one predicted sample of an N x N HEVC intra block in closed form (shared device code).
//
// Reference: source/common/intrapred.cpp - planar_pred_c :87-100, intra_pred_dc_c + dcPredFilter :53-85, intra_pred_ang_c
// :102-204 (modes 2..17 are predicted from the swapped neighbour arms and transposed; the reference line of negative angles
// is extended with samples projected from the side arm through the inverse-angle table, :150-160; pure horizontal / vertical
// modes add the clipped gradient on the first column / row when bFilter is set).
// Neighbour layout: nb[0] corner, nb[1..2N] above + above-right, nb[2N+1..4N] left + below-left.
#pragma once
#include "common.h"

namespace x265hip {

__constant__ int8_t kIsAngle[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int16_t kIsInvAngle[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
__constant__ uint8_t kIsFilterFlags[35] = {           // constants.cpp:561 g_intraFilterFlags
    0x38, 0x00,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38 };

// dc = (sum of the N above + N left neighbours + N) / 2N, needed for mode 1 only.
template <typename Nb>
__device__ __forceinline__ int intra_sample(const Nb* nb, int n, int log2n, int mode, int bFilter, int dc, int maxVal, int x, int y)
{
    const int n2 = 2 * n;
    if (mode == 0)
        return ((n - 1 - x) * (int)nb[n2 + 1 + y] + (n - 1 - y) * (int)nb[1 + x] + (x + 1) * (int)nb[1 + n] + (y + 1) * (int)nb[n2 + 1 + n] + n) >> (log2n + 1);
    if (mode == 1)
    {
        if (bFilter)
        {
            if (x == 0 && y == 0) return ((int)nb[1] + (int)nb[n2 + 1] + 2 * dc + 2) >> 2;
            if (y == 0) return ((int)nb[1 + x] + 3 * dc + 2) >> 2;
            if (x == 0) return ((int)nb[n2 + 1 + y] + 3 * dc + 2) >> 2;
        }
        return dc;
    }
    const bool hor = mode < 18;
    const int r = hor ? x : y, c = hor ? y : x;                      // horizontal modes predict the transpose
    const int mainBase = hor ? n2 : 0, sideBase = hor ? 0 : n2;
    const int aoff = hor ? 10 - mode : mode - 26;
    const int angle = kIsAngle[8 + aoff];
    if (angle == 0)
    {
        int v = nb[mainBase + 1 + c];
        if (bFilter && c == 0)
        {
            const int16_t t = (int16_t)((int)nb[mainBase + 1] + (((int)nb[sideBase + 1 + r] - (int)nb[0]) >> 1));
            v = t < 0 ? 0 : (t > maxVal ? maxVal : t);
        }
        return v;
    }
    const int inv = angle < 0 ? kIsInvAngle[-aoff - 1] : 0;
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

} // namespace x265hip
