// tile_order.h — enumeration of the output tiles of the trailing update, shared by the host
// (tile count) and the device (tile index -> coordinates).  Plain C++: also compiled by the CPU
// unit test tests/test_tile_order.py through g++.
//
// Order: strips of GROUP tile-rows; inside a strip column-major, so GROUP consecutive tiles share
// one B panel and the strip's GROUP A panels stay hot while the strip sweeps its columns.
// lower != 0 keeps only tiles that intersect {col <= row}:  tj <= min(ti, ntn - 1).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define GPMI_HD __host__ __device__ __forceinline__
#else
#define GPMI_HD inline
#endif

namespace gpmi {

constexpr int TILE_GROUP = 8;

GPMI_HD int64_t strip_count(int s, int ntm, int ntn, int lower) {
    const int r0 = s * TILE_GROUP;
    const int h = (ntm - r0 < TILE_GROUP) ? (ntm - r0) : TILE_GROUP;
    if (!lower) return (int64_t)h * ntn;
    const int nfull = (r0 + 1 < ntn) ? (r0 + 1) : ntn;
    int64_t c = (int64_t)h * nfull;
    const int jmax = (r0 + h - 1 < ntn - 1) ? (r0 + h - 1) : (ntn - 1);
    for (int tj = r0 + 1; tj <= jmax; ++tj) c += r0 + h - tj;
    return c;
}

GPMI_HD int64_t tile_count(int ntm, int ntn, int lower) {
    int64_t c = 0;
    const int ns = (ntm + TILE_GROUP - 1) / TILE_GROUP;
    for (int s = 0; s < ns; ++s) c += strip_count(s, ntm, ntn, lower);
    return c;
}

// t in [0, tile_count) -> (ti, tj)
GPMI_HD void tile_decode(int64_t t, int ntm, int ntn, int lower, int* ti, int* tj) {
    int s = 0;
    for (;; ++s) {
        const int64_t c = strip_count(s, ntm, ntn, lower);
        if (t < c) break;
        t -= c;
    }
    const int r0 = s * TILE_GROUP;
    const int h = (ntm - r0 < TILE_GROUP) ? (ntm - r0) : TILE_GROUP;
    if (!lower) {
        *tj = (int)(t / h);
        *ti = r0 + (int)(t % h);
        return;
    }
    const int nfull = (r0 + 1 < ntn) ? (r0 + 1) : ntn;
    if (t < (int64_t)h * nfull) {
        *tj = (int)(t / h);
        *ti = r0 + (int)(t % h);
        return;
    }
    int q = (int)(t - (int64_t)h * nfull);
    for (int c = r0 + 1;; ++c) {
        const int n = r0 + h - c;
        if (q < n) {
            *tj = c;
            *ti = c + q;
            return;
        }
        q -= n;
    }
}

}  // namespace gpmi
