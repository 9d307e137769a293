// tile_order.h — enumeration of the output tiles of the trailing update, shared by the host
// (tile count) and the device (tile index -> coordinates).  Plain C++: also compiled by the CPU
// unit test tests/test_tile_order.py through g++.
//
// Which tiles exist is described by TileShape:
//   ntm x ntn tile grid; row ti keeps columns tj <= cmax(ti) where
//     mode 0 (rectangle)   cmax = ntn - 1
//     mode 1 (lower)       cmax = ti + g0               (C is a trailing square, single GPU; g0 > 0: the block
//                                                       starts g0 tile-rows below the square's top)
//     mode 2 (staircase)   cmax = tpb*(g0 + (ti/tpb)*G) + ti%tpb for ti < nstair, ntn - 1 beyond
//                          (row-block-cyclic shard: local block i — tpb 128-row tiles, 2 for the 256-row blocks of round
//                           1, 4 / 8 / 16 for the super-panel blocks — is global block g0 + i*G; rows past the staircase
//                           are carried rows)
//     mode 3 (lower, half-width column tiles: 128 x 64 output tiles)   cmax = 2*(ti + g0) + 1
// Order: strips of GROUP tile-rows.  Modes 0/1: column-major inside a strip, so GROUP consecutive
// tiles share one B panel and the strip's GROUP A panels stay hot.  Mode 2: row-major inside a strip.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define GPMI_HD __host__ __device__ __forceinline__
#else
#define GPMI_HD inline
#endif

namespace gpmi {

constexpr int TILE_GROUP = 8;

struct TileShape {
    int ntm, ntn, mode;
    int g0, G, nstair;  // mode 2 only (g0 also: row offset of the lower modes)
    int tpb = 2;        // mode 2: 128-row tiles per distributed block
};

GPMI_HD int stair_cmax(const TileShape& s, int ti) {
    int c = s.ntn - 1;
    if (ti < s.nstair) {
        const int tpb = s.tpb > 0 ? s.tpb : 2;
        const int64_t v = (int64_t)tpb * ((int64_t)s.g0 + (int64_t)(ti / tpb) * s.G) + (ti % tpb);
        if (v < c) c = (int)v;
    }
    return c;
}

// modes 1 / 3: number of tiles in tile-rows < r  (row i keeps min(i + g0 + 1, ntn), resp. min(2 (i + g0) + 2, ntn) tiles)
GPMI_HD int64_t lower_tiles_before_row(int r, const TileShape& s) {
    const int64_t off = s.g0;
    if (s.mode == 3) {
        int64_t rt = (int64_t)s.ntn / 2 - off;  // rows below rt are not capped by ntn
        if (rt < 0) rt = 0;
        if (rt > r) rt = r;
        return rt * rt + (2 * off + 1) * rt + ((int64_t)r - rt) * s.ntn;
    }
    int64_t rt = (int64_t)s.ntn - off - 1;  // first row that already holds all ntn columns
    if (rt < 0) rt = 0;
    if (rt > r) rt = r;
    return rt * (rt + 1) / 2 + off * rt + ((int64_t)r - rt) * s.ntn;
}

GPMI_HD int64_t strip_count(int st, const TileShape& s) {
    const int r0 = st * TILE_GROUP;
    const int h = (s.ntm - r0 < TILE_GROUP) ? (s.ntm - r0) : TILE_GROUP;
    if (s.mode == 0) return (int64_t)h * s.ntn;
    if (s.mode == 2) {
        int64_t c = 0;
        for (int i = 0; i < h; ++i) c += stair_cmax(s, r0 + i) + 1;
        return c;
    }
    const int off = s.g0;
    if (s.mode == 3) return lower_tiles_before_row(r0 + h, s) - lower_tiles_before_row(r0, s);
    const int nfull = (r0 + off + 1 < s.ntn) ? (r0 + off + 1) : s.ntn;
    int64_t c = (int64_t)h * nfull;
    const int jmax = (r0 + h - 1 + off < s.ntn - 1) ? (r0 + h - 1 + off) : (s.ntn - 1);
    for (int tj = nfull; tj <= jmax; ++tj) c += r0 + h - (tj - off);
    return c;
}

GPMI_HD int64_t tile_count(const TileShape& s) {
    int64_t c = 0;
    const int ns = (s.ntm + TILE_GROUP - 1) / TILE_GROUP;
    for (int st = 0; st < ns; ++st) c += strip_count(st, s);
    return c;
}

// t in [0, tile_count) -> (ti, tj)
GPMI_HD void tile_decode(int64_t t, const TileShape& s, int* ti, int* tj) {
    if (s.mode == 0) {  // every full strip holds TILE_GROUP * ntn tiles: no scan (a 1e6-row product has ~1000 strips)
        const int64_t per = (int64_t)TILE_GROUP * s.ntn;
        const int st0 = (int)(t / per);
        t -= (int64_t)st0 * per;
        const int r00 = st0 * TILE_GROUP;
        const int h0 = (s.ntm - r00 < TILE_GROUP) ? (s.ntm - r00) : TILE_GROUP;
        *tj = (int)(t / h0);
        *ti = r00 + (int)(t % h0);
        return;
    }
    int st = 0;
    if (s.mode == 1 || s.mode == 3) {  // closed-form prefix count + bisection over the strips (782 tile rows at N = 100 000)
        const int ns = (s.ntm + TILE_GROUP - 1) / TILE_GROUP;
        int lo = 0, hi = ns - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (lower_tiles_before_row(mid * TILE_GROUP, s) <= t) lo = mid;
            else hi = mid - 1;
        }
        st = lo;
        t -= lower_tiles_before_row(st * TILE_GROUP, s);
    } else {
        for (;; ++st) {
            const int64_t c = strip_count(st, s);
            if (t < c) break;
            t -= c;
        }
    }
    const int r0 = st * TILE_GROUP;
    const int h = (s.ntm - r0 < TILE_GROUP) ? (s.ntm - r0) : TILE_GROUP;
    if (s.mode == 0) {
        *tj = (int)(t / h);
        *ti = r0 + (int)(t % h);
        return;
    }
    if (s.mode == 2) {
        for (int i = 0;; ++i) {
            const int n = stair_cmax(s, r0 + i) + 1;
            if (t < n) {
                *ti = r0 + i;
                *tj = (int)t;
                return;
            }
            t -= n;
        }
    }
    const int off = s.g0;
    if (s.mode == 3) {  // columns kept by every row of the strip first (column-major), then the staircase columns
        const int full0 = 2 * (r0 + off) + 2;
        const int nf = full0 < s.ntn ? full0 : s.ntn;
        if (t < (int64_t)h * nf) {
            *tj = (int)(t / h);
            *ti = r0 + (int)(t % h);
            return;
        }
        int q = (int)(t - (int64_t)h * nf);
        for (int c = nf;; ++c) {
            const int first = c / 2 - off;  // first row that keeps column c
            const int n = r0 + h - first;
            if (q < n) {
                *tj = c;
                *ti = first + q;
                return;
            }
            q -= n;
        }
    }
    const int nfull = (r0 + off + 1 < s.ntn) ? (r0 + off + 1) : s.ntn;
    if (t < (int64_t)h * nfull) {
        *tj = (int)(t / h);
        *ti = r0 + (int)(t % h);
        return;
    }
    int q = (int)(t - (int64_t)h * nfull);
    for (int c = nfull;; ++c) {
        const int n = r0 + h - (c - off);
        if (q < n) {
            *tj = c;
            *ti = c - off + q;
            return;
        }
        q -= n;
    }
}

}  // namespace gpmi
