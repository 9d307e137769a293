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
// tiles share one B panel and the strip's GROUP A panels stay hot (all modes).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define GPMI_HD __host__ __device__ __forceinline__
#else
#define GPMI_HD inline
#endif

namespace gpmi {

#ifndef GPMI_TILE_GROUP
#define GPMI_TILE_GROUP 8  // tile-rows per strip (tools/build_variant.sh -DGPMI_TILE_GROUP=n builds an A/B library; 4 / 8 / 16 measured in round 6: profiles/r06_r_*)
#endif
constexpr int TILE_GROUP = GPMI_TILE_GROUP;

struct TileShape {
    int ntm, ntn, mode;
    int g0, G, nstair;  // mode 2 only (g0 also: row offset of the lower modes)
    int tpb = 2;        // mode 2: 128-row tiles per distributed block
    int rc = -1;        // mode 2: first tile-row that keeps every column (stair_finalize; rows from there on are 'capped')
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

// mode 2: the staircase's cmax is non-decreasing, so the rows that keep all ntn columns (diagonal at or past the last
// column, and the carried rows past the staircase) are a suffix [rc, ntm).  Set once on the host.
GPMI_HD void stair_finalize(TileShape& s) {
    if (s.mode != 2) return;
    int rc = s.nstair < s.ntm ? s.nstair : s.ntm;
    // first staircase row whose cmax reaches ntn - 1 (bisection: cmax is monotone)
    int lo = 0, hi = rc;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int tpb = s.tpb > 0 ? s.tpb : 2;
        const int64_t v = (int64_t)tpb * ((int64_t)s.g0 + (int64_t)(mid / tpb) * s.G) + (mid % tpb);
        if (v >= s.ntn - 1) hi = mid;
        else lo = mid + 1;
    }
    s.rc = lo;
}
// mode 2: number of tiles in tile-rows < r, closed form (rows < rc: block b = r / tpb contributes
// tpb^2 (g0 + b G) + tpb (tpb + 1) / 2 tiles, the partial block j = r % tpb rows of tpb (g0 + b G) + jj + 1)
GPMI_HD int64_t stair_tiles_before_row(int r, const TileShape& s) {
    const int64_t tpb = s.tpb > 0 ? s.tpb : 2;
    const int64_t ru = r < s.rc ? r : s.rc;
    const int64_t b = ru / tpb, j = ru % tpb;
    const int64_t u = tpb * tpb * (b * s.g0 + (int64_t)s.G * b * (b - 1) / 2) + b * tpb * (tpb + 1) / 2 +
                      j * tpb * ((int64_t)s.g0 + b * s.G) + j * (j + 1) / 2;
    return u + (r > s.rc ? (int64_t)(r - s.rc) * s.ntn : 0);
}

GPMI_HD int64_t strip_count(int st, const TileShape& s) {
    const int r0 = st * TILE_GROUP;
    const int h = (s.ntm - r0 < TILE_GROUP) ? (s.ntm - r0) : TILE_GROUP;
    if (s.mode == 0) return (int64_t)h * s.ntn;
    if (s.mode == 2) return stair_tiles_before_row(r0 + h, s) - stair_tiles_before_row(r0, s);
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
    {  // closed-form prefix count + bisection over the strips (782 tile rows at N = 100 000); a linear scan over the strips
       // cost the staircase mode 15 % of the K = 1024 update (profiles/r02_sharded_world1.log)
        const int ns = (s.ntm + TILE_GROUP - 1) / TILE_GROUP;
        int lo = 0, hi = ns - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            const int64_t before = s.mode == 2 ? stair_tiles_before_row(mid * TILE_GROUP, s) : lower_tiles_before_row(mid * TILE_GROUP, s);
            if (before <= t) lo = mid;
            else hi = mid - 1;
        }
        st = lo;
        t -= s.mode == 2 ? stair_tiles_before_row(st * TILE_GROUP, s) : lower_tiles_before_row(st * TILE_GROUP, s);
    }
    const int r0 = st * TILE_GROUP;
    const int h = (s.ntm - r0 < TILE_GROUP) ? (s.ntm - r0) : TILE_GROUP;
    if (s.mode == 0) {
        *tj = (int)(t / h);
        *ti = r0 + (int)(t % h);
        return;
    }
    if (s.mode == 2) {
        // column-major inside the strip, like the lower modes (GROUP consecutive tiles share one B panel): the columns
        // every row of the strip keeps first, then the staircase columns (cmax is non-decreasing down a strip)
        const int nfull = (r0 >= s.rc ? s.ntn - 1 : stair_cmax(s, r0)) + 1;
        if (t < (int64_t)h * nfull) {
            *tj = (int)(t / h);
            *ti = r0 + (int)(t % h);
            return;
        }
        int64_t q = t - (int64_t)h * nfull;
        int first = 1;  // first row of the strip that keeps column c
        for (int c = nfull;; ++c) {
            while (first < h && r0 + first < s.rc && stair_cmax(s, r0 + first) < c) ++first;
            const int n = h - first;
            if (q < n) {
                *tj = c;
                *ti = r0 + first + (int)q;
                return;
            }
            q -= n;
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
