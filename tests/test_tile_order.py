"""Host logic (no GPU): the tile enumeration the persistent trailing-update kernel pulls from
covers every output tile exactly once (compiled from the same header with g++)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <set>
#include <utility>
#include "tile_order.h"
using gpmi::TileShape;
static int check(TileShape s) {
    gpmi::stair_finalize(s);
    std::set<std::pair<int,int>> want, got;
    for (int i = 0; i < s.ntm; ++i)
        for (int j = 0; j < s.ntn; ++j) {
            bool ok = true;
            if (s.mode == 1) ok = j <= i + s.g0;
            if (s.mode == 3) ok = j <= 2 * (i + s.g0) + 1;
            if (s.mode == 2 && i < s.nstair) ok = j <= s.tpb * (s.g0 + (i / s.tpb) * s.G) + (i % s.tpb);
            if (ok) want.insert({i, j});
        }
    int64_t n = gpmi::tile_count(s);
    if (n != (int64_t)want.size()) return 1;
    for (int64_t t = 0; t < n; ++t) { int ti, tj; gpmi::tile_decode(t, s, &ti, &tj); got.insert({ti, tj}); }
    return got != want;
}
int main() {
    int bad = 0, cases = 0;
    for (int mode = 0; mode <= 1; ++mode)
        for (int ntm = 1; ntm <= 41; ++ntm)
            for (int ntn = 1; ntn <= (mode ? ntm : 41); ++ntn) { ++cases; bad += check(TileShape{ntm, ntn, mode, 0, 1, 0}); }
    // lower mode with a row offset (the look-ahead update: the block starts `off` tile-rows below the square's top)
    for (int off = 1; off <= 3; ++off)
        for (int ntm = 1; ntm <= 30; ++ntm)
            for (int ntn = 1; ntn <= ntm + off; ++ntn) { ++cases; bad += check(TileShape{ntm, ntn, 1, off, 1, 0}); }
    // lower mode with half-width column tiles (128 x 64 output tiles), with and without the row offset
    for (int off = 0; off <= 2; ++off)
        for (int ntm = 1; ntm <= 30; ++ntm)
            for (int ntn = 1; ntn <= 2 * (ntm + off); ++ntn) { ++cases; bad += check(TileShape{ntm, ntn, 3, off, 1, 0}); }
    // the same enumeration at twice the scale is the 256 x 128 tiling of update256.hip: the look-ahead updates of an N = 50 000
    // factorisation (row offset = next super-panel width / 256: 8, 4, 2; the last tile-row holds the carried right-hand side and
    // is capped by ntn), and the serial-order update (offset 0)
    for (int off : {0, 2, 4, 8})
        for (int rows : {24576, 47360, 50048 - 2048})
            for (int extra : {0, 8}) {
                const int M = rows + extra, N = rows + 256 * off;  // M rows below the next diagonal block, N columns from its first column
                ++cases; bad += check(TileShape{(M + 255) / 256, (N + 127) / 128, 3, off, 1, 0});
            }
    // staircase (row-block-cyclic shards): G ranks, first owned block g0, carried rows past the staircase
    for (int G = 1; G <= 4; ++G)
        for (int g0 = 0; g0 < G + 2; ++g0)
            for (int nblk = 0; nblk <= 9; ++nblk)
                for (int extra = 0; extra <= 2; ++extra) {
                    int ntm = 2 * nblk + extra; if (ntm == 0) continue;
                    int ntn = 2 * (g0 + (nblk ? (nblk - 1) * G : 0)) + 2 + 3;   // wider than the last stair
                    for (int cut = 0; cut <= 4; cut += 2) { ++cases; bad += check(TileShape{ntm, ntn - cut > 0 ? ntn - cut : 1, 2, g0, G, 2 * nblk}); }
                }
    // the same with 4 / 8 tiles per distributed block (the super-panel blocks of the two-level sharded factorisation)
    for (int tpb = 4; tpb <= 8; tpb += 4)
        for (int G = 1; G <= 3; ++G)
            for (int g0 = 0; g0 < G + 1; ++g0)
                for (int nblk = 0; nblk <= 5; ++nblk)
                    for (int extra = 0; extra <= 1; ++extra) {
                        int ntm = tpb * nblk + extra; if (ntm == 0) continue;
                        int ntn = tpb * (g0 + (nblk ? (nblk - 1) * G : 0)) + tpb + 3;
                        for (int cut = 0; cut <= 6; cut += 3) { ++cases; bad += check(TileShape{ntm, ntn - cut > 0 ? ntn - cut : 1, 2, g0, G, tpb * nblk, tpb}); }
                    }
    // the sizes of a real shard: 8 tiles per block, up to 25 local blocks, 8 ranks (N = 200 000 over 8 GPUs)
    for (int G = 1; G <= 8; G += 7)
        for (int g0 = 0; g0 <= G; g0 += G)
            for (int nblk = 20; nblk <= 25; nblk += 5) {
                int ntm = 8 * nblk + 1;
                int ntn = 8 * (g0 + (nblk - 1) * G) + 8 + 5;
                ++cases; bad += check(TileShape{ntm, ntn, 2, g0, G, 8 * nblk, 8});
                ++cases; bad += check(TileShape{ntm, ntn - 40, 2, g0, G, 8 * nblk, 8});
            }
    // staircase locality: 8 consecutive tiles at the start of an interior strip share one column
    {
        int a0, b0, a7, b7;
        TileShape st2{64, 200, 2, 1, 2, 64, 8};
        gpmi::stair_finalize(st2);
        int64_t b2 = 0; for (int st = 0; st < 2; ++st) b2 += gpmi::strip_count(st, st2);
        gpmi::tile_decode(b2, st2, &a0, &b0); gpmi::tile_decode(b2 + 7, st2, &a7, &b7);
        if (!(b0 == 0 && b7 == 0 && a0 == 16 && a7 == 23)) ++bad;
    }
    // mode 1 == mode 2 with (g0, G, nstair) = (0, 1, ntm): same tile SET
    for (int ntm = 1; ntm <= 20; ++ntm) { ++cases; TileShape m2{ntm, ntm, 2, 0, 1, ntm}; gpmi::stair_finalize(m2); bad += gpmi::tile_count(TileShape{ntm, ntm, 1, 0, 1, 0}) != gpmi::tile_count(m2); }
    // locality: 8 consecutive tiles of a full interior strip share one column (mode 1)
    int ti0, tj0, ti7, tj7;
    TileShape s{64, 64, 1, 0, 1, 0};
    int64_t base = 0; for (int st = 0; st < 3; ++st) base += gpmi::strip_count(st, s);
    gpmi::tile_decode(base, s, &ti0, &tj0); gpmi::tile_decode(base + 7, s, &ti7, &tj7);
    if (!(tj0 == 0 && tj7 == 0 && ti0 == 24 && ti7 == 31)) ++bad;
    std::printf("%d cases, %d bad\n", cases, bad);
    return bad != 0;
}
"""


def test_tile_enumeration_is_a_bijection():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(td, "t")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "gaussianprocesses.jl_amd", "csrc"), src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert " 0 bad" in out.stdout
