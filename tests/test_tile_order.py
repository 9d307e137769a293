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
int main() {
    int bad = 0, cases = 0;
    for (int lower = 0; lower <= 1; ++lower)
        for (int ntm = 1; ntm <= 41; ++ntm)
            for (int ntn = 1; ntn <= (lower ? ntm : 41); ++ntn) {
                ++cases;
                std::set<std::pair<int,int>> want, got;
                for (int i = 0; i < ntm; ++i) for (int j = 0; j < ntn; ++j) if (!lower || j <= i) want.insert({i, j});
                int64_t n = gpmi::tile_count(ntm, ntn, lower);
                if (n != (int64_t)want.size()) { ++bad; continue; }
                for (int64_t t = 0; t < n; ++t) { int ti, tj; gpmi::tile_decode(t, ntm, ntn, lower, &ti, &tj); got.insert({ti, tj}); }
                if (got != want) ++bad;
            }
    // locality: 8 consecutive tiles of a full interior strip share one column
    int ti0, tj0, ti7, tj7;
    int64_t base = 0; for (int s = 0; s < 3; ++s) base += gpmi::strip_count(s, 64, 64, 1);
    gpmi::tile_decode(base, 64, 64, 1, &ti0, &tj0); gpmi::tile_decode(base + 7, 64, 64, 1, &ti7, &tj7);
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
