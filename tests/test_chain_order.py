"""Host logic (no GPU): the task list of the persistent chain kernel (csrc/chain_order.h, compiled with g++ from the header the kernel
includes).  The kernel hands tasks out by ONE counter and lets a workgroup wait only on the flags of the tiles its task reads; it makes
progress with ANY number of resident workgroups — no grid barrier, no co-residency assumption — if and only if every task's dependencies
carry SMALLER task numbers (then the lowest unfinished task never waits on an unfinished one).  Checked for every block size the library
uses (nb = 1 .. 32 tiles per side = up to 2048 columns), with and without the explicit inverse: the numbering covers every L tile (i >= j)
and, with the inverse, every X tile (i > j) exactly once, chain_index inverts chain_decode, and every dependency is numbered lower."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <set>
#include <tuple>
#include "chain_order.h"
using gpmi::ChainTask;
static int check(int nb, bool inv) {
    const int n = gpmi::chain_ntasks(nb, inv);
    std::set<std::tuple<int,int,int>> seen;
    for (int t = 0; t < n; ++t) {
        const ChainTask k = gpmi::chain_decode(t, nb, inv);
        if (k.is_x ? !(inv && k.i > k.j && k.i < nb && k.j >= 0) : !(k.i >= k.j && k.i < nb && k.j >= 0)) return 1;   // a valid tile
        if (!seen.insert({k.is_x, k.i, k.j}).second) return 2;                                                         // exactly once
        if (gpmi::chain_index(k, nb, inv) != t) return 3;                                                              // index inverts decode
        auto before = [&](int is_x, int i, int j) { return gpmi::chain_index(ChainTask{is_x, i, j}, nb, inv) < t; };
        if (!k.is_x) {
            const int i = k.i, c = k.j;
            for (int kk = 0; kk < c; ++kk)
                if (!before(0, i, kk) || !before(0, c, kk)) return 4;      // L(i, k), L(c, k), k < c
            if (i > c && !before(0, c, c)) return 5;                        // Linv_c
        } else {
            const int i = k.i, j = k.j;
            for (int kk = j; kk < i; ++kk)
                if (!before(0, i, kk)) return 6;                            // L(i, k), j <= k < i
            for (int kk = j + 1; kk < i; ++kk)
                if (!before(1, kk, j)) return 7;                            // X(k, j), j < k < i
            if (!before(0, j, j) || !before(0, i, i)) return 8;             // X_jj = Linv_j, Linv_i
        }
    }
    const int want = inv ? nb * (nb + 1) / 2 + nb * (nb - 1) / 2 : nb * (nb + 1) / 2;
    return (int)seen.size() == want && n == want ? 0 : 9;
}
int main() {
    int cases = 0;
    for (int nb = 1; nb <= 32; ++nb)
        for (int inv = 0; inv <= 1; ++inv) {
            const int rc = check(nb, inv != 0);
            if (rc) { std::printf("FAIL nb=%d inv=%d rc=%d\n", nb, inv, rc); return 1; }
            ++cases;
        }
    std::printf("OK %d\n", cases);
    return 0;
}
"""


def test_chain_task_order_is_topological_and_complete():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.cpp")
        exe = os.path.join(td, "t")
        with open(src, "w") as fh:
            fh.write(SRC)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "gaussianprocesses.jl_amd", "csrc"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode()
    assert out.strip() == "OK 64", out


DUMP = r"""
#include <cstdio>
#include "chain_order.h"
int main() {
    const int sizes[4] = {1, 2, 5, 16};
    for (int nb : sizes) {
        const int n = gpmi::chain_ntasks(nb, true);
        for (int t = 0; t < n; ++t) {
            const gpmi::ChainTask k = gpmi::chain_decode(t, nb, true);
            std::printf("%d %c %d %d\n", nb, k.is_x ? 'X' : 'L', k.i, k.j);
        }
    }
    return 0;
}
"""


def test_queue_model_and_trace_parser_follow_the_kernels_order():
    """tools/chain_sim.py (the queue model the LABBOOK's order comparison rests on) must model the order the kernel really uses, and
    tools/chain_trace.py must parse what the trace build writes (a synthetic file in that format)."""
    import importlib.util
    import sys

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        argv, sys.argv = sys.argv, [name]
        try:
            spec.loader.exec_module(mod)
        finally:
            sys.argv = argv
        return mod

    sim = load("chain_sim")
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "d.cpp"), os.path.join(td, "d")
        with open(src, "w") as fh:
            fh.write(DUMP)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "gaussianprocesses.jl_amd", "csrc"), src, "-o", exe])
        lines = subprocess.check_output([exe]).decode().split("\n")
    got = {}
    for ln in lines:
        if ln:
            nb, kind, i, j = ln.split()
            got.setdefault(int(nb), []).append((kind, int(i), int(j)))
    for nb, order in got.items():
        assert order == sim.order_diag_ahead(nb), nb
        assert sim.check_topological(order, nb)
        assert not sim.check_topological(order[::-1], nb) or nb == 1
    # the model itself: more workgroups never make a block slower, the diagonal-ahead list beats the column-by-column one
    spans = [sim.simulate(sim.order_diag_ahead(16), 16, w) for w in (1, 2, 4, 8, 16, 32)]
    assert all(a >= b - 1e-9 for a, b in zip(spans, spans[1:])), spans
    assert sim.simulate(sim.order_diag_ahead(16), 16, 8) < sim.simulate(sim.order_round5_first(16), 16, 8)

    tr = load("chain_trace")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "t.txt")
        with open(path, "w") as fh:
            fh.write("launch nb 2 wgs 2 ld 128 inverse 1 beside 0 elem 8 potf2 900 300 140 350\n")
            t = 1000
            for q, (kind, i, j) in enumerate(sim.order_diag_ahead(2)):
                x = 1 if kind == "X" else 0
                fh.write(" ".join(map(str, [q, t, t + 100, t + 300, t + 500, t + 600, 50, ((q % 2) << 8) | 3, (x << 16) | (i << 8) | j])) + "\n")
                t += 400
        (L,) = tr.parse(path)
        assert (L["nb"], L["wgs"], L["inverse"], len(L["tasks"])) == (2, 2, 1, 4) and L["potf2"] == [900, 300, 140, 350]
        assert [(k["is_x"], k["i"], k["j"]) for k in L["tasks"]] == [(0, 0, 0), (0, 1, 0), (0, 1, 1), (1, 1, 0)]
        import io
        buf = io.StringIO()
        tr.report(L, out=buf)
        assert "span 18.0 us" in buf.getvalue() and "diagonal chain, 1 steps" in buf.getvalue(), buf.getvalue()
