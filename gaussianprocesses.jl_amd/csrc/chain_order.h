// chain_order.h — the task list of the persistent chain kernel (chain.hip), as host + device code so that the CPU suite can check the one
// property the kernel's liveness rests on (tests/test_chain_order.py): tasks are numbered in a TOPOLOGICAL order of the tile dependencies,
// so a workgroup that holds task t only ever waits for tasks with smaller numbers.
//
// A W x W block is nb x nb tiles of 64 x 64.  The list starts with L(0, 0); step c (c = 0 .. nb-1) then lists the tile below the diagonal
// of column c and AT ONCE the next diagonal tile, then the rest of column c, then — when the explicit inverse is built — the X tiles of
// row c from the left:
//     L(c+1, c), L(c+1, c+1),   L(c+2, c), ..., L(nb-1, c),   X(c, 0), ..., X(c, c-1)
// The diagonal tiles are the critical path of the block (each carries the 64-column potf2, ~22 us on one wavefront): handed out ahead of
// the bulk of the previous column, L(c+1, c+1) has multiplied all its finished slabs by the time L(c+1, c) arrives, and the nb - 2 tasks
// listed behind it need nothing newer than L(c, c) — work for the other workgroups while the potf2 of L(c+1, c+1) runs.  (With the diagonal
// tile listed after the whole previous step — round 5's first order — the in-kernel timeline, tools/chain_trace.py, showed it picked up
// ~30 us late at every step: 63 us per step of which 22 potf2.)
// Dependencies (chain.hip):  L(i, c) needs L(i, k) and L(c, k) for k < c, and L(c, c) when i > c;
//                            X(i, j) needs L(i, k) for j <= k < i, X(k, j) for j < k < i, L(j, j) (X_jj = Linv_j) and L(i, i) (Linv_i).
#pragma once
#include "tile_order.h"  // GPMI_HD

namespace gpmi {

struct ChainTask {
    int is_x;  // 0: the tile L(i, j) of the factor (i >= j), 1: the tile X(i, j) of the inverse (i > j)
    int i, j;
};

GPMI_HD int chain_ntasks(int nb, bool inverse) { return inverse ? nb * nb : nb * (nb + 1) / 2; }

// tasks of step c behind L(0, 0): column c from row c+1 down plus the next diagonal tile (nb - c of L when c < nb - 1), then c of X
GPMI_HD int chain_step_l(int c, int nb) { return c < nb - 1 ? nb - c : 0; }

GPMI_HD ChainTask chain_decode(int t, int nb, bool inverse) {
    ChainTask k;
    if (t == 0) {
        k.is_x = 0; k.i = 0; k.j = 0;
        return k;
    }
    int c, q = t - 1;
    if (inverse) {  // nb tasks per step (the last step: nb - 1, all of X)
        c = q / nb;
        q -= c * nb;
    } else {
        c = 0;
        while (q >= chain_step_l(c, nb)) {
            q -= chain_step_l(c, nb);
            ++c;
        }
    }
    const int nl = chain_step_l(c, nb);
    k.is_x = q >= nl;
    if (k.is_x) {
        k.i = c;
        k.j = q - nl;
    } else if (q == 1) {  // the next diagonal tile
        k.i = c + 1;
        k.j = c + 1;
    } else {              // q = 0: L(c+1, c);  q >= 2: L(c+q, c)
        k.i = q == 0 ? c + 1 : c + q;
        k.j = c;
    }
    return k;
}

// the number of a task (the inverse of chain_decode): what the test asks of every dependency
GPMI_HD int chain_index(ChainTask k, int nb, bool inverse) {
    if (!k.is_x && k.i == 0) return 0;
    const int c = k.is_x ? k.i : (k.i == k.j ? k.i - 1 : k.j);  // the step that lists the task
    int base = 1;
    if (inverse)
        base += c * nb;
    else
        for (int s = 0; s < c; ++s) base += chain_step_l(s, nb);
    if (k.is_x) return base + chain_step_l(c, nb) + k.j;
    if (k.i == k.j) return base + 1;
    return base + (k.i == k.j + 1 ? 0 : k.i - k.j);
}

}  // namespace gpmi
