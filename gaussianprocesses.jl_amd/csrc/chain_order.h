// chain_order.h — the task list of the persistent chain kernel (chain.hip), as host + device code so that the CPU suite can check the one
// property the kernel's liveness rests on (tests/test_chain_order.py): tasks are numbered in a TOPOLOGICAL order of the tile dependencies,
// so a workgroup that holds task t only ever waits for tasks with smaller numbers.
//
// A W x W block is nb x nb tiles of 64 x 64.  Step c (c = 0 .. nb-1) lists the L tiles of column c from the diagonal down, then — when the
// explicit inverse is built — the X tiles of row c from the left:
//     L(c, c), L(c+1, c), ..., L(nb-1, c),   X(c, 0), ..., X(c, c-1)
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

GPMI_HD ChainTask chain_decode(int t, int nb, bool inverse) {
    int c, q;
    if (inverse) {  // nb tasks per step: nb - c of L, c of X
        c = t / nb;
        q = t - c * nb;
    } else {
        c = 0;
        q = t;
        while (q >= nb - c) {
            q -= nb - c;
            ++c;
        }
    }
    ChainTask k;
    k.is_x = q >= nb - c;
    k.i = k.is_x ? c : c + q;
    k.j = k.is_x ? q - (nb - c) : c;
    return k;
}

// the number of a task (the inverse of chain_decode): what the test asks of every dependency
GPMI_HD int chain_index(ChainTask k, int nb, bool inverse) {
    if (inverse) return k.is_x ? k.i * nb + (nb - k.i) + k.j : k.j * nb + (k.i - k.j);
    int t = 0;
    for (int c = 0; c < k.j; ++c) t += nb - c;
    return t + (k.i - k.j);
}

}  // namespace gpmi
