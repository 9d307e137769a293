// chol.h — the dense factorisation / whitening sequences shared by the exact path (api.hip) and the FITC path
// (fitc.hip).  Header-only templates over the launchers of common.h; everything enqueues on ctx->stream.
#pragma once
#include <algorithm>

#include "common.h"
#include "tile_order.h"

namespace gpmi {

// Solve the block-column [k0, k0+nbk) of a row matrix R (Mr rows) against the already factored diagonal block
// of A (64-column steps: left-looking update + multiplication by the stored inverse, one launch each), then push
// the update into R's remaining columns:
//   R[:, k0:kend] <- R[:, k0:kend] * L_kk^-T ;  R[:, kend:npad] -= R[:, k0:kend] * A[kend:npad, k0:kend]'
template <typename T>
inline void rows_block_solve(gpmi_ctx* c, const T* A, int64_t ld, const T* linv, int64_t npad, T* R, int64_t ldr,
                             int64_t Mr, int64_t k0, int64_t nbk) {
    const int64_t kend = k0 + nbk;
    for (int64_t j0 = k0; j0 < kend; j0 += IB)
        launch_rows64<T>(c, R + k0, ldr, Mr, (int)(j0 - k0), A + j0 * ld + k0, ld, linv + (j0 / IB) * IB * IB, 0, nullptr);
    if (kend < npad)
        launch_gemm_nt<T>(c, R + kend, ldr, R + k0, ldr, A + kend * ld + k0, ld, Mr, npad - kend, nbk, 0, nullptr);
}

// run launches on another stream of the context (the launchers read ctx->stream / ctx->num_cus)
struct StreamScope {
    gpmi_ctx* c;
    hipStream_t s0;
    int cus0;
    StreamScope(gpmi_ctx* ctx, hipStream_t s, int cus) : c(ctx), s0(ctx->stream), cus0(ctx->num_cus) {
        c->stream = s;
        c->num_cus = cus;
    }
    ~StreamScope() {
        c->stream = s0;
        c->num_cus = cus0;
    }
};
// cross-stream dependency events: handed out in order, recycled by la_reset() at the start of every API call (each
// call ends with a stream synchronisation, so none of them is pending then)
inline void la_reset(gpmi_ctx* c) { c->la_next = 0; }
inline hipEvent_t la_event(gpmi_ctx* c) {
    const size_t i = c->la_next++;
    while (c->la_events.size() <= i) {
        hipEvent_t e;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        c->la_events.push_back(e);
    }
    return c->la_events[i];
}

// The same whitening through the explicit NB x NB inverses, out of place: V <- R L^-T with one product per NB columns
//   V[:, k0:kend] = R[:, k0:kend] * Linv_k' ;  R[:, kend:npad] -= V[:, k0:kend] * A[kend:npad, k0:kend]'
// (R is consumed).  Out of place because the two 128-column tiles of a block read each other's input columns.
// rows_upto(kend) gives the number of leading rows that can be non-zero up to column kend (identity right-hand sides).
// (A look-ahead of the next block's solve on the side stream, as in cholesky_lower, bought 1.6 ms per predict while the
//  solve was a 49 us launch of 128 x 128 tiles; with 128 x 64 tiles it is 28 us and the look-ahead measured neutral: removed.)
template <typename T, typename F>
inline void whiten_rows_inv(gpmi_ctx* c, const T* A, int64_t ld, const T* linv256, int64_t npad, T* R, int64_t ldr, T* V,
                            int64_t ldv, F rows_upto) {
    const TileShape rect{0, 0, 0, 0, 1, 0};
    for (int64_t k0 = 0; k0 < npad; k0 += NB) {
        const int64_t nbk = std::min<int64_t>(NB, npad - k0), k1 = k0 + nbk;
        const int64_t Mr = rows_upto(k1);
        launch_gemm_shape<T>(c, V + k0, ldv, R + k0, ldr, linv256 + (k0 / NB) * NB * NB, NB, Mr, nbk, nbk, rect, nullptr,
                             GEMM_OVERWRITE);
        if (k1 < npad) launch_gemm_nt<T>(c, R + k1, ldr, V + k0, ldv, A + k1 * ld + k0, ld, Mr, npad - k1, nbk, 0, nullptr);
    }
}

template <typename T>
inline void whiten_rows(gpmi_ctx* c, const T* A, int64_t ld, const T* linv, int64_t npad, T* R, int64_t ldr, int64_t Mr) {
    for (int64_t k0 = 0; k0 < npad; k0 += NB)
        rows_block_solve<T>(c, A, ld, linv, npad, R, ldr, Mr, k0, std::min<int64_t>(NB, npad - k0));
}

// Blocked right-looking Cholesky of the row-major lower triangle of A (npad x npad), carrying
// `extra` rows below it (row npad = y) through the panel solves and trailing updates, so that
// on exit row npad holds z = L^-1 y (the forward half of cK \ y, GPE.jl:208).
// one NB-wide panel: per 64 columns diag64 (factor + invert the diagonal block) and rows64 over every row below
// (left-looking update inside the panel + TRSM as a product with the inverse + diagonal-block updates)
template <typename T>
inline void factor_panel_diag(gpmi_ctx* c, T* A, int64_t ld, T* linv, T* invdiag, int64_t k0, int64_t nbk, int* d_info);
template <typename T>
inline void factor_panel_below(gpmi_ctx* c, T* A, int64_t ld, const T* linv, int64_t k0, int64_t nbk, int64_t Mtot,
                               const int* d_info);
template <typename T>
inline void factor_panel(gpmi_ctx* c, T* A, int64_t ld, T* linv, T* invdiag, int64_t k0, int64_t nbk, int64_t Mtot, int* d_info) {
    factor_panel_diag<T>(c, A, ld, linv, invdiag, k0, nbk, d_info);
    factor_panel_below<T>(c, A, ld, linv, k0, nbk, Mtot, d_info);
}
// the same panel in two parts: the serial chain on the nbk x nbk diagonal block (one to three workgroups per launch) ...
template <typename T>
inline void factor_panel_diag(gpmi_ctx* c, T* A, int64_t ld, T* linv, T* invdiag, int64_t k0, int64_t nbk, int* d_info) {
    const int64_t kend = k0 + nbk;
    for (int64_t j0 = k0; j0 < kend; j0 += IB) {
        T* linv_j = linv + (j0 / IB) * IB * IB;
        launch_diag64<T>(c, A + j0 * ld + j0, ld, linv_j, invdiag + j0, d_info, j0);
        const int64_t r0 = j0 + IB;
        if (r0 < kend)
            launch_rows64<T>(c, A + r0 * ld + k0, ld, kend - r0, (int)(j0 - k0), A + j0 * ld + k0, ld, linv_j, kend - r0, d_info);
    }
}
// ... and the rows below it (chip-wide), which only need the finished diagonal block and its inverses
template <typename T>
inline void factor_panel_below(gpmi_ctx* c, T* A, int64_t ld, const T* linv, int64_t k0, int64_t nbk, int64_t Mtot,
                               const int* d_info) {
    const int64_t kend = k0 + nbk;
    // one fused launch when every workgroup gets a CU to itself (rows256 keeps 64 x 256 of X in registers: one workgroup
    // per CU; with more row blocks than CUs it would run in two rounds and lose to the four rows64 launches);
    // the refinement step lives in rows64 only
    if (!c->refine_solves && (Mtot - kend + IB - 1) / IB <= c->num_cus) {
        launch_rows256<T>(c, A + kend * ld + k0, ld, Mtot - kend, (int)(nbk / IB), A + k0 * ld + k0, ld, linv + (k0 / IB) * IB * IB,
                          d_info);
        return;
    }
    for (int64_t j0 = k0; j0 < kend; j0 += IB)
        launch_rows64<T>(c, A + kend * ld + k0, ld, Mtot - kend, (int)(j0 - k0), A + j0 * ld + k0, ld,
                         linv + (j0 / IB) * IB * IB, 0, d_info);
}

// blocked right-looking Cholesky, lower, in place; rows npad .. npad+extra-1 are carried along (forward solve for free).
//
// Look-ahead: the serial part of panel k+1 — the diag64 / rows64 chain on its 256 x 256 diagonal block, one to three
// workgroups per launch, ~150 us — runs on a second (high-priority) stream UNDER the trailing update by panel k.
// For that the update is split by tiles, not by columns: the three 128 x 128 tiles of the next diagonal block go first
// on the side stream (a 3-workgroup launch), everything else is one persistent launch on the main stream whose grid
// leaves `lookahead_slots` workgroup slots of the chip free, which is where the side stream's small kernels land.
// (Two measured dead ends, tools/cumask_probe.hip + profiles/: splitting off the next panel's 256 COLUMNS costs a
// single-round GEMM launch as long as the chain it hides; CU-masked streams work but cost the GEMM 4 % for 8 CUs.)
template <typename T>
inline void cholesky_lower(gpmi_ctx* c, T* A, int64_t ld, T* linv, T* invdiag, int64_t npad, int64_t extra, int* d_info) {
    const int64_t Mtot = npad + extra;
    if (c->lookahead_slots <= 0 || !c->side_stream || npad <= 4 * NB) {
        for (int64_t k0 = 0; k0 < npad; k0 += NB) {
            const int64_t nbk = std::min<int64_t>(NB, npad - k0);
            const int64_t kend = k0 + nbk;
            factor_panel<T>(c, A, ld, linv, invdiag, k0, nbk, Mtot, d_info);
            const int64_t M = Mtot - kend;
            if (M > 0 && kend < npad)  // trailing update (SYRK shape, K = nbk): the MFMA-bound bulk
                launch_gemm_nt<T>(c, A + kend * ld + kend, ld, A + kend * ld + k0, ld, A + kend * ld + k0, ld, M, npad - kend,
                                  nbk, 1, d_info);
        }
        return;
    }
    hipStream_t main_s = c->stream, side = c->side_stream;
    bool panel_done = false;  // is panel k0 already factored (by the look-ahead of the previous step)?
    for (int64_t k0 = 0; k0 < npad; k0 += NB) {
        const int64_t nbk = std::min<int64_t>(NB, npad - k0);
        const int64_t k1 = k0 + nbk;
        if (!panel_done) factor_panel<T>(c, A, ld, linv, invdiag, k0, nbk, Mtot, d_info);
        panel_done = false;
        if (k1 >= npad) break;
        const int64_t nb1 = std::min<int64_t>(NB, npad - k1);
        const int64_t k2 = k1 + nb1;
        // the chain takes ~0.4 ms beside the update (contended CUs): look ahead only while the update is longer
        if (npad - k2 < c->lookahead_min_trailing) {
            launch_gemm_nt<T>(c, A + k1 * ld + k1, ld, A + k1 * ld + k0, ld, A + k1 * ld + k0, ld, Mtot - k1, npad - k1, nbk, 1,
                              d_info);
            continue;
        }
        hipEvent_t eb = la_event(c);  // panel k complete
        (void)hipEventRecord(eb, main_s);
        (void)hipStreamWaitEvent(side, eb, 0);
        {
            StreamScope sc(c, side, c->num_cus);
            // the next diagonal block's own tiles, then its factorisation chain
            launch_gemm_shape<T>(c, A + k1 * ld + k1, ld, A + k1 * ld + k0, ld, A + k1 * ld + k0, ld, nb1, nb1, nbk,
                                 TileShape{0, 0, 1, 0, 1, 0}, d_info, GEMM_AUX);
            factor_panel_diag<T>(c, A, ld, linv, invdiag, k1, nb1, d_info);
        }
        hipEvent_t ec = la_event(c);
        (void)hipEventRecord(ec, side);
        // everything below the next diagonal block: rows k2.., columns k1.. up to each row tile's diagonal tile
        // (lower mode with offset: row tile ti keeps column tiles <= ti + 2)
        c->gemm_reserve = c->lookahead_slots;
        launch_gemm_shape<T>(c, A + k2 * ld + k1, ld, A + k2 * ld + k0, ld, A + k1 * ld + k0, ld, Mtot - k2, npad - k1, nbk,
                             TileShape{0, 0, 1, 2, 1, 0}, d_info, 0);
        c->gemm_reserve = 0;
        (void)hipStreamWaitEvent(main_s, ec, 0);
        factor_panel_below<T>(c, A, ld, linv, k1, nb1, Mtot, d_info);
        panel_done = true;
    }
}


}  // namespace gpmi
