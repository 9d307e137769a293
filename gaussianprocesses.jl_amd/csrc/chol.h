// chol.h — the dense factorisation / whitening sequences shared by the exact path (api.hip) and the FITC path
// (fitc.hip).  Header-only templates over the launchers of common.h; everything enqueues on ctx->stream.
#pragma once
#include <algorithm>

#include "common.h"
#include "tile_order.h"

namespace gpmi {

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

// A column range of the factor whose diagonal block has a stored explicit inverse (lw != nullptr: w x w, leading dimension
// w + 64, kept from the factorisation's super-panel, cholesky_lower) or not (solved through the NB x NB inverses).
template <typename T>
struct WhitenSeg {
    int64_t ks, ke;
    const T* lw;
};

// The same whitening through explicit inverses, out of place: V <- R L^-T, in two levels like cholesky_lower.
//   * a segment with a stored super-block inverse is TWO launches: V[:, ks:ke] = R[:, ks:ke] * LW' (the K loop of a column
//     tile ends at its last column: LW is lower triangular), then R[:, ke:npad] -= V[:, ks:ke] * A[ke:npad, ks:ke]'  (K = w);
//   * elsewhere NB-blocks are grouped into super-blocks of up to WS = 1024 columns; inside one the solve is left-looking
//     (block b first receives the products of the super-block's earlier blocks, K = 256 b), and the columns beyond it get
//     ONE update with K = WS when it is complete:
//       R[:, k0:k1] -= V[:, ks:k0] * A[k0:k1, ks:k0]' ;  V[:, k0:k1] = R[:, k0:k1] * Linv_k'      (k0 in the super-block [ks, ke))
//       R[:, ke:npad] -= V[:, ks:ke] * A[ke:npad, ks:ke]'
// (R is consumed).  Out of place because the column tiles of a block read each other's input columns.
// rows_upto(kend) gives the number of leading rows that can be non-zero up to column kend (identity right-hand sides);
// a segment works on rows_upto(its last column) rows throughout, so that V holds zeros — not stale data — wherever
// the K = w product reads it.  `segs` = nullptr: no stored inverses (one plain range).
// (A look-ahead of the next block's solve on the side stream, as in cholesky_lower, bought 1.6 ms per predict while the
//  solve was a 49 us launch of 128 x 128 tiles; with 128 x 64 tiles it is 28 us and the look-ahead measured neutral: removed.)
template <typename T, typename F>
inline void whiten_rows_inv(gpmi_ctx* c, const T* A, int64_t ld, const T* linv256, int64_t npad, T* R, int64_t ldr, T* V,
                            int64_t ldv, F rows_upto, const std::vector<WhitenSeg<T>>* segs = nullptr) {
    const TileShape rect{0, 0, 0, 0, 1, 0};
    const int64_t WS = std::max<int64_t>(NB, c->whiten_super);
    auto plain_range = [&](int64_t r0, int64_t r1) {
        for (int64_t ks = r0; ks < r1; ks += WS) {
            const int64_t ke = std::min<int64_t>(ks + WS, r1);
            const int64_t Mr = rows_upto(ke);
            for (int64_t k0 = ks; k0 < ke; k0 += NB) {
                const int64_t nbk = std::min<int64_t>(NB, ke - k0);
                if (k0 > ks)
                    launch_gemm_nt<T>(c, R + k0, ldr, V + ks, ldv, A + k0 * ld + ks, ld, Mr, nbk, k0 - ks, 0, nullptr);
                launch_gemm_shape<T>(c, V + k0, ldv, R + k0, ldr, linv256 + (k0 / NB) * NB * NB, NB, Mr, nbk, nbk, rect, nullptr,
                                     GEMM_OVERWRITE);
            }
            if (ke < npad) launch_gemm_nt<T>(c, R + ke, ldr, V + ks, ldv, A + ke * ld + ks, ld, Mr, npad - ke, ke - ks, 0, nullptr);
        }
    };
    int64_t pos = 0;
    if (segs)
        for (const auto& sg : *segs) {
            if (!sg.lw || sg.ks < pos) continue;
            if (sg.ks > pos) plain_range(pos, sg.ks);
            const int64_t w = sg.ke - sg.ks, Mr = rows_upto(sg.ke);
            launch_gemm_shape<T>(c, V + sg.ks, ldv, R + sg.ks, ldr, sg.lw, w + IB, Mr, w, w, rect, nullptr, GEMM_OVERWRITE | GEMM_KEND_COL);
            if (sg.ke < npad)
                launch_gemm_nt<T>(c, R + sg.ke, ldr, V + sg.ks, ldv, A + sg.ke * ld + sg.ks, ld, Mr, npad - sg.ke, w, 0, nullptr);
            pos = sg.ke;
        }
    if (pos < npad) plain_range(pos, npad);
}

// In-place whitening through the 64 x 64 inverses (the form that carries the refinement step), in two levels like
// whiten_rows_inv: NB-blocks are grouped into super-blocks of WS = 1024 columns; inside one, block b first receives the
// products of the super-block's earlier blocks (K = 256 b), then its four rows64 steps; the columns beyond the super-block
// get ONE update with K = WS.  (FITC's W = Kfu Luu^-T, n = 1e6 rows: the K = 256 updates of the one-level form ran at
// ~47 TFLOP/s — C traffic — the K = 1024 ones at ~62.)
template <typename T>
inline void whiten_rows(gpmi_ctx* c, const T* A, int64_t ld, const T* linv, int64_t npad, T* R, int64_t ldr, int64_t Mr) {
    const int64_t WS = std::max<int64_t>(NB, c->whiten_super);
    for (int64_t ks = 0; ks < npad; ks += WS) {
        const int64_t ke = std::min<int64_t>(ks + WS, npad);
        for (int64_t k0 = ks; k0 < ke; k0 += NB) {
            const int64_t nbk = std::min<int64_t>(NB, ke - k0);
            if (k0 > ks) launch_gemm_nt<T>(c, R + k0, ldr, R + ks, ldr, A + k0 * ld + ks, ld, Mr, nbk, k0 - ks, 0, nullptr);
            for (int64_t j0 = k0; j0 < k0 + nbk; j0 += IB)
                launch_rows64<T>(c, R + k0, ldr, Mr, (int)(j0 - k0), A + j0 * ld + k0, ld, linv + (j0 / IB) * IB * IB, 0, nullptr);
        }
        if (ke < npad) launch_gemm_nt<T>(c, R + ke, ldr, R + ks, ldr, A + ke * ld + ks, ld, Mr, npad - ke, ke - ks, 0, nullptr);
    }
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
// ... and rows [r0, r1) below it (chip-wide), which only need the finished diagonal block and its inverses
template <typename T>
inline void factor_panel_rows(gpmi_ctx* c, T* A, int64_t ld, const T* linv, int64_t k0, int64_t nbk, int64_t r0, int64_t r1,
                              const int* d_info) {
    const int64_t kend = k0 + nbk, M = r1 - r0;
    if (M <= 0) return;
    // one fused launch when every workgroup gets a CU to itself (rows256 keeps 64 x 256 of X in registers: one workgroup
    // per CU; with more row blocks than CUs it would run in two rounds and lose to the four rows64 launches);
    // the refinement step lives in rows64 only
    // (and never beside the persistent update: a 512-register workgroup needs a whole CU and would wait for the update to end)
    if (!c->refine_solves && !c->beside_update && (M + IB - 1) / IB <= c->num_cus) {
        launch_rows256<T>(c, A + r0 * ld + k0, ld, M, (int)(nbk / IB), A + k0 * ld + k0, ld, linv + (k0 / IB) * IB * IB, d_info);
        return;
    }
    for (int64_t j0 = k0; j0 < kend; j0 += IB)
        launch_rows64<T>(c, A + r0 * ld + k0, ld, M, (int)(j0 - k0), A + j0 * ld + k0, ld, linv + (j0 / IB) * IB * IB, 0, d_info);
}
template <typename T>
inline void factor_panel_below(gpmi_ctx* c, T* A, int64_t ld, const T* linv, int64_t k0, int64_t nbk, int64_t Mtot,
                               const int* d_info) {
    factor_panel_rows<T>(c, A, ld, linv, k0, nbk, k0 + nbk, Mtot, d_info);
}

// Blocked Cholesky, lower, in place; rows npad .. npad+extra-1 are carried along (forward solve for free).
//
// TWO LEVELS (round 2).  Right-looking over SUPER-PANELS of W = 256 * 2^s columns (W from the remaining size, super_width()):
//   1. the W x W diagonal block is factored by the NB = 256 machinery (factor_diag_block: diag64 / rows64 chains and small
//      updates, all latency-bound, a few workgroups per launch);
//   2. the rows below it are solved against that block 256 columns at a time, left-looking (rows_below_super: block b first
//      receives the products of the super-panel's earlier blocks in ONE update with K = 256 b, then the product with the
//      stored inverses) — chip-wide launches;
//   3. the trailing matrix gets ONE update with K = W.
// Every trailing element is read and written once per W columns instead of once per 256: the update kernel's C traffic per
// flop (25 % of its time at K = 256, LABBOOK.md 3.2) drops by W / 256, and the bulk of the n^3 / 3 flops runs at the K = 1024
// rate of gemm_nt_kernel (63 instead of 47 TFLOP/s isolated).  The skinny products of step 2 are (W / M) of the flops.
//
// LOOK-AHEAD at the super-panel level: step 1 of the NEXT super-panel — every latency-bound launch of the factorisation —
// runs on a second, high-priority stream UNDER step 3 of the current one.  For that the update is split by tiles: the lower
// tiles of the next W' x W' diagonal block go first (their own short launch on the main stream), everything else is one
// persistent launch on the main stream (tile_order.h lower mode with a row offset) whose grid leaves `lookahead_slots`
// workgroup slots free, which is where the side stream's small kernels land.
// (Two measured dead ends, tools/cumask_probe.hip + profiles/: splitting off the next panel's COLUMNS costs a
// single-round GEMM launch as long as the chain it hides; CU-masked streams work but cost the GEMM 4 % for 8 CUs.)
// The main-stream launch(es) that a side-stream chain hides under.  With reserved compute units (common.h: upd_stream) they
// go to the CU-masked update stream — ordered after `after` (an event already recorded on the main stream) and joined back
// into the main stream — and the grid is sized for the unreserved CUs; otherwise (round 2) they stay on the main stream and
// leave `lookahead_slots` workgroup slots free.
template <typename T, typename F>
inline void main_update_beside_chain(gpmi_ctx* c, hipEvent_t after, F launch, bool masked = true) {
    if (masked && c->upd_stream && c->reserved_cus > 0) {
        hipStream_t main_s = c->stream;
        (void)hipStreamWaitEvent(c->upd_stream, after, 0);
        {
            StreamScope sc(c, c->upd_stream, c->num_cus - c->reserved_cus);
            launch();
        }
        hipEvent_t eu = la_event(c);
        (void)hipEventRecord(eu, c->upd_stream);
        (void)hipStreamWaitEvent(main_s, eu, 0);
        return;
    }
    c->gemm_reserve = c->lookahead_slots;
    c->update_late_wgs = c->update_full_grid;
    launch_chain_wait(c);  // the chain's workgroups are placed before the update takes the compute units (chain.hip chain_wait_kernel)
    launch();
    c->update_late_wgs = false;
    c->gemm_reserve = 0;
}

// whether a w-wide diagonal block can go through the persistent chain kernel (chain.hip): then ANY multiple of 64 up to its limit is a valid
// width with an explicit inverse (the multi-launch build_super_inverse merges halves: powers of two times NB only)
inline bool chain_takes(const gpmi_ctx* c, int64_t w) {
    return c->chain_kernel && c->chain_sync && !c->refine_solves && w > 0 && w % IB == 0 && w <= (int64_t)c->chain_nb_max * IB;
}
inline int64_t super_width(const gpmi_ctx* c, int64_t trailing) {
    // the tail: everything that is left is ONE diagonal block (one chain launch, nothing below it but the carried rows)
    if (c->tail_fuse > 0 && trailing <= c->tail_fuse && chain_takes(c, trailing)) return trailing;
    // widest first; widths are multiples of NB.  Below super_min[0] the factorisation is the plain NB = 256 one.
    for (int i = 2; i >= 0; --i)
        if (c->super_min[i] > 0 && trailing >= c->super_min[i]) {
            // (the widest class may be any multiple of NB the chain kernel takes: GPMI_SUPER_W, a sweep hook — 1536 / 1792 against 2048 in round 6)
            if (i == 2 && c->super_wide > 0 && chain_takes(c, c->super_wide)) return c->super_wide;
            return (int64_t)NB << (i + 1);
        }
    return NB;
}
// the widest block a factorisation of npad rows will use (scratch / store sizing)
inline int64_t super_width_max(const gpmi_ctx* c, int64_t npad) {
    int64_t w = super_width(c, npad);
    if (c->tail_fuse > 0 && chain_takes(c, std::min<int64_t>(npad, c->tail_fuse))) w = std::max<int64_t>(w, std::min<int64_t>(npad, c->tail_fuse));
    return w;
}

// step 1: Cholesky of the w x w block at (k, k), in place, right-looking in NB panels (no rows below, nothing carried)
template <typename T>
inline void factor_diag_block(gpmi_ctx* c, T* A, int64_t ld, T* linv, T* invdiag, int64_t k, int64_t w, int* d_info) {
    const int64_t kend = k + w;
    for (int64_t k0 = k; k0 < kend; k0 += NB) {
        const int64_t nbk = std::min<int64_t>(NB, kend - k0), k1 = k0 + nbk;
        factor_panel_diag<T>(c, A, ld, linv, invdiag, k0, nbk, d_info);
        if (k1 >= kend) break;
        factor_panel_rows<T>(c, A, ld, linv, k0, nbk, k1, kend, d_info);
        launch_gemm_shape<T>(c, A + k1 * ld + k1, ld, A + k1 * ld + k0, ld, A + k1 * ld + k0, ld, kend - k1, kend - k1, nbk,
                             TileShape{0, 0, 1, 0, 1, 0}, d_info, GEMM_AUX);
    }
}

// The explicit inverse LW of the w x w diagonal super-block at (k, k), w = NB * 2^s, on the CURRENT stream:
//   level 0   the NB x NB diagonal inverses (linv256_kernel over the block's own 64 x 64 inverses), placed on the diagonal
//             of LW and, transposed, of LWT (everything else zeroed);
//   level h   pairs of h x h blocks merge into 2h x 2h:  [A 0; C B]^-1 = [A^-1 0; -B^-1 C A^-1  B^-1].  With NT products only
//             (both operands K-contiguous) this takes the transposes along:  U' = A^-T C' ,  X = -B^-1 (U')' ,  X' = -U' B^-T.
// A dozen tiny launches (2 x 2 to 4 x 4 tiles): on the side stream they ride under the trailing update.
template <typename T>
inline void build_super_inverse(gpmi_ctx* c, const T* A, int64_t ld, const T* linv64, int64_t k, int64_t w, T* LW, int64_t wld,
                                const int* d_info) {
    // LW: w x w with leading dimension wld (the context's scratch, or the caller's store so that it survives the
    // factorisation); its transpose always lives in the scratch LWT with the same leading dimension
    T* LWT = (T*)c->sup_lwt;
    T* UT = (T*)c->sup_ut;
    const TileShape rect{0, 0, 0, 0, 1, 0};
    launch_linv256<T>(c, A + k * ld + k, ld, linv64 + (k / IB) * IB * IB, (T*)c->sup_l256, w, d_info);
    launch_place_inv_blocks<T>(c, (const T*)c->sup_l256, LW, LWT, wld, (int)(w / NB));
    for (int64_t h = NB; h < w; h *= 2)
        for (int64_t p = 0; p < w; p += 2 * h) {
            const T* AinvT = LWT + p * wld + p;
            const T* Binv = LW + (p + h) * wld + (p + h);
            const T* C = A + (k + p + h) * ld + (k + p);
            launch_gemm_shape<T>(c, UT, h, AinvT, wld, C, ld, h, h, h, rect, d_info, GEMM_OVERWRITE | GEMM_AUX);
            launch_gemm_shape<T>(c, LW + (p + h) * wld + p, wld, Binv, wld, UT, h, h, h, h, rect, d_info, GEMM_AUX);
            if (2 * h < w) launch_gemm_shape<T>(c, LWT + p * wld + (p + h), wld, UT, h, Binv, wld, h, h, h, rect, d_info, GEMM_AUX);
        }
}

// step 2: rows [ke, Mtot) of the super-panel's columns [ks, ke):  X <- X L^-T.
//   by_inverse: ONE product with the explicit inverse of the diagonal super-block, whose K loop ends at each column tile's
//   last column (LW is lower triangular: the same flops as the substitution), out of place into S and copied back;
//   otherwise left-looking by NB columns through the stored 64 x 64 inverses (the form that carries the refinement step).
template <typename T>
inline void rows_below_super(gpmi_ctx* c, T* A, int64_t ld, const T* linv, int64_t ks, int64_t ke, int64_t Mtot, const int* d_info,
                             const T* LW, int64_t wld) {
    if (Mtot <= ke) return;
    const int64_t M = Mtot - ke, w = ke - ks;
    if (LW) {
        T* S = (T*)c->sup_s;
        const int64_t lds = w + IB;  // never a 4 KiB-multiple row stride (w * 8 B = 16 KiB would park the stores on a few channels)
        launch_gemm_shape<T>(c, S, lds, A + ke * ld + ks, ld, LW, wld, M, w, w, TileShape{0, 0, 0, 0, 1, 0}, d_info,
                             GEMM_OVERWRITE | GEMM_KEND_COL | GEMM_AUX);
        (void)hipMemcpy2DAsync(A + ke * ld + ks, (size_t)ld * sizeof(T), S, (size_t)lds * sizeof(T), (size_t)w * sizeof(T), (size_t)M,
                               hipMemcpyDeviceToDevice, c->stream);
        return;
    }
    for (int64_t k0 = ks; k0 < ke; k0 += NB) {
        const int64_t nbk = std::min<int64_t>(NB, ke - k0);
        if (k0 > ks)
            launch_gemm_shape<T>(c, A + ke * ld + k0, ld, A + ke * ld + ks, ld, A + k0 * ld + ks, ld, M, nbk, k0 - ks,
                                 TileShape{0, 0, 0, 0, 1, 0}, d_info, GEMM_AUX);
        factor_panel_rows<T>(c, A, ld, linv, k0, nbk, ke, Mtot, d_info);
    }
}

// steps 1 + (ii): the w x w diagonal block at (k, k) factored, and — LW != nullptr — its explicit inverse built: ONE persistent launch
// (chain.hip) when it applies, the multi-launch chain of rounds 1-4 otherwise (refinement wanted, GPMI_CHAIN=0)
template <typename T>
inline void factor_and_invert_block(gpmi_ctx* c, T* A, int64_t ld, T* linv, T* invdiag, int64_t k, int64_t w, T* LW, int64_t wld, int* d_info) {
    if (launch_chain_block<T>(c, A + k * ld + k, ld, w, linv + (k / IB) * IB * IB, invdiag + k, LW, wld, d_info, k)) return;
    factor_diag_block<T>(c, A, ld, linv, invdiag, k, w, d_info);
    if (LW) build_super_inverse<T>(c, A, ld, linv, k, w, LW, wld, d_info);
}

// scratch of the inverse path for super-panels up to wmax columns and mrows rows below; GPMI_OK / GPMI_EDEVICE
template <typename T>
inline int super_scratch(gpmi_ctx* c, int64_t wmax, int64_t mrows) {
    int rc;
    if ((rc = grow(c, &c->sup_lw, &c->sup_lw_cap, wmax * (wmax + IB) * (int64_t)sizeof(T)))) return rc;
    if ((rc = grow(c, &c->sup_lwt, &c->sup_lwt_cap, wmax * (wmax + IB) * (int64_t)sizeof(T)))) return rc;
    if ((rc = grow(c, &c->sup_l256, &c->sup_l256_cap, wmax * NB * (int64_t)sizeof(T)))) return rc;
    if ((rc = grow(c, &c->sup_ut, &c->sup_ut_cap, (wmax / 2) * (wmax / 2) * (int64_t)sizeof(T)))) return rc;
    if ((rc = grow(c, &c->sup_s, &c->sup_s_cap, mrows * (wmax + IB) * (int64_t)sizeof(T)))) return rc;
    c->sup_wld = wmax + IB;
    return GPMI_OK;
}

// Where cholesky_lower keeps the super-block inverses it builds, so that later whitening (predict_f, the gradient's
// L^-T, predict_LOO) can use them: `buf` holds the w x w inverses back to back (leading dimension w each), `parts` their
// column ranges.  Without a store (or when it is full) the inverse of the current super-panel lives in the context's scratch.
template <typename T>
struct SuperStore {
    T* buf = nullptr;
    int64_t cap = 0;  // elements
    std::vector<SuperPart>* parts = nullptr;
};

template <typename T>
inline int cholesky_lower(gpmi_ctx* c, T* A, int64_t ld, T* linv, T* invdiag, int64_t npad, int64_t extra, int* d_info,
                          SuperStore<T>* store = nullptr) {
    const int64_t Mtot = npad + extra;
    const bool want_look = c->lookahead_slots > 0 && npad > 4 * NB;
    // whole CUs for the chain in short factorisations (the chain would be exposed), free slots beside a full-width update in long
    // ones: chosen per factorisation (switching inside one measured WORSE at N = 50 000: 718 against 711 ms, profiles/r03_f_*)
    const bool masked = want_look && set_lookahead_mode(c, npad < c->whole_cus_below) == 1;
    hipStream_t main_s = c->stream, side = masked ? c->side_masked : c->side_stream;
    const bool can_look = want_look && side != nullptr;
    if (store && store->parts) store->parts->clear();

    // the inverse path serves super-panels of NB * 2^s > NB columns; factorisations that carry the refinement step
    // (nugget-regularised matrices) keep the substitution through the 64 x 64 inverses
    const int64_t w0 = super_width(c, npad), wmax = super_width_max(c, npad);
    const bool inv_ok = c->super_inverse && !c->refine_solves && wmax > NB;
    if (inv_ok) {
        const int rc = super_scratch<T>(c, wmax, Mtot);
        if (rc) return rc;
    }
    // NB * 2^s (the multi-launch inverse merges halves) or whatever the chain kernel takes
    auto by_inverse = [&](int64_t w) { return inv_ok && w > NB && ((w & (w - 1)) == 0 || chain_takes(c, w)); };
    // the inverse of the super-panel at k: in the store when there is room (recorded), else in the scratch
    int64_t used = 0;
    auto place = [&](int64_t k, int64_t w, int64_t* wld) -> T* {
        const int64_t wl = w + IB;  // never a 4 KiB-multiple row stride (HBM channel conflicts on the B operand)
        if (store && store->buf && used + w * wl <= store->cap) {
            T* p = store->buf + used;
            if (store->parts) store->parts->push_back(SuperPart{k, w, used});
            used += w * wl;
            *wld = wl;
            return p;
        }
        *wld = c->sup_wld;
        return (T*)c->sup_lw;
    };

    // (a narrower FIRST super-panel — its chain is the one with nothing to hide behind — measured neutral: 1024 first 61.7 against 61.9 – 62.1 ms
    //  at N = 20 000, 678.5 against 678.4 – 680.2 at N = 50 000; 512 / 256 first lose 2 – 3 ms.  profiles/r05_g_*)
    int64_t ks = 0, ke = std::min<int64_t>(w0, npad);
    const T* LW = nullptr;  // inverse of the diagonal block of the current super-panel [ks, ke), or null
    int64_t wld = 0;
    if (by_inverse(ke) && Mtot > ke) LW = place(0, ke, &wld);
    factor_and_invert_block<T>(c, A, ld, linv, invdiag, 0, ke, const_cast<T*>(LW), wld, d_info);
    for (;;) {
        rows_below_super<T>(c, A, ld, linv, ks, ke, Mtot, d_info, LW, wld);
        if (ke >= npad) break;
        const int64_t ke2 = std::min<int64_t>(ke + super_width(c, npad - ke), npad);
        const int64_t K = ke - ks, w2 = ke2 - ke;
        const bool inv2 = by_inverse(w2) && Mtot > ke2;
        int64_t wld2 = 0;
        T* LW2 = inv2 ? place(ke, w2, &wld2) : nullptr;
        // The side stream's chain takes ~0.4 ms per 256 columns beside the update (contended CUs): look ahead only while
        // the update is longer.  Update length in units of a 128 x 128 x 256 tile product:
        const double ntile = 0.5 * (double)(npad - ke2) * (double)(npad - ke2) / (GEMM_BM * GEMM_BN) * (double)K / NB;
        if (!can_look || ntile < (double)(masked ? c->lookahead_min_tiles_masked : c->lookahead_min_tiles) * (double)((w2 + NB - 1) / NB)) {
            launch_gemm_nt<T>(c, A + ke * ld + ke, ld, A + ke * ld + ks, ld, A + ke * ld + ks, ld, Mtot - ke, npad - ke, K, 1, d_info);
            factor_and_invert_block<T>(c, A, ld, linv, invdiag, ke, w2, LW2, wld2, d_info);
        } else {
            // The next diagonal block's own tiles go FIRST.  A 256-wide block (6 workgroups of 128 x 64 tiles) fits the
            // reserved slots and rides on the side stream, as in round 1.  Wider blocks go on the main stream at full speed
            // (one under-filled round: <= 272 workgroups, ~0.2 ms) instead of trickling through the reserved slots.
            const bool tiles_on_side = w2 <= NB;
            if (!tiles_on_side)
                launch_gemm_shape<T>(c, A + ke * ld + ke, ld, A + ke * ld + ks, ld, A + ke * ld + ks, ld, w2, w2, K,
                                     TileShape{0, 0, 1, 0, 1, 0}, d_info, GEMM_AUX);
            hipEvent_t eb = la_event(c);  // columns [ks, ke) complete and their inverse consumed
            (void)hipEventRecord(eb, main_s);
            (void)hipStreamWaitEvent(side, eb, 0);
            // the update below in 256 x 128 tiles leaves eight whole CUs (one per XCD) instead of sixteen half-CU slots: one workgroup
            // per XCD and chain launch then (common.h side_slots)
            const TileShape upd_shape{0, 0, 1, (int)((w2 + GEMM_BM - 1) / GEMM_BM), 1, 0};
            c->side_one_per_xcd = !masked && update256_applies<T>(c, A + ke2 * ld + ke, ld, A + ke2 * ld + ks, ld, A + ke * ld + ks, ld, Mtot - ke2,
                                                                 npad - ke, K, upd_shape);
            {
                StreamScope sc(c, side, c->num_cus);  // the block's factorisation and its inverse: small launches
                c->beside_update = true;  // no launch larger than the reserved slots, no whole-CU kernels (common.h side_cap)
                c->chain_wide_ok = !masked && c->update_full_grid;
                if (tiles_on_side)
                    launch_gemm_shape<T>(c, A + ke * ld + ke, ld, A + ke * ld + ks, ld, A + ke * ld + ks, ld, w2, w2, K,
                                         TileShape{0, 0, 1, 0, 1, 0}, d_info, GEMM_AUX);
                factor_and_invert_block<T>(c, A, ld, linv, invdiag, ke, w2, LW2, wld2, d_info);
                c->beside_update = false;
                c->chain_wide_ok = false;
            }
            c->side_one_per_xcd = false;
            hipEvent_t ec = la_event(c);
            (void)hipEventRecord(ec, side);
            // everything below the next diagonal block: rows ke2.., columns ke.. up to each row tile's diagonal tile
            // (lower mode with offset: row tile ti keeps column tiles <= ti + g0)
            main_update_beside_chain<T>(c, eb, [&]() {
                launch_gemm_shape<T>(c, A + ke2 * ld + ke, ld, A + ke2 * ld + ks, ld, A + ke * ld + ks, ld, Mtot - ke2, npad - ke, K, upd_shape,
                                     d_info, 0);
            }, masked);
            (void)hipStreamWaitEvent(main_s, ec, 0);
        }
        ks = ke;
        ke = ke2;
        LW = LW2;
        wld = wld2;
    }
    return GPMI_OK;
}


}  // namespace gpmi
