// gemm.hip — C[M x N] -= A[M x K] * B[N x K]'  on the gfx950 matrix cores.
//
// This is the Cholesky trailing update (the dpotrf/dsyrk work behind make_posdef!,
// src/GP.jl:110) and the whiten! update of predict (src/GP.jl:27): in the row-major
// lower factor both operands have K contiguous ("NT" form), which is exactly the MFMA
// operand layout: lane l supplies A[row = l & 15][k = l >> 4].
//
// FP64 instruction: v_mfma_f64_16x16x4_f64, one per 16 x 16 x 4 product (accumulator register r of lane l holds
// D[row 4 r + (l >> 4)][col l & 15], probed: tools/mfma16_probe.hip).  History worth keeping: a bare-loop micro-benchmark
// (tools/mfma_bench.hip, profiles/r01_mfma_*) put this instruction at 48 TFLOP/s against 69-76 for v_mfma_f64_4x4x4_4b_f64,
// so the kernel was first built on FOUR 4x4x4 MFMAs per product with the B fragment rotated by v_mov_b32_dpp row_ror
// (Mfma4x4d in mfma.h, still selectable with VARIANT & 512).  Inside the real kernel the one-instruction form is 5 % (two
// workgroups per CU) to 11 % (one) faster, and it is what rocBLAS's MT128x128x16_MI16x16x4 kernel for this shape uses.
// FP32 uses v_mfma_f32_16x16x4_f32 (149.5 TF measured, 95 % of peak).
//
// Structure per 256-thread workgroup (4 wavefronts, 2 x 2):
//   128 x 128 output tile, each wavefront a 64 x 64 sub-tile = 4 x 4 MFMA tiles whose
//   accumulators (128 VGPR for fp64) live in registers for the whole K loop;
//   K is streamed in 128-byte slabs (16 doubles / 32 floats) through a double-buffered,
//   16-byte-padded LDS image filled with coalesced 16-B loads (one cache line per row);
//   fragments are read with ds_read_b128 (E consecutive k per lane — the k permutation is
//   identical for A and B, so the dot products are unchanged);
//   2 workgroups per CU (launch_bounds(256, 2)): one group's LDS reads and barrier gaps hide under the other's
//   MFMA stream (their C loads do not: co-resident groups phase-lock, see below).
// Tile order is XCD-aware: block b runs on XCD b % 8, so each XCD gets a contiguous run of
// 8 x 8 super-tiles (64 concurrent tiles share 16 operand panels in that XCD's 4 MiB L2).
#include <algorithm>
#include <type_traits>
#include <vector>

#include <hip/hip_ext.h>

#include "common.h"
#include "gemm_queue.h"
#include "mfma.h"
#include "tile_order.h"

namespace gpmi {

namespace {

// VARIANT bits are ablation switches used only by gpmi_bench_gemm (0 = the product kernel):
//   1 no C read (accumulators start at zero)   2 no stores (no epilogue at all)   4 no global loads inside the K loop
//   8 no DPP rotations (4x4x4 form only)       16 no LDS fragment reads inside the K loop
//   32 start of each CU pair delayed by a hash of its index (GPMI_STAGGER_US; measured neutral)
//   128 phase timers + timelines (tools/gemm_phases.py)   512 the four-instruction 4x4x4 + DPP form of the fp64 product
//
// PERSISTENT kernel: the grid is at most 2 workgroups per CU.  Tiles are numbered in the order of
// tile_order.h and split into 8 contiguous chunks, one per XCD (workgroup b runs on XCD b % 8 — an
// observed placement used for L2 locality only: any other placement is just slower).  A workgroup
// takes its first tile statically and every further one from its XCD's queue word (one relaxed
// atomicAdd per ~100 us tile), so the 64 tiles in flight on an XCD are 8 rows x 8 columns of one
// strip and share 16 operand panels in that XCD's 4 MiB L2.  Queue words are never reset: a launch
// owns the index range starting at `qbase[x]`, every workgroup over-pulls exactly once, and the
// word advances by exactly the chunk length, which is what the host adds to its copy.  Launches with no more
// tiles than workgroups do not touch the queue at all.
#ifdef GPMI_TOOLS
constexpr bool kTools = true;
#else
constexpr bool kTools = false;  // product build: variant 0 / 64 only, no phase lock, no 4x4x4 form, no access-width override
#endif

template <typename T, int VARIANT, int NI>
__global__ __launch_bounds__(256, NI == 4 ? 2 : 3) void gemm_nt_kernel(T* __restrict__ C, int64_t ldc, const T* __restrict__ A,
                                                         int64_t lda, const T* __restrict__ B, int64_t ldb, int64_t M,
                                                         int64_t N, int64_t K, TileShape shape,
                                                         unsigned long long* __restrict__ queue, QueueArgs qa,
                                                         const int* __restrict__ info, int flags) {
    // (VARIANT & 512, fp64, tools only: the four-instruction 4x4x4 + DPP form of the product, for the ablation)
    using MF = std::conditional_t<(VARIANT & 512) != 0 && std::is_same<T, double>::value, Mfma4x4d, Mfma<T>>;
    using Vec = typename MF::Vec;
    using Acc = typename MF::Acc;
    constexpr int E = MF::E;
    constexpr int BK = MF::BK;
    // NI = 16-column blocks per wave: 4 -> 128 x 128 tiles, two workgroups per CU; 2 -> 128 x 64 tiles (64 x 32 per wave,
    // half the accumulators, 48 KiB of LDS), THREE workgroups per CU
    constexpr int BM = GEMM_BM, BN = 32 * NI, WN = 16 * NI;  // WN = columns per wave

    // one 64 KiB LDS array: [A buf0 | A buf1 | B buf0 | B buf1]
    __shared__ __attribute__((aligned(16))) T smem[2 * (BM + BN) * BK];
    __shared__ long long s_tile;
    T* const As0 = smem;
    T* const Bs0 = smem + 2 * BM * BK;

    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;  // li-th workgroup of its XCD
    const int nloc = (gridDim.x - xcd + 7) >> 3;           // workgroups on this XCD
    const int64_t cbeg = qa.start[xcd], cend = qa.start[xcd + 1];
    if (info && *info != 0) {  // an earlier pivot failed: abandon, but keep the queue arithmetic exact
        if (qa.use_queue && tid == 0 && li == 0) {
            atomicAdd(queue + 8 * xcd, (unsigned long long)(cend - cbeg));
            if (kTools && (flags & GEMM_PHASE_LOCK)) atomicAdd(queue + 8 * xcd + 1, (unsigned long long)(cend - cbeg));
        }
        return;
    }

    // Co-residency (tools/slot_probe.hip, tools/gemm_phases.py): the two workgroups of a CU are (b, b + gridDim/2).  With
    // the 16x16x4 instruction they advance at the same rate and PHASE-LOCK (both load C, then both compute: the one that
    // is behind has the MFMA pipe to itself while the leader loads, and catches up).  Start delays, s_setprio for one of
    // the pair, C fetched inside the K loop and a per-CU load token were all measured neutral
    // (profiles/r01_gemm_c_traffic_experiments.log, LABBOOK.md 3.2), so none of them is done.
    if constexpr (VARIANT & 32) {  // experiment: spread the CUs' tile phases over one tile period (C traffic bursts)
        const unsigned half = gridDim.x >> 1;
        const unsigned c = blockIdx.x % half;  // CU pair index
        const unsigned phi = (c * 2654435769u) >> 19;  // 13 bits, golden-ratio hash
        const long long wait = ((long long)phi * (long long)(flags >> 16)) >> 13;  // flags>>16 = period in 10 ns ticks
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    if (flags & GEMM_AUX) __builtin_amdgcn_s_setprio(2);  // critical-path side launch next to the persistent update

    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);  // provably wave-uniform (LDS-DMA base goes to M0)
    const int r16 = lane & 15, g = lane >> 4;
    const int nk_full = (int)(K / BK);
    // fragment read offsets inside a row for the two k-halves: logical chunk 4h + g, swizzled by the row
    const int fo[2] = {((g) ^ (r16 & 7)) * E, ((4 + g) ^ (r16 & 7)) * E};

    // VARIANT & 128 (tools only): wave 0 accumulates wall-clock ticks (100 MHz) per phase into queue[64 + 4 * blockIdx ...]
    long long tk_pro = 0, tk_loop = 0, tk_epi = 0, tk_n = 0, tk0 = 0, tk1 = 0, tk2 = 0;
    // static first tile, then one queue pull per processed tile.  The pull for the NEXT tile is issued right after the K
    // loop, so its round trip to the L2 atomic unit (~2 us) runs under the epilogue instead of in front of the prologue.
    int64_t t = cbeg + li;
    __syncthreads();
    for (;;) {
        if constexpr (VARIANT & 128) tk0 = wall_clock64();
        if (t >= cend) break;
        if constexpr (VARIANT & 128) tk_n += 1;
        if (kTools && (flags & GEMM_PHASE_LOCK) && qa.use_queue) {
            // wait (bounded: ~17 ms, then go anyway) until the earlier rounds' tiles have left their K loops
            if (tid == 0) {
                const long long r = (t - cbeg) / nloc;
                if (r > 0) {
                    const unsigned long long target = qa.done_base[xcd] + (unsigned long long)(r * nloc);
                    int spins = 0;
                    while (__hip_atomic_load(queue + 8 * xcd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < 20000) {
                        __builtin_amdgcn_s_sleep(32);
                        ++spins;
                    }
                }
            }
            __syncthreads();
        }
        int ti, tj;
        const int64_t bi = t / qa.tiles_per;  // 0 unless batched
        tile_decode(t - bi * qa.tiles_per, shape, &ti, &tj);
        // GEMM_KEND_COL (rectangles only): a tile's K loop ends at its last column, so its cost grows with tj — up to K / 128 (K / 64) times
        // the first column's.  Walking every strip from the RIGHT hands the long tiles out first and leaves the short ones for the end
        // of each XCD's queue: the launch's tail is a 64 ... 128-deep tile instead of a K-deep one (0.33 ms at K = 2048 in 128 x 64 tiles).
        if (flags & GEMM_HEAVY_FIRST) tj = shape.ntn - 1 - tj;
        T* __restrict__ const Ct = C + bi * qa.strideC;
        const int64_t m0 = (int64_t)ti * BM, n0 = (int64_t)tj * BN;

        // Staging: every thread moves 4 x 16-B chunks of A and of B per slab straight from global
        // memory into LDS (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass).  The LDS
        // image of a slab is [128 rows][8 chunks] with NO padding (the DMA writes lane-linear);
        // bank conflicts are avoided by an XOR swizzle applied on the SOURCE side: position p of
        // row r holds global chunk p ^ (r & 7), and the fragment reads apply the same involution.
        const T* __restrict__ Ab = A + bi * qa.strideA + m0 * lda;
        const T* __restrict__ Bb = B + bi * qa.strideB + n0 * ldb;
        const int mrem = (int)((M - m0 < BM ? M - m0 : BM) - 1);  // last valid local row
        const int nrem = (int)((N - n0 < BN ? N - n0 : BN) - 1);
        int oa[4], ob[NI];  // B slab: BN rows x 8 chunks = 256 threads x NI
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 3, kc = (c & 7) ^ (row & 7);
            const int ra = row < mrem ? row : mrem;  // rows past M / N re-read the last valid row; never stored
            oa[i] = (int)(ra * lda) + kc * E;
            if (i < NI) {
                const int rb = row < nrem ? row : nrem;
                ob[i] = (int)(rb * ldb) + kc * E;
            }
        }
        auto stage = [&](int buf, int ko) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ab + oa[i] + ko),
                                                 (__attribute__((address_space(3))) void*)(As0 + buf * BM * BK + (wvu * 64 + 256 * i) * E),
                                                 16, 0, 0);
                if (i < NI)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bb + ob[i] + ko),
                                                     (__attribute__((address_space(3))) void*)(Bs0 + buf * BN * BK + (wvu * 64 + 256 * i) * E),
                                                     16, 0, 0);
            }
        };

        // The accumulators START as the C tile (fragment-shaped loads: in fp64 one wave instruction covers four whole
        // 128-byte row segments) and the MFMAs subtract into them (neg:[1,0,0] on the fp64 instruction; fp32 has no such
        // modifier and keeps -C instead).  The tile then leaves with plain stores: no read-modify-write pass, no LDS
        // transposition, no load latency at the end of a tile — the C loads fly together with the first slab's DMA.
        //   NEG:  acc = C - A B'               (OVERWRITE: acc = -A B', stored negated)
        //   !NEG: acc = -C + A B', stored negated   (OVERWRITE: acc = A B')
        const bool overwrite = (flags & GEMM_OVERWRITE) != 0;
        const bool interior = mrem == BM - 1 && nrem == BN - 1;
        // 16-byte C accesses (fp64, 16x16x4 accumulator layout only) need even leading dimension and 16-byte aligned rows
        constexpr bool PAIR16 = std::is_same<T, double>::value && std::is_same<MF, Mfma<double>>::value;
        using V2 = double __attribute__((ext_vector_type(2)));
        const bool pair16 = PAIR16 && !(ldc & 1) && !(reinterpret_cast<uintptr_t>(Ct + n0) & 15) && !(kTools && (flags & GEMM_NO_PAIR16));
        const int lrow = wm * 64, lcol = wn * WN;  // this wave's corner inside the tile
        T* __restrict__ const Cw = Ct + (m0 + lrow) * ldc + n0 + lcol;
        auto c_index = [&](int64_t ld, int ln, int mi, int ni, int r) -> int64_t {
            return (int64_t)(mi * 16 + MF::row_of(ln, r)) * ld + ni * 16 + MF::col_of(ln, r);
        };
        auto c_valid = [&](int mi, int ni, int r) -> bool {
            return lrow + mi * 16 + MF::row_of(lane, r) <= mrem && lcol + ni * 16 + MF::col_of(lane, r) <= nrem;
        };

        // GEMM_KSTART_ROW: A is upper-triangular-by-rows (A[i][k] = 0 for k < i), so the products below the
        // tile's first row vanish — start the K loop there (this is what makes K^-1 = L^-T L^-1 cost N^3/3)
        const int kbeg = (flags & GEMM_KSTART_ROW) ? (int)(m0 / BK) : 0;
        // GEMM_KEND_COL: B is lower-triangular-by-rows (B[j][k] = 0 for k > j): the products right of the tile's last
        // column vanish — stop the K loop there (product with an explicit triangular inverse, chol.h rows_below_super)
        const int nk = (flags & GEMM_KEND_COL) ? (int)(((n0 + BN < K ? n0 + BN : K)) / BK) : nk_full;
        stage(kbeg & 1, kbeg * BK);
        Acc acc[4][NI];
        if ((VARIANT & 1) || overwrite) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mi][ni].v[r] = T(0);
        } else if (interior && pair16) {
            // fp64, 16 bytes per lane: the even lane of a pair fetches (c, c+1) of the row of register 2q, the odd lane
            // (c-1, c) of the row of register 2q+1, and the two swap the halves that belong to the other (quad_perm DPP)
            if constexpr (PAIR16) {
                const bool odd = (lane & 1) != 0;
                const T* pb = Cw + (int64_t)((odd ? 4 : 0) + (lane >> 4)) * ldc + ((lane & 15) & ~1);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const V2 v = *reinterpret_cast<const V2*>(pb + (int64_t)(mi * 16 + 8 * q) * ldc + ni * 16);
                            const double recv = dpp_rot<0xB1>(odd ? v[0] : v[1]);  // the neighbour's half
                            acc[mi][ni].v[2 * q] = odd ? recv : v[0];
                            acc[mi][ni].v[2 * q + 1] = odd ? v[1] : recv;
                        }
            }
        } else if (interior) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const T c = Cw[c_index(ldc, lane, mi, ni, r)];
                        acc[mi][ni].v[r] = MF::NEG ? c : -c;
                    }
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const T c = c_valid(mi, ni, r) ? Cw[c_index(ldc, lane, mi, ni, r)] : T(0);
                        acc[mi][ni].v[r] = MF::NEG ? c : -c;
                    }
        }
        __syncthreads();  // drains the DMA (vmcnt) and publishes the slab
        if constexpr (VARIANT & 128) {
            tk1 = wall_clock64();
            tk_pro += tk1 - tk0;
        }

        unsigned long long pulled = 0;
        for (int kt = kbeg; kt < nk; ++kt) {
            const int cur = kt & 1;
            // All fragments of the slab go to registers FIRST, then the DMA of the next slab is issued, then the
            // MFMAs run.  The compiler cannot tell the DMA's LDS writes (other buffer) from LDS reads and puts
            // s_waitcnt vmcnt(0) in front of the first LDS read after a global_load_lds: with the reads interleaved
            // into the MFMA stream (the natural order) every slab waited for the NEXT slab's DMA before computing —
            // no load/compute overlap inside a workgroup (lone-workgroup K loop 2.9 us per slab against 2.08 us of
            // MFMA).  In this order the only wait is the one the end-of-slab barrier needs anyway.
            Vec af[2][4], bf[2][NI];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (VARIANT & 16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            af[h][q][e] = T(1) + T(lane) * T(1e-3);
                            if (q < NI) bf[h][q][e] = T(0.5) + T(q) * T(1e-3);
                        }
                        asm volatile("" : "+v"(af[h][q]));
                        if (q < NI) asm volatile("" : "+v"(bf[h][q]));
                    }
                } else {
                    const T* as = As0 + cur * BM * BK + (wm * 64 + r16) * BK + fo[h];
                    const T* bs = Bs0 + cur * BN * BK + (wn * WN + r16) * BK + fo[h];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) af[h][mi] = *reinterpret_cast<const Vec*>(as + mi * 16 * BK);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) bf[h][ni] = *reinterpret_cast<const Vec*>(bs + ni * 16 * BK);
                }
            }
            if constexpr (!(VARIANT & 16)) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {  // reads stay above the DMA
                        asm volatile("" : "+v"(af[h][q]));
                        if (q < NI) asm volatile("" : "+v"(bf[h][q]));
                    }
            }
            const bool last = kt + 1 == nk;
            if (!last) {
                if (!(VARIANT & 4)) stage(cur ^ 1, (kt + 1) * BK);
            } else if (qa.use_queue && tid == 0) {
                // the pull for the NEXT tile rides under the last slab's MFMAs (round trip to the L2 atomic unit ~2 us)
                pulled = atomicAdd(queue + 8 * xcd, 1ull);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        T br[4];
                        if constexpr (VARIANT & 8) {
                            br[0] = br[1] = br[2] = br[3] = bf[h][ni][e];
                        } else {
                            MF::rotations(bf[h][ni][e], br);
                        }
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) MF::mma_sub(af[h][mi][e], br, acc[mi][ni]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs above the barrier (they are what hides the DMA)
            if (last && tid == 0) {
                s_tile = (long long)(pulled - qa.base[xcd]) + cbeg + nloc;
                if (kTools && (flags & GEMM_PHASE_LOCK) && qa.use_queue) atomicAdd(queue + 8 * xcd + 1, 1ull);  // this tile's K loop is over
            }
            __syncthreads();  // (a) next slab has landed for everyone  (b) everyone is done reading `cur`  (c) s_tile
        }
        if (kbeg >= nk && qa.use_queue) {  // (no slab at all: cannot happen for the shapes launched, kept for safety)
            if (tid == 0) s_tile = (long long)(atomicAdd(queue + 8 * xcd, 1ull) - qa.base[xcd]) + cbeg + nloc;
            __syncthreads();
        }
        // the next write of s_tile is behind the next tile's prologue barrier, which no wave passes before this read
        const long long t_next = qa.use_queue ? (long long)__builtin_amdgcn_readfirstlane((int)s_tile) : (long long)cend;

        if constexpr (VARIANT & 128) {
            tk2 = wall_clock64();
            tk_loop += tk2 - tk1;
        }
        // ---- epilogue: the tile leaves straight from the accumulators -------------------------
        if constexpr (VARIANT & 2) {
            T sacc = T(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc += acc[mi][ni].v[r];
            if (sacc == T(-1.2345e300)) C[0] = sacc;
        } else {
            // GEMM_NEGOUT (with OVERWRITE): C = -A B'
            const bool negate = (overwrite ? MF::NEG : !MF::NEG) != (overwrite && (flags & GEMM_NEGOUT) != 0);
            // the addresses are recomputed from opaque copies: kept alive across the K loop they cost 32 VGPRs (spills)
            int64_t ld2 = ldc;
            int ln2 = lane;
            asm volatile("" : "+s"(ld2), "+v"(ln2));
            if (interior && pair16) {
                if constexpr (PAIR16) {
                    const bool odd = (ln2 & 1) != 0;
                    T* pb = Cw + (int64_t)((odd ? 4 : 0) + (ln2 >> 4)) * ld2 + ((ln2 & 15) & ~1);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const double a0 = negate ? -acc[mi][ni].v[2 * q] : acc[mi][ni].v[2 * q];
                                const double a1 = negate ? -acc[mi][ni].v[2 * q + 1] : acc[mi][ni].v[2 * q + 1];
                                const double recv = dpp_rot<0xB1>(odd ? a0 : a1);
                                V2 v;
                                v[0] = odd ? recv : a0;
                                v[1] = odd ? a1 : recv;
                                *reinterpret_cast<V2*>(pb + (int64_t)(mi * 16 + 8 * q) * ld2 + ni * 16) = v;
                            }
                }
            } else if (interior) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const T v = acc[mi][ni].v[r];
                            Cw[c_index(ld2, ln2, mi, ni, r)] = negate ? -v : v;
                        }
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const T v = acc[mi][ni].v[r];
                            if (c_valid(mi, ni, r)) Cw[c_index(ld2, ln2, mi, ni, r)] = negate ? -v : v;
                        }
            }
        }
        // No barrier here: the epilogue touches no LDS, and the next tile's first DMA may overwrite the slab buffers
        // because every wave left its last fragment read before the K loop's final barrier.
        t = t_next;
        if constexpr (VARIANT & 128) {
            const long long tk3 = wall_clock64();
            tk_epi += tk3 - tk2;
            const int half = (int)(gridDim.x >> 1);
            if (qa.use_queue && tid == 0 && tk_n <= 24 && (blockIdx.x == 40 || blockIdx.x == 40 + half)) {
                unsigned long long* st = queue + 64 + 2048 + (blockIdx.x == 40 ? 0 : 96) + 4 * (tk_n - 1);
                st[0] = (unsigned long long)tk0;
                st[1] = (unsigned long long)tk1;
                st[2] = (unsigned long long)tk2;
                st[3] = (unsigned long long)tk3;
            }
        }
    }
    if constexpr (VARIANT & 128) {
        if (tid == 0) {
            unsigned long long* dbg = queue + 64 + 4 * blockIdx.x;
            dbg[0] = (unsigned long long)tk_pro;
            dbg[1] = (unsigned long long)tk_loop;
            dbg[2] = (unsigned long long)tk_epi;
            dbg[3] = (unsigned long long)tk_n;
        }
    }
}

}  // namespace

template <typename T, int V, int NI>
static void launch_persistent_ni(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M,
                                 int64_t N, int64_t K, TileShape shape, const int* info, int flags, const GemmBatch* batch) {
    constexpr int BN = 32 * NI;
    shape.ntm = (int)((M + GEMM_BM - 1) / GEMM_BM);
    shape.ntn = (int)((N + BN - 1) / BN);
    if (NI == 2 && shape.mode == 1) shape.mode = 3;  // the same lower region in 128 x 64 tiles
    stair_finalize(shape);
    const int64_t tiles_per = tile_count(shape);
    const int64_t ntiles = tiles_per * (batch ? batch->count : 1);
    if (ntiles <= 0) return;
    const int per_cu = NI == 2 ? (ctx->gemm_wgs_per_cu == 1 ? 1 : 3) : ctx->gemm_wgs_per_cu;
    // beside the persistent update (side stream): no more workgroups than the slots it leaves free, and a queue of its own
    const bool side = ctx->beside_update;
    const int slots = side ? side_slots(ctx) : per_cu * ctx->num_cus - ctx->gemm_reserve;  // multiples of 8
    // small launches: one workgroup per tile (rounded up to a multiple of 8 so XCD chunks stay contiguous)
    const int grid = (int)std::min<int64_t>(slots, (ntiles + 7) / 8 * 8);
    unsigned long long* const qbase = side ? ctx->queue_base_side : ctx->queue_base;
    unsigned long long* const dbase = side ? ctx->done_base_side : ctx->done_base;
    QueueArgs qa;
    qa.use_queue = ntiles > grid;
    qa.tiles_per = tiles_per;
    qa.strideA = batch ? batch->strideA : 0;
    qa.strideB = batch ? batch->strideB : 0;
    qa.strideC = batch ? batch->strideC : 0;
    for (int x = 0; x <= 8; ++x) qa.start[x] = ntiles * x / 8;
    for (int x = 0; x < 8; ++x) {
        qa.base[x] = qbase[x];
        // (chunk - nloc) successful pulls + one failing pull per workgroup (or `chunk` failing pulls when
        // the chunk is smaller than the XCD's workgroup count): the word advances by `chunk` either way
        if (qa.use_queue) qbase[x] += (unsigned long long)(qa.start[x + 1] - qa.start[x]);
        qa.done_base[x] = dbase[x];
        if (qa.use_queue && (flags & GEMM_PHASE_LOCK)) dbase[x] += (unsigned long long)(qa.start[x + 1] - qa.start[x]);
    }
#ifdef GPMI_TOOLS
    static const bool no_pair16 = getenv("GPMI_GEMM_NO_PAIR16") != nullptr;  // tools: A/B of the C access width
    if (no_pair16) flags |= GEMM_NO_PAIR16;
#endif
    // (launches of few rounds only: there the last round's K-deep tiles are a visible tail — C2's panel solves 51.8 -> 51.3 ms per fit —, while
    //  the 15-round solves of the first N = 50 000 panels measured 1 ms per fit SLOWER walked from the right: profiles/r06_e_*)
    if ((flags & GEMM_KEND_COL) && shape.mode == 0 && ctx->kend_heavy_first && ntiles < 8 * (int64_t)slots) flags |= GEMM_HEAVY_FIRST;
    if (ctx->attach_a) {  // profiled launch: the dispatch carries its own start / stop events (ProfScope attach mode)
        hipEvent_t ea = ctx->attach_a, eb = ctx->attach_b;
        ctx->attach_a = ctx->attach_b = nullptr;
        hipExtLaunchKernelGGL((gemm_nt_kernel<T, V, NI>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, ea, eb, 0, C, ldc, A, lda, B, ldb, M, N, K,
                              shape, side ? ctx->d_queue_side : ctx->d_queue, qa, info, flags);
        return;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, V, NI>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, C, ldc, A, lda, B, ldb, M, N,
                       K, shape, side ? ctx->d_queue_side : ctx->d_queue, qa, info, flags);
}
// 128 x 64 tiles with three workgroups per CU for launches that do not fill the chip for long: measured at K = 256
// (tools/gemm_phases.py) 16 tiles 49 -> 28 us, 310 tiles 92 -> 70 us, 780 lower tiles 34.5 -> 42 TFLOP/s, and no
// difference once there are thousands of tiles.  The look-ahead main launches (gemm_reserve > 0) stay at 128 x 128:
// their reserved slots must leave whole half-CUs free for the 256-register chain kernels.  The staircase order of
// the sharded path exists for 128 x 128 only.
static bool use_narrow_tiles(const gpmi_ctx* ctx, int64_t M, int64_t N, TileShape shape, const GemmBatch* batch) {
    if (shape.mode == 2 || ctx->gemm_ni == 4) return false;
    if (ctx->gemm_ni == 2) return true;  // forced (GPMI_GEMM_NI=2)
    if (ctx->gemm_reserve != 0) return false;
    shape.ntm = (int)((M + GEMM_BM - 1) / GEMM_BM);
    shape.ntn = (int)((N + GEMM_BN - 1) / GEMM_BN);
    return tile_count(shape) * (batch ? batch->count : 1) < 2048;
}
template <typename T, int V>
static void launch_persistent(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M,
                              int64_t N, int64_t K, TileShape shape, const int* info, int flags, const GemmBatch* batch, bool narrow) {
    if (narrow)
        launch_persistent_ni<T, V, 2>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch);
    else
        launch_persistent_ni<T, V, 4>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch);
}

static double shape_entries(int64_t M, int64_t N, const TileShape& s) {
    if (s.mode == 0) return (double)M * (double)N;
    if (s.mode == 1 || s.mode == 3) {  // row i keeps columns j <= i + 128 * g0
        const double off = (double)s.g0 * GEMM_BM;
        const double tri_rows = std::max(0.0, std::min((double)M, (double)N - off));  // rows still under the diagonal
        return tri_rows * (off + 0.5 * (tri_rows + 1.0)) + ((double)M - tri_rows) * (double)N;
    }
    TileShape t = s;  // staircase: count whole tiles
    t.ntm = (int)((M + GEMM_BM - 1) / GEMM_BM);
    t.ntn = (int)((N + GEMM_BN - 1) / GEMM_BN);
    stair_finalize(t);
    return (double)tile_count(t) * GEMM_BM * GEMM_BN;
}

template <typename T>
void launch_gemm_shape(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M,
                       int64_t N, int64_t K, TileShape shape, const int* info, int flags, const GemmBatch* batch) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    // The same code under different template tags so that profilers separate the launches by name:
    // <T, 0, 4> is the Cholesky trailing update in 128 x 128 tiles (the roofline kernel of bench.py; its short tail
    // launches run as narrow tiles and are accounted with the panel), <T, 64, *> every other product.
    const bool narrow = use_narrow_tiles(ctx, M, N, shape, batch);
    const bool trailing = shape.mode && !flags && !narrow;
    // FITC's tall products (n x m matrices, n ~ 1e6) in 256 x 128 tiles as well: rectangular or overwriting or split-K batched
    // (update256.hip decides: plain or GEMM_OVERWRITE launches only, M >= update256_rect_min_m for rectangles)
    // ... and the tall panel solves X LW' (GEMM_OVERWRITE | GEMM_KEND_COL: the K loop of a column tile ends at its last column), round 6
    const bool tall = !trailing && !narrow && !(flags & ~(GEMM_OVERWRITE | GEMM_AUX | GEMM_KEND_COL)) && (shape.mode == 0 || shape.mode == 1) &&
                      (batch || M >= ctx->update256_rect_min_m);
    // the trailing update with a long K: phase-lock the tiles of an XCD (operand panels of 128 x K exceed the 4 MB L2 16 at a time)
    if (trailing && K >= ctx->phase_lock_min_k && ctx->phase_lock_min_k > 0) flags |= GEMM_PHASE_LOCK;
    // algorithmic bytes: every output entry read and written once, the operand panels once (B inside A for the SYRK shape)
    const double entries = shape_entries(M, N, shape) * (batch ? batch->count : 1);
    const bool b_in_a = B >= A && B < A + M * lda;
    const double abytes = sizeof(T) * ((flags & GEMM_OVERWRITE ? 1.0 : 2.0) * entries +
                                       ((double)M + (b_in_a ? 0.0 : (double)N)) * (double)K * (batch ? batch->count : 1));
    ProfScope ps(ctx, trailing ? GPMI_PROF_SYRK : GPMI_PROF_PANEL, 2.0 * entries * (double)K, abytes, /*attach_to_launch=*/true);
    if (trailing) {
        // the big updates of the dense path go in 256 x 128 tiles (update256.hip); everything else in 128 x 128 / 128 x 64
        if (!batch && !flags && launch_update256<T>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, 0, nullptr)) return;
        launch_persistent<T, 0>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch, false);
    } else if (tall && launch_update256<T>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags & (GEMM_OVERWRITE | GEMM_KEND_COL), batch)) {
        return;
    } else
        launch_persistent<T, 64>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch, narrow);
}

template <typename T>
void launch_gemm_nt(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M,
                    int64_t N, int64_t K, int lower, const int* info) {
    launch_gemm_shape<T>(ctx, C, ldc, A, lda, B, ldb, M, N, K, TileShape{0, 0, lower ? 1 : 0, 0, 1, 0}, info);
}

template void launch_gemm_shape<double>(gpmi_ctx*, double*, int64_t, const double*, int64_t, const double*, int64_t, int64_t,
                                        int64_t, int64_t, TileShape, const int*, int, const GemmBatch*);
template void launch_gemm_shape<float>(gpmi_ctx*, float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t,
                                       int64_t, int64_t, TileShape, const int*, int, const GemmBatch*);
template void launch_gemm_nt<double>(gpmi_ctx*, double*, int64_t, const double*, int64_t, const double*, int64_t,
                                     int64_t, int64_t, int64_t, int, const int*);
template void launch_gemm_nt<float>(gpmi_ctx*, float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t,
                                    int64_t, int64_t, int, const int*);


// ---- isolated timing of the update kernel (tools / bench only) ------------------------------
template <typename T, int V>
static void launch_variant(gpmi_ctx* ctx, T* C, int64_t ld, const T* A, int64_t M, int64_t N, int64_t K, int lower) {
    const TileShape shape{0, 0, lower ? 1 : 0, 0, 1, 0};
    int flags = 0;
#ifdef GPMI_TOOLS
    if (V & 32) {
        const char* e = getenv("GPMI_STAGGER_US");
        flags = (int)((e ? atof(e) : 80.0) * 100.0) << 16;
    }
#endif
    launch_persistent<T, V>(ctx, C, ld, A, ld, A, ld, M, N, K, shape, nullptr, flags, nullptr, use_narrow_tiles(ctx, M, N, shape, nullptr));
}

template <typename T>
int gemm_bench(gpmi_ctx* ctx, int64_t M, int64_t N, int64_t K, int lower, int variant, int iters, double* ms_out) {
    // operands: rows of a (M x ld) random matrix; C is its own buffer of the same shape
    const int64_t ld = ((std::max(N, K) + 63) / 64) * 64;
    T *A = nullptr, *C = nullptr;
    GPMI_HIP(ctx, hipMalloc(&A, (size_t)(M * ld) * sizeof(T)));
    GPMI_HIP(ctx, hipMalloc(&C, (size_t)(M * ld) * sizeof(T)));
    std::vector<T> h((size_t)(M * ld));
    uint64_t st = 88172645463325252ull;
    for (auto& v : h) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        v = (T)((double)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
    GPMI_HIP(ctx, hipMemcpy(A, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    GPMI_HIP(ctx, hipMemset(C, 0, h.size() * sizeof(T)));
    hipEvent_t e0, e1;
    GPMI_HIP(ctx, hipEventCreate(&e0));
    GPMI_HIP(ctx, hipEventCreate(&e1));
#ifdef GPMI_TOOLS
    // tools: the same launch as the look-ahead makes it — GPMI_BENCH_STREAM=reserve: plain stream, 16 slots (8 CUs) left free;
    // =masked: the CU-masked update stream (248 CUs)
    const char* bs = getenv("GPMI_BENCH_STREAM");
    const bool on_masked = bs && bs[0] == 'm' && ctx->upd_stream;
    hipStream_t s_keep = ctx->stream;
    const int cus_keep = ctx->num_cus;
    if (on_masked) {
        ctx->stream = ctx->upd_stream;
        ctx->num_cus = cus_keep - 8;
    } else if (bs && bs[0] == 'r') {
        ctx->gemm_reserve = 16;
    }
#endif
    auto go = [&]() {
        switch (variant) {
            case 0: launch_variant<T, 0>(ctx, C, ld, A, M, N, K, lower); break;
            case 256: case 257: case 259: case 261: case 263: case 271:  // the 256 x 128 form of the trailing update (update256.hip; + ablation bits in tools builds)
                ctx->update256_ablation = variant - 256;
                if (!lower || !launch_update256<T>(ctx, C, ld, A, ld, A, ld, M, N, K, TileShape{0, 0, 1, 0, 1, 0}, nullptr, 0, nullptr))
                    launch_variant<T, 0>(ctx, C, ld, A, M, N, K, lower);
                break;
#ifdef GPMI_TOOLS  // ablations of tools/gemm_ablate.py / gemm_phases.py: not instantiated in the product library
            case 1: launch_variant<T, 1>(ctx, C, ld, A, M, N, K, lower); break;
            case 2: launch_variant<T, 2>(ctx, C, ld, A, M, N, K, lower); break;
            case 4: launch_variant<T, 4>(ctx, C, ld, A, M, N, K, lower); break;
            case 6: launch_variant<T, 6>(ctx, C, ld, A, M, N, K, lower); break;
            case 8: launch_variant<T, 8>(ctx, C, ld, A, M, N, K, lower); break;
            case 14: launch_variant<T, 14>(ctx, C, ld, A, M, N, K, lower); break;
            case 22: launch_variant<T, 22>(ctx, C, ld, A, M, N, K, lower); break;
            case 30: launch_variant<T, 30>(ctx, C, ld, A, M, N, K, lower); break;
            case 32: launch_variant<T, 32>(ctx, C, ld, A, M, N, K, lower); break;
            case 160: launch_variant<T, 160>(ctx, C, ld, A, M, N, K, lower); break;
            case 128: launch_variant<T, 128>(ctx, C, ld, A, M, N, K, lower); break;
            case 512: launch_variant<T, 512>(ctx, C, ld, A, M, N, K, lower); break;
            case 640: launch_variant<T, 640>(ctx, C, ld, A, M, N, K, lower); break;
            case 514: launch_variant<T, 514>(ctx, C, ld, A, M, N, K, lower); break;
            case 542: launch_variant<T, 542>(ctx, C, ld, A, M, N, K, lower); break;
#endif
            default: break;
        }
    };
    go();
    GPMI_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters; ++i) go();
    GPMI_HIP(ctx, hipEventRecord(e1, ctx->stream));
    GPMI_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    GPMI_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *ms_out = (double)ms / iters;
#ifdef GPMI_TOOLS
    ctx->stream = s_keep;
    ctx->num_cus = cus_keep;
    ctx->gemm_reserve = 0;
#endif
    if (kTools && (variant & 128)) {  // per-phase wall-clock split of the last launch (100 MHz ticks), averaged over workgroups
        std::vector<unsigned long long> dbg(4 * 512);
        GPMI_HIP(ctx, hipMemcpy(dbg.data(), ctx->d_queue + 64, dbg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double pro = 0, loop = 0, epi = 0, nt = 0;
        for (int w = 0; w < 512; ++w) {
            pro += (double)dbg[4 * w];
            loop += (double)dbg[4 * w + 1];
            epi += (double)dbg[4 * w + 2];
            nt += (double)dbg[4 * w + 3];
        }
        {
            std::vector<unsigned long long> st(192);
            GPMI_HIP(ctx, hipMemcpy(st.data(), ctx->d_queue + 64 + 2048, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            const unsigned long long z = std::min(st[0], st[96]);
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "[gemm timeline] WG %s (us: start/kloop/epi/end):", w ? "40+half" : "40");
                for (int t = 0; t < 10; ++t)
                    fprintf(stderr, " %.0f/%.0f/%.0f/%.0f", (st[96 * w + 4 * t] - z) * 0.01, (st[96 * w + 4 * t + 1] - z) * 0.01,
                            (st[96 * w + 4 * t + 2] - z) * 0.01, (st[96 * w + 4 * t + 3] - z) * 0.01);
                fprintf(stderr, "\n");
            }
        }
        fprintf(stderr, "[gemm phases] tiles %.0f; per tile: pull+prologue %.2f us, K loop %.2f us, epilogue %.2f us; per WG total %.1f us\n",
                nt, pro / nt * 0.01, loop / nt * 0.01, epi / nt * 0.01, (pro + loop + epi) / 512 * 0.01);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(A);
    hipFree(C);
    return GPMI_OK;
}
template int gemm_bench<double>(gpmi_ctx*, int64_t, int64_t, int64_t, int, int, int, double*);
template int gemm_bench<float>(gpmi_ctx*, int64_t, int64_t, int64_t, int, int, int, double*);


}  // namespace gpmi
