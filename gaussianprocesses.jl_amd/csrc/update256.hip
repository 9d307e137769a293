// update256.hip — the Cholesky trailing update  C[M x N] -= A[M x K] * B[N x K]'  (lower region, the dsyrk work behind
// make_posdef!, src/GP.jl:110) in 256 x 128 output tiles: the round-3 form of the roofline kernel.
//
// What round 2's 128 x 128 kernel (gemm.hip, two 256-thread workgroups per CU) still lost at K = 1024 / 2048 was not C traffic any
// more but the K loop itself: its ablation puts the operand DMA at 8 % and the LDS fragment reads at 6 % of the kernel (LABBOOK 3.2), because
// with TWO slab buffers a workgroup can read the fragments of slab k + 1 only after the barrier that publishes it — every slab
// starts with an LDS round trip during which that workgroup's waves issue no MFMA, and co-resident workgroups phase-lock.
// This kernel removes the round trip instead of hiding it behind a second workgroup:
//   * ONE 512-thread workgroup per CU (8 wavefronts, 4 x 2, each a 64 x 64 sub-tile = 4 x 4 MFMA tiles, 128 accumulator
//     VGPRs in fp64 — the same per-wave shape as before), 256 x 128 output tile: 384 operand rows per slab for the flops of
//     two 128 x 128 tiles (512 rows): a quarter less operand traffic through L2, LDS and HBM;
//   * THREE slab buffers (3 x 48 KiB of the CU's 160 KiB LDS): the barrier at the end of slab k publishes slab k + 2, so
//     slab k + 1 is already complete when slab k is being multiplied, and the fragments are software-pipelined in halves ACROSS
//     the barrier: while the 32 MFMAs of (slab k, k-half 0) issue, the fragments of (k, half 1) are in flight; while those
//     multiply, the fragments of (k + 1, half 0).  A wave arrives at every barrier with the operands of its next 32 MFMAs in
//     registers — the matrix pipe is never drained by an LDS latency;
//   * everything that is not an MFMA is issued INSIDE the MFMA stream, after groups of four: one LDS-DMA piece after each of the
//     first half's groups 1 - 6, the eight ds_read_b128 of the other half after group 0 (both waves of a SIMD stand at the same
//     place in the code: six pieces issued in a row after the barrier were ~600 cycles without an MFMA);
//   * the fragment reads and the LDS-DMA are inline assembly: hipcc cannot tell the LDS writes of global_load_lds from LDS reads
//     and puts s_waitcnt vmcnt(0) in front of every LDS read it sees after a DMA (gemm.hip, K loop comment), and it waits for the
//     DMA before it overwrites a register the DMA used as its address — either would serialise exactly this pipeline; the waits
//     are written by hand (lgkmcnt before use, vmcnt(0) before the barrier);
//   * the C tile is read in the EPILOGUE, into the then-free fragment registers (four batches of four fragments, two in flight),
//     while the next tile's first two slabs are already on their way (its index comes from the queue pull issued under the last
//     slab), and the tile's 32 stores per lane drain under the next tile's first slab.  (C inside the K loop — one fragment per
//     slab during the first 16 — was built first; hipcc's register allocation does not survive it: LABBOOK.md 3.2b.)
// Tile order: tile_order.h mode 3 (the lower region in tiles twice as tall as wide), per-XCD persistent queues as in gemm.hip.
// Used for the dense path's big updates only (launch_update256 says when); everything else stays on gemm_nt_kernel.
#include <algorithm>
#include <type_traits>
#include <utility>

#include <hip/hip_ext.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include <vector>
#include "gemm_queue.h"
#include "mfma.h"
#include "tile_order.h"

namespace gpmi {

namespace {

constexpr int U_BM = 256, U_BN = 128, U_NBUF = 3;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <typename Vec, int OFF>
__device__ __forceinline__ Vec lds_read16(unsigned addr) {
    Vec v;
    // (no "memory" clobber: with it hipcc treats the statement as an LDS access of unknown address and waits for the DMA in flight)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// wait for every outstanding LDS read; the operands tie the fragments to the wait (their consumers cannot move above it)
template <typename Vec>
__device__ __forceinline__ void lds_wait(Vec (&a)[4], Vec (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
template <typename Vec>
__device__ __forceinline__ void slab_wait(Vec (&a)[4], Vec (&b)[4]) {  // own DMA parts landed, prefetched fragments arrived
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 :
                 : "memory");
}

// ABL (tools builds only, gpmi_bench_gemm variants 257 ...): 1 no epilogue (no C read, no store)  2 no operand DMA after the first two
// slabs  4 no fragment reads after the first  8 no slab barrier (with 2 | 4: the bare MFMA stream of the loop)
// OVW: C = A B' (GEMM_OVERWRITE: the C tile is not read) instead of C -= A B'.  Batched launches (QueueArgs::tiles_per / stride*: split-K
// with separate outputs) and rectangular regions (tile_order mode 0) take the same path: round 4 widened the kernel from the Cholesky
// trailing update to FITC's n m^2 products (W = Kfu Luu^-T over 10^6 rows, U' U'' in 16 K-chunks).
// ATOM (interior tiles of a subtracting launch): the epilogue does not read C at all — every accumulator element goes out as ONE
// no-return global_atomic_add (the L2 does the read-modify-write; exactly one add per element, so the result is bit-identical and
// deterministic): no load latency left in the epilogue (LABBOOK 3.2b: the epilogue was 2.4 % of the kernel at K = 2048, 5.5 % at
// K = 1024 — four dependent load batches per tile, not bytes).
// KEND (with OVW; rectangles): B is lower-triangular-by-rows (B[j][k] = 0 for k > j — an explicit triangular inverse: the panel solve
// X LW' of chol.h rows_below_super), so a tile's K loop ends at its last column (gemm.hip GEMM_KEND_COL).  Its own instantiation: the
// trailing update's code is not touched by it.
template <typename T, int ABL, bool OVW, bool ATOM = false, bool KEND = false>
__global__ __launch_bounds__(512, 2) void update256_kernel(T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda,
                                                           const T* __restrict__ B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                                                           TileShape shape, unsigned long long* __restrict__ queue, QueueArgs qa,
                                                           const int* __restrict__ info) {
    using MF = Mfma<T>;
    using Vec = typename MF::Vec;
    using Acc = typename MF::Acc;
    constexpr int E = MF::E;
    constexpr int BK = MF::BK;                         // one slab = 128 bytes of every operand row
    constexpr int SLAB = (U_BM + U_BN) * BK;           // elements per buffer: [256 rows of A | 128 rows of B], 8 chunks of 16 B each
    constexpr unsigned SLAB_BYTES = SLAB * sizeof(T);  // 49152
    constexpr int ROW16 = 16 * BK * (int)sizeof(T);    // 16 rows further on: 2048 bytes
    constexpr int ES = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ long long s_tile;

    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int nloc = (gridDim.x - xcd + 7) >> 3;
    const int64_t cbeg = qa.start[xcd], cend = qa.start[xcd + 1];
    if (info && *info != 0) {  // an earlier pivot failed: abandon, but keep the queue arithmetic exact
        if (qa.use_queue && tid == 0 && li == 0) atomicAdd(queue + 8 * xcd, (unsigned long long)(cend - cbeg));
        return;
    }
    // Everything a lane keeps across the K loop besides accumulators and fragments is 32-bit: ONE LDS offset (the other three
    // fragment addresses differ from it by wave-uniform amounts and one XOR) and six staging offsets.  The pointers they are added
    // to are wave-uniform (scalar registers).
    const int lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int wm = wvu >> 1, wn = wvu & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const int nk_full = (int)(K / BK);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    // fragment rows: byte offset of (row r16, logical chunk g) of this wave's first A block; k-half 1 is chunk g + 4 = the
    // same address with bit 6 flipped (rows are 128 bytes); B rows lie a wave-uniform distance further on
    const unsigned fr0 = (unsigned)((r16 * BK + ((g ^ (r16 & 7)) * E)) * ES);
    const unsigned a_base = lds0 + (unsigned)(wm * 64 * BK * ES);
    const unsigned b_base = lds0 + (unsigned)((U_BM + wn * 64) * BK * ES);
    auto read_frags = [&](Vec(&a)[4], Vec(&b)[4], unsigned buf_bytes, int h) __attribute__((always_inline)) {
        // (opaque copy: hipcc otherwise keeps all twelve (buffer, half, operand) addresses of the unrolled slabs in registers)
        unsigned f = fr0;
        asm volatile("" : "+v"(f));
        if (h) f ^= 64u;
        const unsigned aa = f + (a_base + buf_bytes), bb = f + (b_base + buf_bytes);
        a[0] = lds_read16<Vec, 0>(aa);
        a[1] = lds_read16<Vec, ROW16>(aa);
        a[2] = lds_read16<Vec, 2 * ROW16>(aa);
        a[3] = lds_read16<Vec, 3 * ROW16>(aa);
        b[0] = lds_read16<Vec, 0>(bb);
        b[1] = lds_read16<Vec, ROW16>(bb);
        b[2] = lds_read16<Vec, 2 * ROW16>(bb);
        b[3] = lds_read16<Vec, 3 * ROW16>(bb);
    };

    constexpr bool PAIR16 = std::is_same<T, double>::value;
    using V2 = double __attribute__((ext_vector_type(2)));
    using No = std::false_type;
    using Yes = std::true_type;
    const unsigned row0 = (unsigned)tid >> 3, kc16 = (unsigned)(((tid & 7) ^ ((tid >> 3) & 7)) * 16);
    const unsigned lda_b = (unsigned)(lda * ES), ldb_b = (unsigned)(ldb * ES);  // row strides in bytes (< 2^24: launch_update256)

    // staging: chunk c = tid + 512 i of a slab image is row (tid >> 3) + 64 i, position tid & 7, and holds global chunk
    // (tid & 7) ^ (row & 7) (the same for every i).  Rows past M / N re-read the last valid row; they are never stored.
    // The six byte offsets of a tile stay in six registers that nothing else writes during its K loop, and the slab advance is in
    // the scalar base: hipcc waits for every DMA in flight before it overwrites a register one of them used as its address.
    unsigned oa[4], ob[2];
    auto offsets = [&](int mrem, int nrem) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            oa[i] = min(row0 + 64u * i, (unsigned)mrem) * lda_b + kc16;
            if (i < 2) ob[i] = min(row0 + 64u * i, (unsigned)nrem) * ldb_b + kc16;
        }
    };
    // The DMA itself is written in assembly, in the scalar-base + 32-bit-offset form: through the builtin hipcc builds 64-bit
    // per-lane pointers in temporaries and then waits for the DMA before reusing them.  vmcnt is waited for by hand (slab_wait,
    // prologue); waits hipcc computes for its own loads / stores can only come out stricter for the extra operations in flight.
    auto dma16 = [&](unsigned voff, const char* sbase, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr));  // (m0 is written: nothing else in this kernel keeps a value in it)
    };
    // piece j of a slab's six: j = 0..3 the A rows (tid >> 3) + 64 j, j = 4, 5 the B rows (tid >> 3) + 64 (j - 4)
    auto stage_piece = [&](auto jtag, const char* Ab, const char* Bb, int buf, int kt) __attribute__((always_inline)) {
        constexpr int j = decltype(jtag)::value;
        const unsigned base = lds0 + (unsigned)buf * SLAB_BYTES + (unsigned)(wvu * 64 * 16);  // this wave's 1 KiB of the image
        if constexpr (j < 4)
            dma16(oa[j], Ab + (size_t)kt * (BK * ES), base + 512u * 16u * j);
        else
            dma16(ob[j - 4], Bb + (size_t)kt * (BK * ES), base + (unsigned)(U_BM * BK * ES) + 512u * 16u * (j - 4));
    };
    auto stage = [&](const char* Ab, const char* Bb, int buf, int kt) __attribute__((always_inline)) {
        asm volatile("s_nop 4");  // the scalar bases may come straight from a v_readfirstlane (VALU-written SGPR -> VMEM base)
        static_for<0, 6>([&](auto j) __attribute__((always_inline)) { stage_piece(j, Ab, Bb, buf, kt); });
    };
    // a tile's operand rows and extent (wave-uniform)
    struct Tile {
        const char *Ab, *Bb;
        int64_t m0, n0, coff;  // coff: the batch's offset into C (elements)
        int mrem, nrem;        // last valid local row / column
    };
    auto locate = [&](int64_t t) __attribute__((always_inline)) -> Tile {
        int ti, tj;
        const int64_t bt = t / qa.tiles_per;  // batch index (0 for plain launches: tiles_per = the tile count)
        tile_decode(t - bt * qa.tiles_per, shape, &ti, &tj);
        Tile w;
        w.m0 = (int64_t)ti * U_BM;
        w.n0 = (int64_t)tj * U_BN;
        w.coff = bt * qa.strideC;
        w.Ab = reinterpret_cast<const char*>(A + bt * qa.strideA + w.m0 * lda);
        w.Bb = reinterpret_cast<const char*>(B + bt * qa.strideB + w.n0 * ldb);
        w.mrem = (int)((M - w.m0 < U_BM ? M - w.m0 : U_BM) - 1);
        w.nrem = (int)((N - w.n0 < U_BN ? N - w.n0 : U_BN) - 1);
        return w;
    };

    int64_t t = cbeg + li;
    __syncthreads();
    if (t >= cend) return;
    Tile cur_t = locate(t);
    bool prev_stores32 = false;  // the tile before this one ended with exactly 32 stores per lane (fp64, interior)
    offsets(cur_t.mrem, cur_t.nrem);
    stage(cur_t.Ab, cur_t.Bb, 0, 0);
    stage(cur_t.Ab, cur_t.Bb, 1, 1);
    for (;;) {
        const char* const Ab = cur_t.Ab;
        const char* const Bb = cur_t.Bb;
        const int mrem = cur_t.mrem, nrem = cur_t.nrem;
        // (KEND: at least U_BN / BK = 8 (fp64) / 4 (fp32) slabs — the two the prologue stages are always inside the loop)
        const int nk = KEND ? (int)((cur_t.n0 + U_BN < K ? cur_t.n0 + U_BN : K) / BK) : nk_full;
        Acc acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni].v[r] = T(0);
        // slabs 0 and 1 have landed; the previous tile's stores (issued AFTER those DMAs: vmcnt retires in order) may still be in
        // flight in fp64 — 32 per lane — and drain under the first slab, whose closing wait is vmcnt(0)
        if (PAIR16 && prev_stores32)
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        Vec fa[2][4], fb[2][4];  // [k-half][16-row block]
        read_frags(fa[0], fb[0], 0u, 0);
        lds_wait(fa[0], fb[0]);

        // 16 x E MFMAs of one k-half in groups of four (one B column block against the four A row blocks); `between(g)` is
        // issued after group g.  The slab's other instructions ride in those gaps — an LDS-DMA piece costs a wave 60 - 180 cycles
        // of issue (MI355X_MICROARCH.md), and with ONE workgroup per CU both waves of a SIMD stand at the same place: six pieces
        // issued in a row after the barrier are ~600 cycles in which neither feeds the matrix pipe.
        auto mfmas = [&](Vec(&a)[4], Vec(&b)[4], auto between) __attribute__((always_inline)) {
            static_for<0, 4 * E>([&](auto gt) __attribute__((always_inline)) {
                constexpr int gi = decltype(gt)::value, e = gi / 4, ni = gi % 4;
                T br[4];
                MF::rotations(b[ni][e], br);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) MF::mma_sub(a[mi][e], br, acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
                between(gt);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // One slab: kt its index, cur / nx1 / nx2 the buffers of slabs kt, kt + 1, kt + 2.  The tile's LAST slab is its own instance:
        // the queue pull is a returning atomic, and with one anywhere in the rolled loop hipcc's wait insertion puts
        // s_waitcnt vmcnt(0) behind every slab's first instructions.
        unsigned long long pulled = 0;
        auto slab = [&](auto last_tag, int kt, int cur, int nx1, int nx2) __attribute__((always_inline)) {
            constexpr bool last = decltype(last_tag)::value;
            const bool more_slabs = !last && kt + 2 < nk;
            if constexpr (last) {
                if (qa.use_queue && tid == 0) pulled = atomicAdd(queue + 8 * xcd, 1ull);  // the pull for the NEXT tile rides under these MFMAs
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[0], fb[0], [&](auto gt) __attribute__((always_inline)) {
                constexpr int gi = decltype(gt)::value;
                if constexpr (gi == 0 && !(ABL & 4)) read_frags(fa[1], fb[1], (unsigned)cur * SLAB_BYTES, 1);
                if constexpr (!last && gi >= 1 && gi <= 6 && !(ABL & 2)) {
                    if (more_slabs) stage_piece(std::integral_constant<int, gi - 1>{}, Ab, Bb, nx2, kt + 2);
                }
            });
            lds_wait(fa[1], fb[1]);
            mfmas(fa[1], fb[1], [&](auto gt) __attribute__((always_inline)) {
                constexpr int gi = decltype(gt)::value;
                if constexpr (!last && gi == 0 && !(ABL & 4)) read_frags(fa[0], fb[0], (unsigned)nx1 * SLAB_BYTES, 0);  // complete since the previous barrier
            });
            if constexpr (last) {
                if (tid == 0) s_tile = (long long)(pulled - qa.base[xcd]) + cbeg + nloc;
            }
            slab_wait(fa[0], fb[0]);
            if constexpr (!(ABL & 8) || last) __builtin_amdgcn_s_barrier();  // slab kt + 2 has landed for everyone; everyone is done reading slab kt; s_tile
        };
        {
            int cur = 0, nx1 = 1, nx2 = 2;
#pragma clang loop unroll(disable)
            for (int kt = 0; kt + 1 < nk; ++kt) {
                slab(No{}, kt, cur, nx1, nx2);
                const int o = cur;
                cur = nx1;
                nx1 = nx2;
                nx2 = o;
            }
            slab(Yes{}, nk - 1, cur, nx1, nx2);
        }
        const long long t_next = qa.use_queue ? (long long)__builtin_amdgcn_readfirstlane((int)s_tile) : (long long)cend;

        // The NEXT tile's first two slabs start now (every wave left its last fragment read before the K loop's final barrier): their
        // latency runs under this tile's epilogue.
        const bool interior = mrem == U_BM - 1 && nrem == U_BN - 1;
        T* __restrict__ const Cw = C + cur_t.coff + (cur_t.m0 + wm * 64) * ldc + cur_t.n0 + wn * 64;  // this wave's corner of the C tile (wave-uniform)
        const int lrow = wm * 64, lcol = wn * 64;
        const bool more = t_next < cend;
        if (more) {
            cur_t = locate(t_next);
            offsets(cur_t.mrem, cur_t.nrem);
            stage(cur_t.Ab, cur_t.Bb, 0, 0);
            stage(cur_t.Ab, cur_t.Bb, 1, 1);
        }

        // ---- epilogue.  NEG (fp64): acc = -A B', C += acc;  !NEG (fp32): acc = A B', C -= acc.  The C tile is read HERE, in two
        // batches of eight 16 x 16 fragments that land in the (now free) fragment registers: two exposed load latencies per tile
        // (~1 % at K = 2048) instead of a prologue.  Addresses from opaque copies: computed from `lane` / `ldc` directly they are
        // invariant across TILES, hipcc hoists them out of the persistent loop and keeps ~40 registers alive across the K loop.
        int ln2 = lane;
        int64_t ld2 = ldc;
        asm volatile("" : "+v"(ln2), "+s"(ld2));
        if constexpr (ABL & 1) {
            T sacc = T(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc += acc[mi][ni].v[r];
            if (sacc == T(-1.2345e30)) C[0] = sacc;
        } else if (ATOM && !OVW && interior) {
            T* const pc = Cw + (int64_t)MF::row_of(ln2, 0) * ld2 + MF::col_of(ln2, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const T v = acc[mi][ni].v[r];
                        unsafeAtomicAdd(pc + (int64_t)(mi * 16 + r * MF::RSTEP) * ld2 + ni * 16, MF::NEG ? v : -v);
                    }
        } else if (interior) {
            if constexpr (PAIR16) {
                // fp64, 16-byte accesses (gemm.hip): the even lane of a pair moves (c, c + 1) of the row of register 2q, the odd lane
                // (c - 1, c) of the row of register 2q + 1, and the two swap the halves that belong to the other.  Four batches of
                // four fragments (one row block each), two in flight: batch b + 1 is requested before batch b is added and stored.
                const bool odd2 = (ln2 & 1) != 0;
                T* const pc = Cw + (int64_t)((odd2 ? 4 : 0) + (ln2 >> 4)) * ld2 + ((ln2 & 15) & ~1);
                V2 cv[2][4][2];
                auto fetch = [&](int mi, V2(&dst)[4][2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int q = 0; q < 2; ++q) dst[ni][q] = *reinterpret_cast<const V2*>(pc + (int64_t)(mi * 16 + 8 * q) * ld2 + ni * 16);
                };
                if constexpr (!OVW) fetch(0, cv[0]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if constexpr (!OVW)
                        if (mi + 1 < 4) fetch(mi + 1, cv[(mi + 1) & 1]);
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const double a0 = acc[mi][ni].v[2 * q], a1 = acc[mi][ni].v[2 * q + 1];
                            const double recv = dpp_rot<0xB1>(odd2 ? a0 : a1);
                            V2 v;
                            if constexpr (OVW) {  // acc = -A B' (neg MFMA): the product itself is its negative
                                v[0] = -(odd2 ? recv : a0);
                                v[1] = -(odd2 ? a1 : recv);
                            } else {
                                v = cv[mi & 1][ni][q];
                                v[0] += odd2 ? recv : a0;
                                v[1] += odd2 ? a1 : recv;
                            }
                            *reinterpret_cast<V2*>(pc + (int64_t)(mi * 16 + 8 * q) * ld2 + ni * 16) = v;
                        }
                }
            } else {
                T* const pc = Cw + (int64_t)MF::row_of(ln2, 0) * ld2 + MF::col_of(ln2, 0);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    T cv[2][4][4];
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r) cv[m2][ni][r] = OVW ? T(0) : pc[(int64_t)((2 * half + m2) * 16 + r * MF::RSTEP) * ld2 + ni * 16];
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int mi = 2 * half + m2;
                                const T v = acc[mi][ni].v[r];
                                pc[(int64_t)(mi * 16 + r * MF::RSTEP) * ld2 + ni * 16] =
                                    OVW ? (MF::NEG ? -v : v) : (MF::NEG ? cv[m2][ni][r] + v : cv[m2][ni][r] - v);
                            }
                }
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = mi * 16 + MF::row_of(ln2, r), col = ni * 16 + MF::col_of(ln2, r);
                        if (lrow + row <= mrem && lcol + col <= nrem) {
                            T* const pc = Cw + (int64_t)row * ld2 + col;
                            const T v = acc[mi][ni].v[r];
                            *pc = OVW ? (MF::NEG ? -v : v) : (MF::NEG ? *pc + v : *pc - v);
                        }
                    }
        }
        if (!more) break;
        prev_stores32 = PAIR16 && interior && !(ABL & 1) && !(ATOM && !OVW);  // (the atomic epilogue has 64 operations per lane in flight: full wait)
    }
}

template <typename T, int ABL, bool OVW, bool ATOM, bool KEND = false>
bool prepare(gpmi_ctx* ctx) {
    // 144 KiB of dynamic LDS need the attribute once per device and instantiation
    static bool done[64] = {false};
    const int dev = ctx->device & 63;
    if (done[dev]) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&update256_kernel<T, ABL, OVW, ATOM, KEND>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            U_NBUF * (U_BM + U_BN) * 128) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done[dev] = true;
    return true;
}

}  // namespace

// The update in 256 x 128 tiles when it applies: a lower-mode region (shape as given to launch_gemm_shape, offsets in 128-row
// tiles) whose row offset is a multiple of 256, K in whole slabs and at least 16 of them (the C fragments arrive during the
// first 16), not beside another persistent launch, and enough tiles to fill the chip for several rounds.  Returns false
// without launching anything otherwise.
// Does the 256 x 128 kernel take this update?  (a lower-mode region whose row offset is a multiple of 256, K in whole slabs and at least
// 16 of them, a plain stream, operands aligned for 16-byte accesses, enough tiles to fill the chip for several rounds)
template <typename T>
static bool update256_plan(const gpmi_ctx* ctx, const T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M, int64_t N,
                           int64_t K, TileShape shape, TileShape* out, int64_t* ntiles_out, int flags = 0, const GemmBatch* batch = nullptr) {
    constexpr int BK = Mfma<T>::BK;
    if (!ctx->update256 || (shape.mode != 1 && shape.mode != 0) || (shape.g0 & 1) || ctx->beside_update) return false;
    // (K-loop start per tile, negated output: the 128 x 128 kernel's; the K-loop END per column tile only for overwriting rectangles)
    if (flags & ~(GEMM_OVERWRITE | GEMM_KEND_COL)) return false;
    if ((flags & GEMM_KEND_COL) && (!(flags & GEMM_OVERWRITE) || shape.mode != 0 || batch || !ctx->update256_kend)) return false;
    // rectangles: the tall products only (FITC's n x m matrices; predict_f's P x N updates stay on the 128 x 128 kernel, whose tile
    // count fills the chip in more even rounds at M = 1024)
    if (shape.mode == 0 && M < ctx->update256_rect_min_m) return false;
    if (batch) {
        const int64_t a16 = 16 / (int64_t)sizeof(T);
        if (batch->count <= 0 || (batch->strideA % a16) || (batch->strideB % a16) || (batch->strideC % a16)) return false;
    }
    // on the CU-masked update stream the kernel measured 10 % slower than the 128 x 128 one (profiles/r03_r_update256.log)
#ifdef GPMI_TOOLS
    static const bool on_masked_too = getenv("GPMI_UPDATE256_ON_MASKED") != nullptr;  // tools: measure it there (tools/update256_streams.py)
#else
    constexpr bool on_masked_too = false;
#endif
    if (ctx->upd_stream && ctx->stream == ctx->upd_stream && !on_masked_too) return false;
    if (K % BK != 0 || K / BK < 16 || (lda % (16 / (int)sizeof(T))) || (ldb % (16 / (int)sizeof(T)))) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return false;
    if (sizeof(T) == 8 && ((ldc & 1) || (reinterpret_cast<uintptr_t>(C) & 15))) return false;  // fp64: 16-byte accesses to the C tile
    if (lda * (int64_t)sizeof(T) >= (1 << 24) || ldb * (int64_t)sizeof(T) >= (1 << 24)) return false;  // 32-bit staging offsets: 256 rows x stride
    TileShape s = shape;
    if (shape.mode == 1) {
        s.mode = 3;
        s.g0 = shape.g0 / 2;
    }
    s.ntm = (int)((M + U_BM - 1) / U_BM);
    s.ntn = (int)((N + U_BN - 1) / U_BN);
    const int64_t ntiles = tile_count(s);
    if (ntiles * (batch ? batch->count : 1) < ctx->update256_min_tiles) return false;
    *out = s;
    *ntiles_out = ntiles;
    return true;
}

template <typename T, int ABL, bool OVW = false, bool ATOM = false, bool KEND = false>
static bool launch_update256_abl(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                                 TileShape shape, const int* info, int flags = 0, const GemmBatch* batch = nullptr) {
    TileShape s;
    int64_t tiles_per;
    if (!update256_plan<T>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, &s, &tiles_per, flags, batch)) return false;
    if (!prepare<T, ABL, OVW, ATOM, KEND>(ctx)) return false;
    const int64_t ntiles = tiles_per * (batch ? batch->count : 1);
    // one workgroup per CU; the look-ahead's free slots (two per CU in the 128 x 128 kernel's terms) become whole free CUs
    // Beside a chain launch that is already resident (chol.h: chain_wait_kernel went first) the grid still covers EVERY compute unit: the
    // workgroups that find theirs taken wait in the dispatcher until the chain's workgroup on that unit exits, then take tiles from their
    // XCD's queue like everyone else — the chain is busy a sixth of a factorisation, and its eight units used to idle for the rest of it.
    const int cus = ctx->update_late_wgs ? ctx->num_cus / 8 * 8 : (ctx->num_cus - (ctx->gemm_reserve + 1) / 2) / 8 * 8;
    const int grid = (int)std::min<int64_t>(cus, (ntiles + 7) / 8 * 8);
    if (grid <= 0) return false;
    QueueArgs qa;
    qa.use_queue = ntiles > grid;
    qa.tiles_per = tiles_per;
    qa.strideA = batch ? batch->strideA : 0;
    qa.strideB = batch ? batch->strideB : 0;
    qa.strideC = batch ? batch->strideC : 0;
    for (int x = 0; x <= 8; ++x) qa.start[x] = ntiles * x / 8;
    for (int x = 0; x < 8; ++x) {
        qa.base[x] = ctx->queue_base[x];
        if (qa.use_queue) ctx->queue_base[x] += (unsigned long long)(qa.start[x + 1] - qa.start[x]);
        qa.done_base[x] = 0;
    }
    const unsigned lds = U_NBUF * (U_BM + U_BN) * 128;
    if (ctx->attach_a) {  // profiled launch: the dispatch carries its own start / stop events (ProfScope attach mode)
        hipEvent_t ea = ctx->attach_a, eb = ctx->attach_b;
        ctx->attach_a = ctx->attach_b = nullptr;
        hipExtLaunchKernelGGL((update256_kernel<T, ABL, OVW, ATOM, KEND>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, ea, eb, 0, C, ldc, A, lda, B, ldb, M, N, K, s,
                              ctx->d_queue, qa, info);
        return true;
    }
    hipLaunchKernelGGL((update256_kernel<T, ABL, OVW, ATOM, KEND>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, C, ldc, A, lda, B, ldb, M, N, K, s, ctx->d_queue,
                       qa, info);
    return true;
}
template <typename T>
bool launch_update256(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                      TileShape shape, const int* info, int flags, const GemmBatch* batch) {
    if (flags & GEMM_KEND_COL) return launch_update256_abl<T, 0, true, false, true>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch);
    if (flags & GEMM_OVERWRITE) return launch_update256_abl<T, 0, true>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch);
    if (batch) return launch_update256_abl<T, 0, false>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch);
    if (ctx->update256_atomic) return launch_update256_abl<T, 0, false, true>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info, flags, batch);
#ifdef GPMI_TOOLS
    switch (ctx->update256_ablation) {
        case 1: return launch_update256_abl<T, 1>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info);
        case 3: return launch_update256_abl<T, 3>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info);
        case 7: return launch_update256_abl<T, 7>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info);
        case 15: return launch_update256_abl<T, 15>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info);
        case 5: return launch_update256_abl<T, 5>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info);
        default: break;
    }
#endif
    return launch_update256_abl<T, 0>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, info);
}
// The measured matrix-core ceiling (gpmi_mfma_peak, SURVEY 8(d) "builder must confirm with a micro-benchmark"): update256_kernel's OWN K loop
// with everything but the MFMAs compiled out — ABL = 15: no operand DMA after the first two slabs, no fragment reads after the first, no slab
// barrier, no epilogue — over 4096 rectangular 256 x 128 tiles with K = 2048 (16 tiles per compute unit through the same per-XCD queues).
// What it times is the instruction stream the product kernel issues (16 accumulator tiles per wave, two waves per SIMD, the loop's scalar
// bookkeeping) at the clock the chip sustains under it: the rate the real kernel would reach if memory, LDS and barriers cost nothing.
// (A plain loop of v_mfma_f64_16x16x4 on four accumulators — rounds 1-5's gpmi_mfma_peak — measured 48 TFLOP/s, BELOW the product kernel:
// not a ceiling of anything; profiles/r01_mfma_bench_instruction_ceilings.log.)
template <typename T>
int mfma_peak(gpmi_ctx* ctx, double* tflops) {
    constexpr int64_t M = 16384, N = 8192, K = 2048;
    T* A = nullptr;
    GPMI_HIP(ctx, hipMalloc(&A, (size_t)(M * K) * sizeof(T)));
    auto fail = [&](const char* what) {
        (void)hipGetLastError();
        hipFree(A);
        ctx->err = std::string("gpmi_mfma_peak: ") + what;
        return GPMI_EDEVICE;
    };
    {   // operands of the magnitude and variety of a factor panel (all-zero data would flatter the clock: MFMA power follows the toggling)
        std::vector<T> h((size_t)(M * K));
        unsigned long long st = 0x9E3779B97F4A7C15ull;
        for (auto& v : h) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            v = (T)((double)(long long)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5) * T(0.03125);
        }
        if (hipMemcpyAsync(A, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            return fail("operand upload");
    }
    if (!prepare<T, 15, false, false>(ctx)) return fail("LDS attribute");
    TileShape s{(int)(M / U_BM), (int)(N / U_BN), 0, 0, 1, 0};
    const int64_t ntiles = tile_count(s);
    const int grid = ctx->num_cus / 8 * 8;
    const unsigned lds = U_NBUF * (U_BM + U_BN) * 128;
    auto go = [&]() {
        QueueArgs qa;
        qa.use_queue = ntiles > grid;
        qa.tiles_per = ntiles;
        qa.strideA = qa.strideB = qa.strideC = 0;
        for (int x = 0; x <= 8; ++x) qa.start[x] = ntiles * x / 8;
        for (int x = 0; x < 8; ++x) {
            qa.base[x] = ctx->queue_base[x];
            if (qa.use_queue) ctx->queue_base[x] += (unsigned long long)(qa.start[x + 1] - qa.start[x]);
            qa.done_base[x] = 0;
        }
        // (C is never touched with ABL & 1; the operand panel stands in for the pointer)
        hipLaunchKernelGGL((update256_kernel<T, 15, false, false>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, A, K, (const T*)A, K, (const T*)A, K, M, N, K,
                           s, ctx->d_queue, qa, (const int*)nullptr);
    };
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail("events");
    go();  // warm-up: clocks, code object
    const int iters = 3;
    (void)hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < iters; ++i) go();
    (void)hipEventRecord(e1, ctx->stream);
    float ms = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (!ok) return fail("timing");
    hipFree(A);
    *tflops = 2.0 * (double)M * (double)N * (double)K * iters / ((double)ms * 1e-3) / 1e12;
    return GPMI_OK;
}
template int mfma_peak<double>(gpmi_ctx*, double*);
template int mfma_peak<float>(gpmi_ctx*, double*);

template <typename T>
bool update256_applies(const gpmi_ctx* ctx, const T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                       TileShape shape) {
    TileShape s;
    int64_t n;
    return update256_plan<T>(ctx, C, ldc, A, lda, B, ldb, M, N, K, shape, &s, &n);
}
template bool update256_applies<double>(const gpmi_ctx*, const double*, int64_t, const double*, int64_t, const double*, int64_t, int64_t, int64_t,
                                        int64_t, TileShape);
template bool update256_applies<float>(const gpmi_ctx*, const float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t, int64_t, int64_t,
                                       TileShape);
template bool launch_update256<double>(gpmi_ctx*, double*, int64_t, const double*, int64_t, const double*, int64_t, int64_t, int64_t, int64_t,
                                       TileShape, const int*, int, const GemmBatch*);
template bool launch_update256<float>(gpmi_ctx*, float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t, int64_t, int64_t,
                                      TileShape, const int*, int, const GemmBatch*);

}  // namespace gpmi
