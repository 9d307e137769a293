// chain.hip — ONE persistent launch per W x W diagonal super-block: Cholesky factor in place + 64 x 64 diagonal inverses + the explicit
// inverse LW = L^-1 (chol.h: what factor_diag_block + build_super_inverse did in ~15 dependent launches per 256 columns).
//
// The block is cut into 64 x 64 tiles (nb = W / 64 per side).  Every output tile is ONE task, written exactly once:
//   L(c, c)   potrf:  S = A_cc - sum_{k<c} L_ck L_ck' ;  L_cc = chol(S), Linv_c = L_cc^-1         (potf2.h, one wavefront)
//   L(i, c)   i > c:  T = A_ic - sum_{k<c} L_ik L_ck' ;  L_ic = T Linv_c'                          (left-looking: dpotrf's TRSM as a product)
//   X(i, j)   i > j:  X_ij = -Linv_i sum_{k=j}^{i-1} L_ik X_kj ,  X_jj = Linv_j                   (row i of L^-1 by forward substitution)
// Tasks are handed out by ONE device-scope counter in a fixed topological order (chain_order.h: L(0, 0), then step c lists L(c+1, c) and the
// next diagonal tile L(c+1, c+1) ahead of L(c+2 .. nb-1, c) and X(c, 0 .. c-1) — the diagonal tiles are the block's critical path),
// so a workgroup only ever waits for tasks with SMALLER indices, which are finished or held by a workgroup that is running: the launch
// makes progress with any number of resident workgroups (late or never-scheduled ones simply take no tasks; no grid barrier, no co-residency
// requirement), and every spin is bounded.  Dependencies are per-tile flags; whole finished columns of L / rows of X are tracked by
// counters so that the long accumulations over old columns run without polling.
//
// Inter-workgroup visibility (MI355X_MICROARCH.md "inter-workgroup visibility", cdna_hip_programming.md §6 G16, form R1): every tile another
// workgroup will read is stored WRITE-THROUGH (8-byte agent-scope relaxed atomics = global_store_dwordx2 sc1), every storing wave drains
// (s_waitcnt vmcnt(0)), __syncthreads(), then one lane stores the flag (sc1); readers poll the flag relaxed and read the tile with sc1 loads
// (L1 bypassed) — no fences, no L2 write-back / invalidate that would disturb the trailing update running beside the chain.
#include "common.h"
#include "chain_order.h"
#ifdef GPMI_CHAIN_TRACE
#include <string>
#include <vector>
#endif
#include "mfma.h"
#include "potf2.h"

namespace gpmi {

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ unsigned long long ld_sc1(const void* p) {
    return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(void* p, unsigned long long v) { __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
    return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_flag(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// words of the synchronisation area (zeroed by the launcher before every launch)
constexpr int CH_TASK = 0;    // next task index
constexpr int CH_ABORT = 1;   // 1: a pivot failed (tasks after it publish without computing), 2: a wait timed out
constexpr int CH_HDR = 4;
// (the word BEHIND the zeroed area — chain_started_word() — counts the workgroups of ALL chain launches that have come up: never reset)
// then: FL[nb * nb] (L tile final), FX[nb * nb] (X tile final), CL[nb] (finished tiles of L column k), CX[nb] (finished tiles of X row r)

constexpr int LD = 65;  // leading dimension of the 64 x 64 LDS tiles (as the panel kernels)

template <typename T>
struct Tile {  // a 64 x 64 tile as 8-byte words: WR per row, WPT per thread of a 256-thread workgroup
    static constexpr int WR = 64 * (int)sizeof(T) / 8;
    static constexpr int WPT = 64 * WR / 256;
    static constexpr int EPW = 8 / (int)sizeof(T);  // elements per word
};

template <typename T>
__device__ __forceinline__ void unpack(unsigned long long w, T* e);
template <>
__device__ __forceinline__ void unpack<double>(unsigned long long w, double* e) { e[0] = __longlong_as_double((long long)w); }
template <>
__device__ __forceinline__ void unpack<float>(unsigned long long w, float* e) {
    e[0] = __uint_as_float((unsigned)(w & 0xffffffffull));
    e[1] = __uint_as_float((unsigned)(w >> 32));
}
template <typename T>
__device__ __forceinline__ unsigned long long pack(const T* e);
template <>
__device__ __forceinline__ unsigned long long pack<double>(const double* e) { return (unsigned long long)__double_as_longlong(e[0]); }
template <>
__device__ __forceinline__ unsigned long long pack<float>(const float* e) {
    return (unsigned long long)__float_as_uint(e[0]) | ((unsigned long long)__float_as_uint(e[1]) << 32);
}

// Addressing of a 64 x 64 global tile (row-major, leading dimension ld elements) by a 256-thread workgroup, 8-byte words: thread t's word
// q is word (t + 256 q) of the tile = row (t / WR) + q (256 / WR), column word t % WR.  The tile's base pointer is WAVE-UNIFORM (it comes
// from the task index, which the task loop passes through readfirstlane), so base + q * row-step stays in scalar registers and each
// thread keeps ONE 32-bit byte offset for all its words (global_load ... v_off, s[base:base+1]): the 16 + 16 loads of a slab would
// otherwise hold 64 vector registers of addresses.
template <typename T>
struct TileAddr {
    unsigned voff;      // byte offset of this thread's word 0
    int64_t qstep;      // bytes between consecutive words of a thread (uniform)
    __device__ __forceinline__ TileAddr(int64_t ld) {
        constexpr int WR = Tile<T>::WR;
        const int t = (int)threadIdx.x;
        voff = (unsigned)(((int64_t)(t / WR) * ld) * (int64_t)sizeof(T) + (int64_t)(t % WR) * 8);
        qstep = (int64_t)(256 / WR) * ld * (int64_t)sizeof(T);
    }
};

// global -> registers, coalesced; COHERENT: the tile was (or may have been) written by another workgroup of this launch
template <typename T, bool COHERENT>
__device__ __forceinline__ void tile_fetch(unsigned long long (&r)[Tile<T>::WPT], const T* g, const TileAddr<T>& ad) {
#pragma unroll
    for (int q = 0; q < Tile<T>::WPT; ++q) {
        const char* p = reinterpret_cast<const char*>(g) + (int64_t)q * ad.qstep + ad.voff;
        if constexpr (COHERENT)
            r[q] = ld_sc1(p);
        else
            r[q] = *reinterpret_cast<const unsigned long long*>(p);
    }
}
// registers -> LDS tile (leading dimension LD); TRANSPOSE: buf[col][row] = tile[row][col]
template <typename T, bool TRANSPOSE>
__device__ __forceinline__ void tile_publish(T* buf, const unsigned long long (&r)[Tile<T>::WPT]) {
    constexpr int WR = Tile<T>::WR, EPW = Tile<T>::EPW;
#pragma unroll
    for (int q = 0; q < Tile<T>::WPT; ++q) {
        const int e = (int)threadIdx.x + 256 * q;
        const int row = e / WR, col = (e % WR) * EPW;
        T v[EPW];
        unpack<T>(r[q], v);
#pragma unroll
        for (int t = 0; t < EPW; ++t) {
            if constexpr (TRANSPOSE)
                buf[(col + t) * LD + row] = v[t];
            else
                buf[row * LD + col + t] = v[t];
        }
    }
}
// LDS tile -> global, coalesced; COHERENT: write-through stores (sc1) for tiles other workgroups of this launch will read.
// TRANSPOSE: global[row][col] = buf[col][row].
template <typename T, bool COHERENT, bool TRANSPOSE>
__device__ __forceinline__ void tile_store(T* g, const TileAddr<T>& ad, const T* buf) {
    constexpr int WR = Tile<T>::WR, EPW = Tile<T>::EPW;
#pragma unroll
    for (int q = 0; q < Tile<T>::WPT; ++q) {
        const int e = (int)threadIdx.x + 256 * q;
        const int row = e / WR, col = (e % WR) * EPW;
        T v[EPW];
#pragma unroll
        for (int t = 0; t < EPW; ++t) v[t] = TRANSPOSE ? buf[(col + t) * LD + row] : buf[row * LD + col + t];
        char* p = reinterpret_cast<char*>(g) + (int64_t)q * ad.qstep + ad.voff;
        if constexpr (COHERENT)
            st_sc1(p, pack<T>(v));
        else
            *reinterpret_cast<unsigned long long*>(p) = pack<T>(v);
    }
}
template <typename T>
__device__ __forceinline__ void tile_zero(T* g, const TileAddr<T>& ad) {
#pragma unroll
    for (int q = 0; q < Tile<T>::WPT; ++q) *reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(g) + (int64_t)q * ad.qstep + ad.voff) = 0ull;
}

// publish: every storing wave has drained its write-through stores; then ONE lane raises the flag and counts the tile
__device__ __forceinline__ void publish(unsigned* flag, unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        st_flag(flag, 1u);
        __hip_atomic_fetch_add((gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct ChainShared {
    int task;
    int wmL, wmX;  // columns of L [0, wmL) and rows of X [0, wmX) are complete (monotone, per workgroup)
    int abort;
};

// Wait (every wave for itself: wave-uniform, no barrier) until *flag is set.  Bounded: after ~4 s of polling the launch is declared dead
// (CH_ABORT = 2, *info = INT_MIN: gpmi_fit returns GPMI_EDEVICE) and every wait returns at once.
// ISA assumption (write side: publish(); read side: here): the producer's tile stores are `sc1` write-through stores that are acknowledged
// (vmcnt) only once they are visible at the device-coherent level, and it raises the flag after s_waitcnt vmcnt(0) + a workgroup barrier; a
// consumer wave's loads are issued and return in program order, and LLVM does not hoist a monotonic (relaxed atomic) or any other load
// across `asm volatile("" ::: "memory")` — the compiler barrier below is what keeps the tile loads BEHIND the successful poll at the source
// level; tests/test_gpu_chain.py::test_chain_stress_many_workgroup_counts is the canary for a toolchain that breaks either half.
__device__ __forceinline__ void flag_seen() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ void wait_flag(const unsigned* flag, unsigned* sync, int* info, unsigned long long* waited = nullptr) {
    if (ld_flag(flag) != 0u) {
        flag_seen();
        return;
    }
    const unsigned long long t0 = waited ? wall_clock64() : 0ull;  // (trace builds only: nullptr is a compile-time constant in the product)
    for (unsigned spins = 0;; ++spins) {
        __builtin_amdgcn_s_sleep(4);
        if (ld_flag(flag) != 0u) break;
        if ((spins & 63u) == 63u) {
            if (ld_flag(sync + CH_ABORT) != 0u) break;
            if (spins > (1u << 22)) {
                // (a pivot failure recorded meanwhile stays: dpotrf's info outranks the timeout)
                if ((threadIdx.x & 63) == 0 && ld_flag(sync + CH_ABORT) == 0u) {
                    st_flag(sync + CH_ABORT, 2u);
                    atomicCAS(info, 0, (int)0x80000000);
                }
                break;
            }
        }
    }
    flag_seen();
    if (waited) *waited += wall_clock64() - t0;
}

// two flags at once: both loads are in flight together (one round trip when both are already set)
__device__ __forceinline__ void wait_flags2(const unsigned* fa, const unsigned* fb, unsigned* sync, int* info, unsigned long long* waited = nullptr) {
    const unsigned a = ld_flag(fa), b = ld_flag(fb);
    if (a != 0u && b != 0u) {
        flag_seen();
        return;
    }
    if (a == 0u) wait_flag(fa, sync, info, waited);
    if (b == 0u) wait_flag(fb, sync, info, waited);
}

template <typename T>
struct ChainArgs {
    T* A;           // the W x W block (row-major), factored in place
    int64_t ld;
    int nb;         // W / 64
    T* linv;        // nb x (64 x 64): inverses of the diagonal tiles
    T* invdiag;     // W reciprocals of the diagonal of L
    T* LW;          // explicit inverse (W x W, leading dimension wld; strict upper part zeroed) or nullptr: factor only
    int64_t wld;
    int* info;
    int64_t pivot_base;
    unsigned* sync;
    unsigned* started;  // monotonic count of chain workgroups that have started (chain_wait_kernel)
#ifdef GPMI_CHAIN_TRACE
    unsigned long long* trace;  // tools/chain_trace.py: 8 words per task (100 MHz clock marks), or nullptr
#endif
};

// Task timeline of tools/chain_trace.py — compiled in only with -DGPMI_CHAIN_TRACE (tools/build_chain_trace.sh; never in libgpmi.so):
// CH_TR(slot) stores the 100 MHz clock (s_memrealtime: ~1 us each — only at task phase boundaries and in the slow path of a wait).
#ifdef GPMI_CHAIN_TRACE
#define CH_TR(slot)                                                                      \
    do {                                                                                 \
        if (a.trace && threadIdx.x == 0) a.trace[(int64_t)t * 8 + (slot)] = wall_clock64(); \
    } while (0)
#define CH_TW , &tr_wait  // the flag waits add the time of their SLOW path (flag not yet set) to the task's wait total
#define CH_TR_END()                                                                                   \
    do {                                                                                              \
        if (a.trace && threadIdx.x == 0) {                                                            \
            a.trace[(int64_t)t * 8 + 5] = tr_wait;                                                    \
            a.trace[(int64_t)t * 8 + 6] = ((unsigned long long)blockIdx.x << 8) | (unsigned long long)(__smid() & 0xff); \
            a.trace[(int64_t)t * 8 + 7] = ((unsigned long long)is_x << 16) | ((unsigned long long)i << 8) | (unsigned long long)j; \
        }                                                                                             \
    } while (0)
#else
#define CH_TR(slot) ((void)0)
#define CH_TW
#define CH_TR_END() ((void)0)
#endif

// acc[mi][ni] += a[rows mi] b[rows ni]' over one 64-deep slab: each wave's 32 x 32 quadrant as 2 x 2 MFMA tiles.  Per k-step the two A
// and the two B fragments are read ONCE and feed four MFMAs (mma16_nt per tile would read each fragment twice), and the 16 steps are
// unrolled so that the LDS reads run ahead of the MFMAs: one wave per SIMD has nothing else to hide their latency behind.  Every tile
// still accumulates k = 0, 4, 8, ... in order: the same bits as four mma16_nt calls.
template <typename T>
__device__ __forceinline__ void product64(typename Mfma<T>::Acc (&acc)[2][2], const T* a, const T* b) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const T* ap = a + (wm * 32 + (lane & 15)) * LD + (lane >> 4);  // fragment element of tile mi = 0; mi = 1: 16 rows further down
    const T* bp = b + (wn * 32 + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < 64; kk += 4) {
        const T a0 = ap[kk], a1 = ap[16 * LD + kk];
        T b0[4], b1[4];
        Mfma<T>::rotations(bp[kk], b0);
        Mfma<T>::rotations(bp[16 * LD + kk], b1);
        Mfma<T>::mma(a0, b0, acc[0][0]);
        Mfma<T>::mma(a0, b1, acc[0][1]);
        Mfma<T>::mma(a1, b0, acc[1][0]);
        Mfma<T>::mma(a1, b1, acc[1][1]);
    }
}
// accumulators -> LDS tile (leading dimension LD); TRANSPOSE: buf[col][row]; every element multiplied by sgn (1 or -1)
template <typename T, bool TRANSPOSE>
__device__ __forceinline__ void acc_to_lds(T* buf, const typename Mfma<T>::Acc (&acc)[2][2], T sgn) {
    using MF = Mfma<T>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv >> 1, wn = wv & 1;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                const T v = sgn * acc_get<T>(acc[mi][ni], r);
                if constexpr (TRANSPOSE)
                    buf[col * LD + row] = v;
                else
                    buf[row * LD + col] = v;
            }
}

// The workgroup's LDS pool: two 64 x 65 tile buffers (the K loops' operands; potf2_wg's S and XT) + potf2_wg's small blocks.  Two workgroups
// of this size share a compute unit's 160 KiB (the half-CU slots of the masked chain stream).
__shared__ double g_pool[POTF2_WG_POOL];
static_assert(sizeof(double) * POTF2_WG_POOL + 64 <= 80 * 1024, "two chain workgroups per compute unit");

template <typename T>
__global__ __launch_bounds__(256, 2) void chain_block_kernel(ChainArgs<T> a) {
    using MF = Mfma<T>;
    using Acc = typename MF::Acc;
    constexpr int WPT = Tile<T>::WPT;
    __builtin_amdgcn_s_setprio(3);  // beside the trailing update's waves (see diag64_kernel)
    T* const pool = reinterpret_cast<T*>(g_pool);
    __shared__ ChainShared sh;
    T* const buf1 = pool;            // diag64_body's S
    T* const buf2 = pool + 64 * LD;  // diag64_body's XT
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int nb = a.nb;
    const bool inv = a.LW != nullptr;
    const int ntasks = chain_ntasks(nb, inv);
    unsigned* const FL = a.sync + CH_HDR;
    unsigned* const FX = FL + nb * nb;
    unsigned* const CL = FX + nb * nb;
    unsigned* const CX = CL + nb;
    if (tid == 0) {
        __hip_atomic_fetch_add((gu32*)a.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // "this workgroup holds a compute unit now"
        sh.wmL = 0;
        sh.wmX = 1;  // row 0 of X has no off-diagonal tile
        sh.abort = 0;
        sh.task = *a.info;  // an earlier factorisation step failed: nothing to do.  Read ONCE per workgroup: another workgroup of this very
                            // launch may set *info while this one starts, and the waves of a workgroup must not disagree
    }
    __syncthreads();
    if (sh.task != 0) return;
    __syncthreads();

    for (;;) {
        // ---- next task + refresh of the completed-column / completed-row marks (one round trip, wave 0) ----
        if (wv == 0) {
            int t = 0;
            if (lane == 0) t = (int)__hip_atomic_fetch_add((gu32*)(a.sync + CH_TASK), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int wl = sh.wmL, wx = sh.wmX;
            // lanes 0..31 look at L columns wl .. wl+31, lanes 32..63 at X rows wx .. wx+31
            const bool isx = lane >= 32;
            const int idx = (isx ? wx : wl) + (lane & 31);
            bool done = false;
            if (idx < nb) done = ld_flag((isx ? CX : CL) + idx) >= (unsigned)(isx ? idx : nb - idx);
            const unsigned long long m = __ballot(done);
            const unsigned lo = (unsigned)(m & 0xffffffffull), hi = (unsigned)(m >> 32);
            const int nl = lo == 0xffffffffu ? 32 : __builtin_ctz(~lo);  // leading run of completed columns
            const int nx = hi == 0xffffffffu ? 32 : __builtin_ctz(~hi);
            const unsigned ab = ld_flag(a.sync + CH_ABORT);
            if (lane == 0) {
                sh.task = t;
                sh.wmL = wl + nl;
                sh.wmX = wx + nx;
                sh.abort = (int)ab;
            }
        }
        __syncthreads();
        // (readfirstlane: the values are the same in every lane, and now the compiler knows — everything derived from the task index,
        //  the tile base pointers above all, lives in scalar registers)
        const int t = __builtin_amdgcn_readfirstlane(sh.task);
        const int wmL_raw = __builtin_amdgcn_readfirstlane(sh.wmL), wmX_raw = __builtin_amdgcn_readfirstlane(sh.wmX);
        const int wmL = wmL_raw < nb ? wmL_raw : nb, wmX = wmX_raw < nb ? wmX_raw : nb;
        const bool aborted = __builtin_amdgcn_readfirstlane(sh.abort) != 0;
        __syncthreads();  // (sh is rewritten at the top of the next iteration)
        if (t >= ntasks) break;
        // ---- decode (chain_order.h: the diagonal tile of the next step first, then the column, then the row of X; topological — tests/test_chain_order.py) ----
        const ChainTask task = chain_decode(t, nb, inv);
        const bool is_x = task.is_x != 0;
        const int i = task.i, j = task.j;  // output tile row / column
        const int c = is_x ? i : j;        // the step
        unsigned* const my_flag = (is_x ? FX : FL) + i * nb + j;
        unsigned* const my_count = is_x ? CX + i : CL + j;
#ifdef GPMI_CHAIN_TRACE
        unsigned long long tr_wait = 0;
#endif
        CH_TR(0);
        if (aborted) {  // a pivot failed / a wait timed out: publish so that nobody waits, compute nothing
            publish(my_flag, my_count);
            continue;
        }

        Acc acc[2][2];
        const TileAddr<T> adA(a.ld), adW(a.wld), adI(64);

        if (!is_x) {
            // =============================== L(i, c) ===============================
            const bool diag = i == c;
            // the accumulators start at MINUS the original entries of the tile (plain loads: written before this launch), so that after the
            // K loop they hold  sum_k L_ik L_ck' - A_ic = -T  and no second copy of the tile occupies registers meanwhile
            {
                const T* At = a.A + (int64_t)(i * 64) * a.ld + c * 64;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                            acc[mi][ni].v[r] = -At[(int64_t)row * a.ld + col];
                        }
            }
            if (inv && !diag) tile_zero<T>(a.LW + (int64_t)(c * 64) * a.wld + i * 64, adW);  // the mirrored (strictly upper) tile of LW
            // ---- acc += sum_{k < c} L_ik L_ck'  (one operand when i == c), next slab in flight while this one is multiplied ----
            unsigned long long ra[WPT], rb[WPT];
            auto fetch = [&](int k) {
                if (k >= wmL) {
                    if (diag)
                        wait_flag(FL + i * nb + k, a.sync, a.info CH_TW);
                    else
                        wait_flags2(FL + i * nb + k, FL + c * nb + k, a.sync, a.info CH_TW);
                }
                tile_fetch<T, true>(ra, a.A + (int64_t)(i * 64) * a.ld + k * 64, adA);
                if (!diag) tile_fetch<T, true>(rb, a.A + (int64_t)(c * 64) * a.ld + k * 64, adA);
            };
            if (c > 0) fetch(0);
            for (int k = 0; k < c; ++k) {
                tile_publish<T, false>(buf1, ra);
                if (!diag) tile_publish<T, false>(buf2, rb);
                __syncthreads();
                if (k == 0) CH_TR(1);
                // the next slab's loads fly under this slab's product when no wait stands before them (old, complete columns); at the front
                // the product goes first — it needs nothing that is still being computed — and the wait after it
                const bool early = k + 1 < c && k + 1 < wmL;
                if (early) fetch(k + 1);
                product64<T>(acc, buf1, diag ? buf1 : buf2);
                if (!early && k + 1 < c) fetch(k + 1);
                __syncthreads();
            }
            acc_to_lds<T, false>(buf1, acc, T(-1));  // T = A_ic - sum
            CH_TR(2);
            if (diag) {
                // (a pivot that failed in an earlier diagonal tile while this task was already under way: what is in S is garbage — no potf2.
                //  Read by ONE thread: the flag may flip between two waves' loads, and the whole workgroup must take the same branch)
                if (tid == 0) sh.abort = (int)ld_flag(a.sync + CH_ABORT);
                __syncthreads();
                int fail = 0;
                if (sh.abort == 0) fail = potf2_wg<T>(a.invdiag + c * 64, a.info, a.pivot_base + (int64_t)c * 64, pool);  // the whole workgroup (potf2.h)
                if (tid == 0 && fail) st_flag(a.sync + CH_ABORT, 1u);
                __syncthreads();
                CH_TR(3);
                // (on failure the stores below write garbage that nobody uses: *info is set, every later kernel returns at once)
                T* Lcc = a.A + (int64_t)(c * 64) * a.ld + c * 64;
                tile_store<T, false, false>(Lcc, adA, buf1);                                   // L_cc, strict upper part zero
                tile_store<T, true, true>(a.linv + (int64_t)c * 64 * 64, adI, buf2);           // Linv_c = XT'
                if (inv) tile_store<T, false, true>(a.LW + (int64_t)(c * 64) * a.wld + c * 64, adW, buf2);   // X_cc
                publish(my_flag, my_count);
                CH_TR(4);
                CH_TR_END();
            } else {
                // ---- L_ic = T Linv_c' ----
                wait_flag(FL + c * nb + c, a.sync, a.info CH_TW);
                tile_fetch<T, true>(ra, a.linv + (int64_t)c * 64 * 64, adI);
                tile_publish<T, false>(buf2, ra);
                __syncthreads();
                CH_TR(3);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc_zero<T>(acc[mi][ni]);
                product64<T>(acc, buf1, buf2);
                __syncthreads();
                acc_to_lds<T, false>(buf1, acc, T(1));
                __syncthreads();
                tile_store<T, true, false>(a.A + (int64_t)(i * 64) * a.ld + c * 64, adA, buf1);
                publish(my_flag, my_count);
                CH_TR(4);
                CH_TR_END();
            }
        } else {
            // =============================== X(i, j), i > j ===============================
            // acc = sum_{k = j}^{i-1} L_ik X_kj :  A operand L_ik (rows m, k contiguous), B operand [n][k] = X_kj[k][n] (transposed on its way into LDS)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc_zero<T>(acc[mi][ni]);
            unsigned long long ra[WPT], rb[WPT];
            auto fetch = [&](int k) {
                // the two operands' flags in one round trip (an operand known complete polls the task's own, always-set... no: its own flag
                // is not set yet — a complete operand simply repeats the other's flag)
                const unsigned* fa = FL + i * nb + k;                                   // L_ik
                const unsigned* fb = k == j ? FL + j * nb + j : FX + k * nb + j;        // X_jj = Linv_j, or X_kj
                const bool need_a = k >= wmL, need_b = k == j ? j >= wmL : k >= wmX;
                if (need_a || need_b) wait_flags2(need_a ? fa : fb, need_b ? fb : fa, a.sync, a.info CH_TW);
                tile_fetch<T, true>(ra, a.A + (int64_t)(i * 64) * a.ld + k * 64, adA);
                if (k == j)
                    tile_fetch<T, true>(rb, a.linv + (int64_t)j * 64 * 64, adI);
                else
                    tile_fetch<T, true>(rb, a.LW + (int64_t)(k * 64) * a.wld + j * 64, adW);
            };
            fetch(j);
            for (int k = j; k < i; ++k) {
                tile_publish<T, false>(buf1, ra);
                tile_publish<T, true>(buf2, rb);
                __syncthreads();
                if (k == j) CH_TR(1);
                const bool early = k + 1 < i && k + 1 < wmL && k + 1 < wmX;
                if (early) fetch(k + 1);
                product64<T>(acc, buf1, buf2);
                if (!early && k + 1 < i) fetch(k + 1);
                __syncthreads();
            }
            // X_ij = -Linv_i W :  A operand Linv_i (rows m, k contiguous), B operand [n][k] = W[k][n]
            acc_to_lds<T, true>(buf2, acc, T(-1));
            CH_TR(2);
            wait_flag(FL + i * nb + i, a.sync, a.info CH_TW);
            tile_fetch<T, true>(ra, a.linv + (int64_t)i * 64 * 64, adI);
            tile_publish<T, false>(buf1, ra);
            __syncthreads();
            CH_TR(3);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc_zero<T>(acc[mi][ni]);
            product64<T>(acc, buf1, buf2);
            __syncthreads();
            acc_to_lds<T, false>(buf1, acc, T(1));
            __syncthreads();
            tile_store<T, true, false>(a.LW + (int64_t)(i * 64) * a.wld + j * 64, adW, buf1);
            publish(my_flag, my_count);
            CH_TR(4);
            CH_TR_END();
        }
    }
}

// One wavefront on the UPDATE's stream, in front of a trailing update that a chain launch on another (unmasked) stream is meant to run
// beside: returns when all `expect` chain workgroups launched so far have come up (or after ~100 us).  Without it the two launches
// race for compute units, and the dispatcher queues a chain workgroup on a shader engine whose compute units the update has just taken —
// where it stays until the update ends, although another engine of the same XCD has the free one: measured 2 of 8 chain workgroups resident
// on average (14 ms per 2048 block instead of 4; hidden at N = 50 000, exposed at N = 20 000 — and WHICH workgroups lose the lottery moved
// with the instrumentation of the run: 61 ms per step un-instrumented, 82 with events on the update).  With the chain placed first on the
// idle chip it is the update whose last workgroups may wait for the chain's few milliseconds: a persistent kernel with a tile queue does
// not care.
// `ticks`: how long to wait at most (100 MHz).  ~100 us when the update leaves the chain's compute units free anyway (a late chain then only
// starts late); 5 ms when the update's grid covers EVERY unit (update256.hip update_late_wgs): a chain that lost the race would wait for
// the whole update — measured on the blocked handle's one-rank path, where the chain's launch sits behind a cross-stream event and a
// memset: 9.2 ms per step in the chain phase instead of 1.3 (profiles/r06_c_*).
__global__ void chain_wait_kernel(const unsigned* started, unsigned expect, long long ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();  // 100 MHz
    while ((int)(ld_flag(started) - expect) < 0) {
        if (wall_clock64() - t0 > ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

}  // namespace

// bytes of the synchronisation area for blocks of up to nb_max tiles per side (+ the never-reset `started` word behind the zeroed part)
static int64_t chain_zeroed_words(int nb) { return (int64_t)CH_HDR + 2 * (int64_t)nb * nb + 2 * nb; }
int64_t chain_sync_bytes(int nb_max) { return (chain_zeroed_words(nb_max) + 16) * 4; }

void launch_chain_wait(gpmi_ctx* ctx) {
    if (!ctx->chain_sync || !ctx->chain_wait_pending) return;
    ctx->chain_wait_pending = false;
    const unsigned* started = (const unsigned*)ctx->chain_sync + chain_zeroed_words(ctx->chain_nb_max);
    hipLaunchKernelGGL(chain_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, started, ctx->chain_started_expect,
                       (long long)(ctx->update_late_wgs ? 500000 : 10000));
}

template <typename T>
bool launch_chain_block(gpmi_ctx* ctx, T* A, int64_t ld, int64_t w, T* linv, T* invdiag, T* LW, int64_t wld, int* info, int64_t pivot_base) {
    if (!ctx->chain_kernel || ctx->refine_solves || w % 64 != 0 || w <= 0 || w > (int64_t)ctx->chain_nb_max * 64 || !ctx->chain_sync) return false;
    const int nb = (int)(w / 64);
    unsigned* sync = (unsigned*)ctx->chain_sync;
    (void)hipMemsetAsync(sync, 0, (size_t)chain_zeroed_words(nb) * 4, ctx->stream);
    ChainArgs<T> a{A, ld, nb, linv, invdiag, LW, wld, info, pivot_base, sync, sync + chain_zeroed_words(ctx->chain_nb_max)};
#ifdef GPMI_CHAIN_TRACE
    static unsigned long long* tr_dev = nullptr;
    static long long tr_left = 0, tr_skip = 0;  // (re)read whenever the file variable changes (the tool sets it after each warm-up fit)
    static std::string tr_name;
    const char* tr_file = getenv("GPMI_CHAIN_TRACE_FILE");
    if (tr_file && tr_name != tr_file) {
        tr_name = tr_file;
        tr_left = getenv("GPMI_CHAIN_TRACE_LAUNCHES") ? atoll(getenv("GPMI_CHAIN_TRACE_LAUNCHES")) : 1;
        tr_skip = getenv("GPMI_CHAIN_TRACE_SKIP") ? atoll(getenv("GPMI_CHAIN_TRACE_SKIP")) : 0;
    }
    const int64_t tr_words = (int64_t)chain_ntasks(nb, LW != nullptr) * 8;
    if (!tr_dev) (void)hipMalloc(&tr_dev, (size_t)chain_ntasks(32, true) * 8 * sizeof(unsigned long long));
    const bool tr_on = tr_file && tr_dev && tr_left > 0 && (tr_skip-- <= 0);
    a.trace = tr_on ? tr_dev : nullptr;
    if (tr_on) (void)hipMemsetAsync(tr_dev, 0, (size_t)tr_words * 8, ctx->stream);
#endif
    // workgroups: what fits beside the trailing update (chol.h beside_update: the reserved compute units / free slots), otherwise enough
    // for the tasks of one step (nb) with one workgroup per compute unit
    int64_t g = ctx->beside_update ? side_slots(ctx) : std::max<int64_t>(8, std::min<int64_t>(2 * nb, ctx->chain_wgs_max));
    if (ctx->beside_update && ctx->side_one_per_xcd && ctx->chain_wide_ok) g = std::max<int64_t>(g, ctx->chain_beside_wgs);
    if (ctx->chain_wgs > 0) g = ctx->chain_wgs;
    if (g < 1) g = 1;
    const double flops = (LW ? 2.0 : 1.0) * (double)w * (double)w * (double)w / 3.0;
    ProfScope ps(ctx, GPMI_PROF_PANEL, flops, 0.0, false, /*chain_kernel=*/false);
    hipLaunchKernelGGL(chain_block_kernel<T>, dim3((unsigned)g), dim3(256), 0, ctx->stream, a);
#ifdef GPMI_CHAIN_TRACE
    if (tr_on) {
        --tr_left;
        std::vector<unsigned long long> h((size_t)tr_words);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h.data(), tr_dev, (size_t)tr_words * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(tr_file, "a")) {
            unsigned long long pm[8] = {0};
            (void)hipMemcpyFromSymbol(pm, HIP_SYMBOL(g_potf2_marks), sizeof(pm));
            fprintf(f, "launch nb %d wgs %lld ld %lld inverse %d beside %d elem %d potf2 %llu %llu %llu %llu\n", nb, (long long)g, (long long)ld, LW ? 1 : 0,
                    ctx->beside_update ? 1 : 0, (int)sizeof(T), pm[1] - pm[0], pm[2] - pm[1], pm[3] - pm[2], pm[4] - pm[3]);
            for (int64_t q = 0; q < tr_words / 8; ++q) {
                fprintf(f, "%lld", (long long)q);
                for (int m = 0; m < 8; ++m) fprintf(f, " %llu", h[(size_t)(q * 8 + m)]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
#endif
    ctx->chain_started_expect += (unsigned)g;
    // beside an update on an UNMASKED stream the update's launch first waits for these workgroups to be placed (chain_wait_kernel)
    ctx->chain_wait_pending = ctx->beside_update && ctx->la_mode != 1;
    return true;
}
template bool launch_chain_block<double>(gpmi_ctx*, double*, int64_t, int64_t, double*, double*, double*, int64_t, int*, int64_t);
template bool launch_chain_block<float>(gpmi_ctx*, float*, int64_t, int64_t, float*, float*, float*, int64_t, int*, int64_t);

}  // namespace gpmi
