// common.h — shared declarations of libgpmi.so (gfx950 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/gpmi.h"
#include "tile_order.h"

namespace gpmi {

// ------------------------------------------------------------------------------------------
// Blocking constants of the factorisation
// ------------------------------------------------------------------------------------------
constexpr int IB = 64;        // inner block: potf2 / trsm granularity; n is padded to a multiple
constexpr int NB = 256;       // outer block: K of the MFMA trailing update
constexpr int GEMM_BM = 128;  // trailing-update tile
constexpr int GEMM_BN = 128;
constexpr int SUPER = 8;      // tiles per super-tile edge (XCD-aware ordering)

// ------------------------------------------------------------------------------------------
// Device-side kernel program: the gpmi_kernel postfix descriptor, digested on the host.
// Every leaf carries a DENSE weight vector over all d input rows (0 on rows a Masked
// wrapper hides, il2[z] on active ARD rows, 1 on active iso rows), so the device loop
// is the same for Masked / ARD / iso leaves:  r = sum_k w_k (x_k - y_k)^2.
// ------------------------------------------------------------------------------------------
struct DevLeaf {
    int32_t op;     // gpmi_op
    int32_t woff;   // offset of this leaf's d weights inside DevProgram::w (leaf ops only)
    double s2;      // signal variance
    double p0;      // reciprocal constant: 1/l2 (SE iso), 1/l (Matern iso), 1/(2 a l2) (RQ iso), 0.5/a (RQ ard), 1 otherwise
    double p1;      // RQ: alpha
    // gradient path (update_dmll!): tree structure and where this node's log-parameters sit in get_params order
    int32_t left, right;  // children of a SUM / PROD node (indices into leaf[])
    int32_t poff;         // first hyper-parameter slot of this leaf
    int32_t nd;           // number of active input rows (ARD leaves own nd length scales, slots poff .. poff+nd-1)
};
struct DevProgram {
    int32_t n_ops;
    int32_t d;
    int32_t has_noise_leaf;
    int32_t fast_class;  // >= 0: a multi-leaf program the specialised interior-tile kernel takes (cov.hip cov_multi_kernel): stationary /
                         // Const / Noise leaves only, evaluation depth <= 3; bit 0 = an RQ leaf (pow), bit 1 = a Noise leaf.  -1 otherwise
    double kdiag;  // k(x,x): the program evaluated with every leaf at r = 0
    int32_t n_hyp;  // total number of kernel hyper-parameters (get_params order)
    int32_t pad2_;
    DevLeaf leaf[GPMI_MAX_OPS];
    // The per-leaf tables FOLLOW the struct in the same allocation (sized at run time: any input dimension d — the reference's
    // distance loops take any `dim`, src/kernels/distance.jl:41-106):  double w[w_count]  (leaf l's d weights at w[leaf[l].woff ..]),
    // then  int32_t pmap[w_count]  (ARD leaves: slot offset, within the leaf, of input row k, or -1).
    int64_t w_count;
    __host__ __device__ const double* wtab() const { return reinterpret_cast<const double*>(this + 1); }
    __host__ __device__ double* wtab() { return reinterpret_cast<double*>(this + 1); }
    __host__ __device__ const int32_t* pmtab() const { return reinterpret_cast<const int32_t*>(wtab() + w_count); }
    __host__ __device__ int32_t* pmtab() { return reinterpret_cast<int32_t*>(wtab() + w_count); }
};
static_assert(sizeof(DevProgram) % 8 == 0, "the weight table behind DevProgram must be 8-byte aligned");
inline int64_t program_bytes(int64_t w_count) { return (int64_t)sizeof(DevProgram) + w_count * 12; }

// digest + validate into a buffer of program_bytes(leaves * d) (grown as needed); returns GPMI_OK / GPMI_EARG and fills err
int digest_kernel(const gpmi_kernel* k, int d, std::vector<unsigned char>* buf, std::string* err);

// ------------------------------------------------------------------------------------------
// profiling (event pairs around launches, per class)
// ------------------------------------------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    int cls;
    double work;
    double bytes;
    bool attached;  // start / stop events of a dispatch (hipExtLaunchKernelGGL): they pin the kernel command, see drain_profile
};

}  // namespace gpmi

namespace gpmi {
struct SuperPart {
    int64_t ks, w, off;  // columns [ks, ks + w), offset (elements) of the w x w inverse in the store
};
}  // namespace gpmi

struct gpmi_ctx {
    int device = 0;
    int cu_part = 0;        // 0: the whole device; 1 / 2: every stream of this context is confined to half (p - 1) of every XCD (gpmi_ctx_create)
    void* group = nullptr;  // non-null: member of an in-process device group (gpmi_ctx_create with n_devices > 1; dev_hip.hip)
    int group_rank = 0;     // 0 = the primary (the handle the caller holds)
    hipStream_t stream = nullptr;
    std::string err;
    gpmi::DevProgram* d_prog = nullptr;  // device copy of the current kernel program (header + weight tables, grown to fit)
    gpmi::DevProgram* h_prog = nullptr;  // pinned staging, same size
    int64_t prog_cap = 0;                // bytes allocated for each of the two
    std::vector<unsigned char> prog_buf; // digest scratch
    int* d_info = nullptr;               // not-PD flag (1-based pivot)
    double* d_scal = nullptr;            // small double outputs (mll, logdet, dot)
    double* h_scal = nullptr;            // pinned
    unsigned long long* d_queue = nullptr;  // 8 per-XCD tile-queue words, 64 B apart (never reset)
    unsigned long long queue_base[8] = {0}; // value of each word when the next launch starts
    unsigned long long done_base[8] = {0};      // 'tiles finished' words (queue word + 1), GEMM_PHASE_LOCK launches only
    unsigned long long done_base_side[8] = {0};
    int64_t phase_lock_min_k = 0;               // trailing updates with K >= this are phase-locked (0 = never; GPMI_PHASE_LOCK)
    unsigned long long* d_queue_side = nullptr;   // a second set for persistent launches beside the update (side stream):
    unsigned long long queue_base_side[8] = {0};  // the two run concurrently and must not share queue words
    bool refine_default = false;         // GPMI_REFINE=1: refine everywhere (bring-up / accuracy studies)
    bool refine_solves = false;          // rows64: one refinement step on every product with a stored inverse
                                         // (set for factorisations regularised only by a nugget; panel.hip)
    int gemm_ni = 0;                     // 0: per launch (gemm.hip); 4: always 128 x 128 tiles, 2 workgroups per CU;
                                         // 2: always 128 x 64 tiles, 3 per CU (GPMI_GEMM_NI)
    int gemm_wgs_per_cu = 2;             // tools: GPMI_GEMM_WGS=1 runs one workgroup per CU
    int num_cus = 256;                   // CUs the persistent GEMM sizes its grid for on the CURRENT stream
    // look-ahead Cholesky (api.hip: cholesky_lower): the next panel's serial chain runs on side_stream under the
    // trailing update, which leaves lookahead_slots workgroup slots free (gemm_reserve is set around that launch)
    hipStream_t side_stream = nullptr;
    // round 3: WHOLE compute units for the look-ahead chain.  In that mode the chain runs on side_masked (a CU mask of
    // `reserved_cus` CUs, one per XCD) and the update it hides under on upd_stream (the complementary mask; event hop from / to
    // the main stream), so the chain's single-wave kernels no longer share a SIMD with GEMM waves (diag64: 26 us alone, ~130 us
    // beside a GEMM workgroup on the same CU).  Measured (profiles/r03_a_cumask_ab.log): whole CUs cost the update 4.3 % (589 ->
    // 615 ms at N = 50 000) where round 2's free workgroup slots cost ~1.5 %, and shorten the chain 5x: N = 20 000 goes from 78.3
    // to 73.0 ms per step, N = 50 000 from 710 to 732.  So the mode is chosen PER FACTORISATION (whole CUs below whole_cus_below
    // rows; set_lookahead_mode).  All streams are created with the context, in a fixed order (api.hip create_lookahead_streams).
    hipStream_t upd_stream = nullptr;
    hipStream_t side_masked = nullptr;
    int reserved_cus = 0;
    int la_mode = -1;                // -1 none yet, 0 free slots (side_stream), 1 whole CUs (side_masked + upd_stream)
    bool mask_ok = false;            // CU-masked streams are available (256 CUs, GPMI_CUMASK != 0, creation has not failed)
    int64_t whole_cus_below = 12288; // factorisations of fewer rows reserve whole CUs for the chain (the blocked path decides per
                                     // step: its fixed-width blocks leave a long chain-bound tail, blocked.cpp).  32768 while the chain was a
                                     // string of launches; with the persistent chain kernel free slots win from ~20 000 rows on
                                     // (N = 28 000: 140.2 against 144.0 ms; N = 12 288: 21.6 against 21.1; profiles/r05_d_knob_sweeps_fine.log); 18432 in round 5.
                                     // Round 6 (the update's grid covers every CU and the chain, placed first, takes 16 workgroups): free slots win from
                                     // ~13 000 rows on — N = 18 000 46.5 -> 45.2 ms, N = 16 000 35.4 -> 34.5, N = 12 000 18.67 against 18.77 (profiles/r06_o_*)
    int64_t lookahead_min_tiles_masked = 288;  // = a 3072-row trailing matrix at K = 256: the fast chain hides under shorter updates
    int lookahead_slots = 0;
    int64_t lookahead_min_tiles = 650;   // update length (in 128 x 128 x 256 tile products) below which the serial order is
                                         // faster (update < chain); 650 = the lower tiles of a 4608-row trailing matrix
    // two-level factorisation (chol.h): a super-panel of 512 / 1024 / 2048 columns is used while the remaining matrix has
    // at least super_min[0 / 1 / 2] rows (0 = never)
    // {8192, 12288, 24576} in rounds 2-4 (profiles/r02_super_sweep.log); re-swept with the persistent chain kernel, whose blocks cost a
    // third of the multi-launch chain's: wider panels now pay at much smaller trailing sizes (N = 20 000: 66.0 -> 61.3 ms per fit + predict,
    // N = 50 000: 689.0 -> 683.9; profiles/r05_c_knob_sweeps_with_the_chain_kernel.log, r05_d_knob_sweeps_fine.log)
    // round 6 (potf2_wg, sixteen chain workgroups beside a full-grid update): 1024-wide panels from 4096 rows (was 6144): N = 12 000 19.15 -> 18.95 ms,
    // N = 20 000 58.75 -> 58.63, N = 50 000 flat; everything else of two 15-configuration sweeps within noise (profiles/r06_i_*, r06_n_*)
    int64_t super_min[3] = {2048, 4096, 13312};
    int64_t super_wide = 0;              // > 0: the width of the widest super-panel class instead of 2048 (a multiple of 256 up to 2048; GPMI_SUPER_W: sweep hook)
    // scratch of the two-level factorisation (grown on demand, chol.h): the explicit inverse of the current W x W diagonal
    // super-block (sup_lw, leading dimension sup_wld) and its transpose, the packed 256-inverses it is built from, an
    // (W/2)^2 product buffer, and the out-of-place image of the solved rows below (rows x W)
    void* sup_lw = nullptr;  int64_t sup_lw_cap = 0;
    void* sup_lwt = nullptr; int64_t sup_lwt_cap = 0;
    void* sup_l256 = nullptr; int64_t sup_l256_cap = 0;
    void* sup_ut = nullptr;  int64_t sup_ut_cap = 0;
    void* cov_scaled = nullptr;  int64_t cov_scaled_cap = 0;  // cov.hip: pre-scaled, zero-padded copies of the two input blocks of a single-leaf cov!
    void* sup_s = nullptr;   int64_t sup_s_cap = 0;
    int64_t sup_wld = 0;
    int64_t grad_chunk = 2048;           // K chunk of the gradient's K^-1 = L^-T L^-1 accumulation (GPMI_GRAD_CHUNK; 0 = one product)
    int super_inverse = 1;               // rows below a super-panel through its explicit inverse (GPMI_SUPER_INV=0: NB-block substitution)
    int whiten_by_super_inverse = 1;     // predict / gradient whitening through the stored super-block inverses (GPMI_WHITEN_INV=0: NB blocks)
    int64_t whiten_super = 1024;         // super-block width of whiten_rows_inv (GPMI_WHITEN_SUPER; 256 = one level)
    int gemm_reserve = 0;
    bool update_full_grid = true;   // dense look-ahead: the 256 x 128 update is launched with a workgroup for EVERY compute unit; the ones whose unit the chain
                                     // holds start when the chain's workgroup there exits and pull what is left of their XCD's queue (GPMI_UPDATE_FULL_GRID=0: off)
    int update256_kend = 0;              // GPMI_UPDATE256_KEND=1: tall panel solves X LW' (GEMM_KEND_COL) in 256 x 128 tiles too (update256_kernel's KEND instantiation).
                                         // Built, parity-green (tests/test_gpu_twolevel.py) and measured NEUTRAL: N = 50 000 667.3 / 667.3 ms, C2 59.3 / 59.5 (profiles/r06_g_*): off
    bool kend_heavy_first = true;    // GEMM_KEND_COL launches hand their long tiles out first (GPMI_KEND_HEAVY_FIRST=0: columns in ascending order, rounds 1-5)
    bool update_late_wgs = false;    // set around that launch (chol.h main_update_beside_chain)
    int update256 = 1;                   // big trailing updates in 256 x 128 tiles, one 512-thread workgroup per CU (update256.hip;
                                         // GPMI_UPDATE256=0: round 2's 128 x 128 kernel everywhere)
    bool side_one_per_xcd = false;       // set around a look-ahead chain whose update will run as update256_kernel (chol.h, side_slots)
    int update256_ablation = 0;         // tools builds: ABL bits of update256_kernel for gpmi_bench_gemm (variant 256 + bits)
    int update256_atomic = 0;            // the C tile of a subtracting launch goes out as no-return atomic adds instead of load / add / store (GPMI_UPDATE256_ATOMIC)
    int64_t update256_rect_min_m = 8192; // rectangular / batched products go to the 256 x 128 kernel from this many rows on (GPMI_UPDATE256_RECT: test hook)
    int64_t update256_min_tiles = 1024;  // ... from this many 256 x 128 tiles on (GPMI_UPDATE256_MIN: test hook)
    // the persistent chain kernel (chain.hip): one launch per diagonal super-block instead of ~15 dependent launches per 256 columns
    int chain_kernel = 1;                // GPMI_CHAIN=0: the multi-launch chain of rounds 1-4 (factor_diag_block + build_super_inverse)
    void* chain_sync = nullptr;          // its task counter, tile flags and column / row counters (zeroed before every launch)
    int chain_nb_max = 32;               // blocks of up to 32 x 64 = 2048 columns
    int chain_wgs_max = 64;              // workgroups of a chain launch that has the device to itself (the first block, serial tails)
    int chain_wgs = 0;                   // > 0: the number of workgroups of EVERY chain launch (GPMI_CHAIN_WGS: test hook)
    int chain_beside_wgs = 16;           // workgroups of a chain launch beside a FULL-GRID update (update_full_grid: the chain is placed before the update starts, so
                                         // more than one per XCD is no dispatcher lottery any more; its units go back to the update when it exits).  GPMI_CHAIN_BESIDE_WGS;
                                         // 8 / 16 / 32: a 1024 block 0.82 / 0.60 / 0.52 ms by events beside the update, C2 59.8 / 59.5 / 59.8 ms, N = 50 000 flat (profiles/r06_m_*)
    bool chain_wide_ok = false;          // set around such a chain launch (chol.h dense path; dev_hip.hip one-rank blocked path)
    unsigned chain_started_expect = 0;   // workgroups of all chain launches so far (what the never-reset `started` word counts up to)
    bool chain_wait_pending = false;     // the next trailing update on an unmasked stream first waits for the last chain launch's workgroups to be placed
    int64_t tail_fuse = 2048;            // the LAST rows of a factorisation (at most this many) are ONE diagonal block: one chain launch instead of a
                                         // dozen 256-wide panels with their updates (GPMI_TAIL_FUSE; 0 = off; needs the chain kernel)
    hipStream_t own_stream = nullptr;    // the stream created with the context
    bool beside_update = false;          // launches made now run in the reserved slots beside the persistent update: no whole-CU kernels
    std::vector<hipEvent_t> la_events;
    size_t la_next = 0;   // cross-stream dependencies, reused by every factorisation
    bool prof_on = false;
    int prof_only = -1;  // >= 0: bracket the launches of THIS class only (gpmi_profile_enable(ctx, 2 + cls))
    bool prof_phases_only = false;  // gpmi_profile_enable(ctx, 65): the GPMI_PROF_STEP_* phases of the blocked driver only
    bool prof_skip_chain = false;  // gpmi_profile_enable(ctx, 64): every class, but not the thousands of tiny chain kernels (diag64 / rows64 / rows256)
    hipEvent_t attach_a = nullptr, attach_b = nullptr;  // events the NEXT update-kernel launch carries itself (hipExtLaunchKernelGGL): no
                                                         // marker packets around the persistent kernel (gemm.hip, ProfScope attach mode)
    std::vector<gpmi::ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    int64_t prof_n[GPMI_PROF_NCLASS] = {0};
    double prof_ms[GPMI_PROF_NCLASS] = {0};
    double prof_work[GPMI_PROF_NCLASS] = {0};
    double prof_bytes[GPMI_PROF_NCLASS] = {0};  // algorithmic HBM bytes (trailing update: C read + write, operand panels once)
};

namespace gpmi {
// Launches made beside the persistent trailing update (chol.h: beside_update) must not have more workgroups than the slots
// that update leaves free: measured (profiles/r02_twolevel_critical_path.txt), a 12-workgroup rows64 launch on the side
// stream ended only when the update did, 25 ms later, while launches of <= 8 workgroups ran in ~0.1 ms.  Capped launches
// walk their work items grid-stride.
// Beside the 256 x 128 update on an unmasked stream the free room is eight WHOLE CUs, one per XCD, instead of sixteen half-CU slots:
// a side launch of more than one workgroup per XCD ends with the update (the dispatcher queues workgroups on shader engines that
// have no free CU — measured on rows64 and on the 128 x 64 products alike, profiles/r03_r_update256.log), so the cap is 8 there.
inline int side_slots(const gpmi_ctx* c) { return (c->side_one_per_xcd && c->lookahead_slots > 8) ? 8 : c->lookahead_slots; }
inline int64_t side_cap(const gpmi_ctx* c, int64_t nwg) {
    const int64_t cap = side_slots(c);
    return c->beside_update && nwg > cap ? cap : nwg;
}
}  // namespace gpmi

struct gpmi_gp {
    gpmi_ctx* ctx = nullptr;
    void* group = nullptr;    // non-null: a blocked model sharded over the devices of the context's group, one BlockedGP per member
    void* blocked = nullptr;  // non-null: a BLOCKED handle (gpmi_gp_create_blocked; dev_hip.hip / blocked.cpp) — the dense fields below are unused
    int dtype = 64;
    int d = 0;
    int64_t n = 0;     // observations
    int64_t npad = 0;  // n rounded up to IB (padding rows/cols are the identity)
    int64_t ld = 0;    // leading dimension (elements) of the row-major factor
    void* x = nullptr;       // n x d row-major (== d x n col-major), dtype
    void* A = nullptr;       // (npad + 8) x ld; lower triangle holds L (K = L L'), row npad holds z = L^-1 y
    void* ymu = nullptr;     // y - mu, npad elements (zero padded)
    void* alpha = nullptr;   // npad elements
    void* invdiag = nullptr; // 1 / L_ii, npad elements
    void* g1 = nullptr;      // gradient path scratch: L^-T rows, then reused (npad x ld), allocated on first gpmi_grad
    void* g2 = nullptr;      // gradient path: (K + noise)^-1, lower triangle (npad x ld)
    double* gpart = nullptr; // gradient path: per-block partial sums
    int64_t gpart_cap = 0;
    void* supinv = nullptr;  // explicit inverses of the diagonal super-blocks of the last factorisation, back to back (chol.h SuperStore)
    int64_t supinv_cap = 0;
    std::vector<gpmi::SuperPart> sup_parts;  // their column ranges
    void* linv256 = nullptr; // explicit inverses of the NB x NB diagonal blocks, ceil(npad / NB) x NB x NB (linv256_kernel)
    void* linv = nullptr;    // inverses of the 64 x 64 diagonal blocks, (npad / 64) x 64 x 64 (every later solve is a GEMM)
    double* noise = nullptr; // per-point nugget (heteroscedastic) or nullptr
    bool fitted = false;
    double logdet = 0.0;
    double mll = 0.0;
    // scratch for predict / solve, grown on demand
    void* rows = nullptr;
    int64_t rows_cap = 0;  // rows allocated (each ld wide)
    void* xp = nullptr;
    int64_t xp_cap = 0;
    void* small = nullptr;  // mean / mu / var staging
    int64_t small_cap = 0;
};

namespace gpmi {

// blocked handles (dev_hip.hip)
class BlockedGP;
BlockedGP* blocked_of(gpmi_gp* gp);  // the (rank-0) driver of a blocked handle, or nullptr
void blocked_destroy(void* p);
// in-process device groups: gpmi_ctx_create(n_devices > 1) — one context per device id, an in-process communicator (peer copies
// ordered by events), one worker thread per member for the duration of a call (dev_hip.hip)
int create_member_context(int dev, gpmi_ctx** out);
int group_create(gpmi_ctx* primary, int n, const int* device_ids);
std::vector<gpmi_ctx*> group_members(gpmi_ctx* c);  // the members of c's device group, or {c}
void group_destroy(gpmi_ctx* primary);
int group_gp_create(gpmi_ctx* primary, int dtype, int d, int64_t n, const void* x, int64_t block_rows, int stripe_blocks, gpmi_gp** out);
void group_gp_destroy(gpmi_gp* gp);
int group_fit(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, int64_t n_noise, const void* ymu, double* mll_out, void* alpha_out,
              int64_t* info_out);
int group_predict(gpmi_gp* gp, const gpmi_kernel* k, int64_t p, const void* xpred, const void* mean_pred, int full_cov, void* mu_out, void* var_out);
int group_grad(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, int64_t n_noise, double* dkern_out, int n_kern, double* dnoise_out);
int group_factor_diag(gpmi_gp* gp, void* out);
int group_solve(gpmi_gp* gp, int64_t nrhs, void* b_inout, bool backward);
int group_update_alpha(gpmi_gp* gp, const void* ymu, double* mll_out, void* alpha_out);
int group_inv_diag(gpmi_gp* gp, void* out);
int group_factor_to_host(gpmi_gp* gp, void* U_out);

// which look-ahead stream set the next factorisation uses: whole CUs (side_masked + upd_stream) or free slots (side_stream).
// Returns the mode in effect (0 when masks are unavailable).
int set_lookahead_mode(gpmi_ctx* c, bool whole_cus);

// RAII-free helpers -------------------------------------------------------------------------
#define GPMI_HIP(ctx, call)                                                                     \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                    \
            return GPMI_EDEVICE;                                                                \
        }                                                                                       \
    } while (0)

struct ProfScope {
    gpmi_ctx* c;
    int idx = -1;
    bool attach = false;  // the events ride on the kernel dispatch itself (ctx->attach_a / _b) instead of marker packets around it
    ProfScope(gpmi_ctx* ctx, int cls, double work, double bytes = 0.0, bool attach_to_launch = false, bool chain_kernel = false);
    ~ProfScope();
};

// host helpers defined in api.hip
int upload_program(gpmi_ctx* c, const gpmi_kernel* k, int d);                 // digest + upload the kernel program
int grow(gpmi_ctx* c, void** p, int64_t* cap, int64_t need_bytes);            // (re)allocate a device scratch buffer

// kernel launchers (each enqueues on ctx->stream; T = double | float) -------------------------
enum CovFlags { COV_LOWER = 1, COV_NUGGET = 2, COV_PAD_IDENTITY = 4, COV_NO_FAST = 8 /* internal: the interpreter takes every tile */,
                COV_GLOBAL_X = 16 /* internal: d too large to stage in LDS, operands read from global memory */ };

// C[i][j] = k(xa_i, xb_j) for i < nrows_total, j < ncols_total (row-major, ld = ldc).
// rows >= na / cols >= nb are padding: 0, or the identity when COV_PAD_IDENTITY.
// row_off = global index of local row 0 (diagonal / nugget / lower-skip tests use row_off + i; nugget_vec is
// indexed by the global row).
template <typename T>
void launch_cov(gpmi_ctx* ctx, const T* xa, int64_t na, const T* xb, int64_t nb, int d, T* C, int64_t ldc,
                int64_t nrows_total, int64_t ncols_total, int flags, double nugget, const double* nugget_vec,
                int64_t row_off = 0);

// C[M x N] -= A[M x K] * B[N x K]'   (all row-major; K % 16 == 0 for double, % 32 for float)
// lower != 0: only tiles that intersect {col <= row} are computed (C is a trailing square).
template <typename T>
void launch_gemm_nt(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M,
                    int64_t N, int64_t K, int lower, const int* info);

// same kernel with an explicit tile shape (mode 2 = staircase of a row-block-cyclic shard, tile_order.h);
// shape.ntm / shape.ntn are filled in from M and N
// batch != nullptr: `count` independent products in one launch, operands / output of product b offset by b * stride
// elements (split-K into separate partial outputs: strideA = strideB = K, strideC = one output matrix)
struct GemmBatch {
    int count;
    int64_t strideA, strideB, strideC;
};
template <typename T>
void launch_gemm_shape(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M,
                       int64_t N, int64_t K, TileShape shape, const int* info, int flags = 0, const GemmBatch* batch = nullptr);
// the trailing update in 256 x 128 tiles (update256.hip): true when it took the launch
template <typename T>
bool launch_update256(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                      TileShape shape, const int* info, int flags, const GemmBatch* batch);
// ... and whether it WOULD take it (no launch): the look-ahead sizes the chain's launches by it (chol.h)
template <typename T>
bool update256_applies(const gpmi_ctx* ctx, const T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                       TileShape shape);
enum GemmFlags { GEMM_OVERWRITE = 1 /* C = A B' instead of C -= A B' */, GEMM_KSTART_ROW = 2 /* A[i][k] = 0 for k < i: start K at the tile's first row */,
                 GEMM_KEND_COL = 8 /* B[j][k] = 0 for k > j: end K at the tile's last column */,
                 GEMM_PHASE_LOCK = 64 /* tiles of an XCD start round by round (gemm.hip QueueArgs::done_base) */,
                 GEMM_NEGOUT = 256 /* with GEMM_OVERWRITE: C = -A B' */,
                 GEMM_HEAVY_FIRST = 512 /* set by the launcher on GEMM_KEND_COL rectangles: every strip is walked from its last (longest-K) column */,
                 GEMM_AUX = 4 /* no effect on the kernel: account the launch to the panel class, not to the trailing update */,
                 GEMM_NO_PAIR16 = 32 /* tools: 8-byte instead of 16-byte C accesses in fp64 (A/B of the access width) */ };

// in-place Cholesky of the 64 x 64 block at A (row-major, ld): lower factor, upper part zeroed; linv (64 x 64,
// row-major, ld 64) receives L^-1 and invdiag[0..64) 1 / L_jj.  On a non-positive pivot j (0-based) writes
// *info = pivot_base + j + 1.
template <typename T>
void launch_diag64(gpmi_ctx* ctx, T* A, int64_t ld, T* linv, T* invdiag, int* info, int64_t pivot_base);

// Panel step for the rows below column block j of a panel (Xp, Lp point at the panel's first column k0):
//   X_j <- (X_j - X[:, 0:K1] Lp[0:64, 0:K1]') Linv'   and, for the first diag_rows rows (Cholesky only),
//   the rows' own 64 x 64 diagonal block -= X_j X_j'.
template <typename T>
void launch_rows64(gpmi_ctx* ctx, T* Xp, int64_t ldx, int64_t M, int K1, const T* Lp, int64_t ldl, const T* linv,
                   int64_t diag_rows, const int* info);

// The whole panel step (nsub <= 4 column blocks of 64) for rows BELOW the panel's diagonal block, one launch:
//   Xp: the rows at the panel's first column; Lp: the panel's diagonal block (ldl); linv: its nsub stored 64 x 64 inverses
template <typename T>
void launch_rows256(gpmi_ctx* ctx, T* Xp, int64_t ldx, int64_t M, int nsub, const T* Lp, int64_t ldl, const T* linv,
                    const int* info);

// One step of the backward solve  L' alpha = z  for the 64-block starting at j0:
//   alpha[j0..j0+64) = Linv_b' z[j0..)  (linv = stored inverse of the diagonal block);  z[0..j0) -= L[j0..j0+64, 0..j0)' alpha_b
// Arow points at row j0 of the factor (so a shard can pass its local copy of that block-row).
template <typename T>
void launch_bsolve_step(gpmi_ctx* ctx, const T* Arow, int64_t ld, int64_t j0, const T* linv, T* z, T* alpha);

// mll / logdet / y'alpha  ->  out[0] = mll, out[1] = logdet, out[2] = y'alpha
template <typename T>
void launch_linv256(gpmi_ctx* ctx, const T* A, int64_t ld, const T* linv64, T* out, int64_t npad, const int* info);
// level 0 of the super-panel inverse: packed NB x NB inverses onto the diagonal of LW and (transposed) LWT
template <typename T>
void launch_place_inv_blocks(gpmi_ctx* ctx, const T* l256, T* LW, T* LWT, int64_t wld, int nblk);
// chain.hip: the w x w diagonal block at A (leading dimension ld) factored in place, its 64 x 64 diagonal inverses (linv, w / 64 tiles),
// 1 / L_ii (invdiag) and — LW != nullptr — its explicit inverse (leading dimension wld, strict upper part zero), all in ONE launch on
// ctx->stream.  false: the launch does not apply (switched off, refinement wanted, w not a multiple of 64 or too wide): nothing was enqueued.
template <typename T>
bool launch_chain_block(gpmi_ctx* ctx, T* A, int64_t ld, int64_t w, T* linv, T* invdiag, T* LW, int64_t wld, int* info, int64_t pivot_base);
int64_t chain_sync_bytes(int nb_max);
void launch_chain_wait(gpmi_ctx* ctx);  // on ctx->stream, when a chain launch is pending beside it (no-op otherwise)
template <typename T>
void launch_bsolve256(gpmi_ctx* ctx, const T* Arow, int64_t ld, int64_t k0, int nbk, const T* linv256, T* z, T* alpha,
                      int64_t ldinv = NB);  // ldinv: row stride of the NB x NB inverse (a diagonal block of a wider explicit inverse)
template <typename T>
void launch_finalize(gpmi_ctx* ctx, const T* A, int64_t ld, int64_t n, const T* y, const T* alpha, double* out);

// mu[p] = mean[p] + sum_j R[p][j] * alpha[j]   (j < n)
template <typename T>
void launch_row_gemv(gpmi_ctx* ctx, const T* R, int64_t ldr, int64_t P, int64_t n, const T* alpha, const T* mean,
                     T* mu);
// var[p] = max(kdiag - sum_j R[p][j]^2, 0)
template <typename T>
void launch_row_var(gpmi_ctx* ctx, const T* R, int64_t ldr, int64_t P, int64_t n, double kdiag, T* var);

// out[p] = sum_{j >= (p / blk) * blk} R[p][j]^2   (blk = 0: whole row)
template <typename T>
void launch_row_sumsq(gpmi_ctx* ctx, const T* R, int64_t ldr, int64_t P, int64_t n, int64_t blk, T* out);

template <typename T>
void launch_logdiag(gpmi_ctx* ctx, const T* A, int64_t ld, int64_t nrows, int64_t col_off, double* out);

// gradient path -------------------------------------------------------------------------------------------
// the register / LDS forms of the gradient kernel (grad.hip) take up to this many hyper-parameters ((n_hyp + 1) x 256 double
// accumulators + the 64 x d row points share the 160 KB of LDS) and input dimensions (the pair's d squared differences live in
// registers); beyond either the limit-free form runs (dmll_kernel<T, 0, .>): no kernel is refused
constexpr int GRAD_MAX_HYP = 64;
constexpr int GRAD_MAX_D = 32;
// A[i][i] = 1, everything else 0 (n x n, row-major)
template <typename T>
void launch_set_identity(gpmi_ctx* ctx, T* A, int64_t ld, int64_t n);
// partial[b][0..n_hyp) = sum over block b of (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta_p (diag counted half),
// partial[b][n_hyp] = block b's share of tr(alpha alpha' - Kinv); returns the number of blocks
template <typename T>
int64_t launch_dmll(gpmi_ctx* ctx, const T* x, int64_t n, int d, const T* alpha, const T* Kinv, int64_t ld, double* partial,
                    int n_hyp, bool kinv_negated = false);  // kinv_negated: Kinv holds -K^-1
// the same reduction over the rectangle of pairs (xa_i, xb_j) with an explicit weight matrix Wt (na x nb, ld): FITC gradient;
// partial[b][0..n_hyp) (slot n_hyp is zero); returns the number of blocks
template <typename T>
int64_t launch_dmll_rect(gpmi_ctx* ctx, const T* xa, int64_t na, const T* xb, int64_t nb, int d, const T* Wt, int64_t ld,
                         double* partial, int n_hyp);
// out[s] = sum_b partial[b][s]  (deterministic order)
void launch_reduce_partials(gpmi_ctx* ctx, const double* partial, int64_t nblocks, int nslots, double* out);

// isolated timing of the update kernel on random operands (variant 0 = product kernel)
template <typename T>
int gemm_bench(gpmi_ctx* ctx, int64_t M, int64_t N, int64_t K, int lower, int variant, int iters, double* ms_out);

// MFMA peak micro-benchmark; returns achieved TFLOP/s
template <typename T>
int mfma_peak(gpmi_ctx* ctx, double* tflops);

}  // namespace gpmi
