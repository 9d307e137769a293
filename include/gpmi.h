/* gpmi.h — C ABI of libgpmi.so: MI355X-native exact-GP fit / predict.
 *
 * This is the drop-in boundary for ONE hot path of STOR-i/GaussianProcesses.jl:
 *   GPE.update_mll!  (cov! -> nugget -> Cholesky -> alpha -> logdet -> mll)
 *   predict_f        (cross-cov -> whiten -> mean / variance)
 * and, as the "next" rows of SURVEY.md 8(f) built on the same kernels:
 *   update_dmll!     (gpmi_grad), FITC update_mll! / predict_f (gpmi_fitc_*),
 *   predict_LOO      (gpmi_inv_diag).
 * The reference has no FFI; its seam is Julia dispatch on CovarianceStrategy /
 * AbstractPDMat (src/GP.jl:10-20).  Each entry point below names the reference
 * code it replaces (paths relative to the reference repository root); the
 * Julia-side binding that calls them is shown in INTEGRATION.md and
 * gaussianprocesses.jl_amd/julia/GPMI355X.jl.
 *
 * Conventions
 *   - Every matrix is column-major as Julia passes it.  `x` is d x n (one
 *     observation per column, src/GPE.jl:41) == a C row-major n x d array.
 *   - Return codes: GPMI_OK; GPMI_ENOTPD (-> LinearAlgebra.PosDefException(info),
 *     src/GP.jl:110, caught by src/optimize.jl:56-58,81-83); GPMI_EARG
 *     (-> ArgumentError, src/GPE.jl:42,129, src/GP.jl:65,103,
 *     src/kernels/kernels.jl:34,41,62,64); GPMI_EDEVICE (-> ErrorException with
 *     gpmi_last_error()).  After any non-zero return the handle stays usable.
 *   - The caller owns every host pointer; nothing is retained past return.
 *     Device memory belongs to the handle (Julia: finalizer -> gpmi_gp_destroy).
 *   - Calls are synchronous: results are on the host when a call returns.
 *   - dtype 64 = IEEE double end to end (the reference's only precision,
 *     src/GP.jl:14-20); dtype 32 = float storage/arithmetic with fp64
 *     reductions (no reference counterpart; parity is rtol 1e-2 vs fp64).
 *   - There is NO CPU backend: without a gfx950 device gpmi_ctx_create fails.
 */
#ifndef GPMI_H
#define GPMI_H

#include <stdint.h>

/* libgpmi.so is built with -fvisibility=hidden: the entry points below are its ONLY dynamic symbols */
#define GPMI_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

#define GPMI_OK 0
#define GPMI_ENOTPD 1
#define GPMI_EARG 2
#define GPMI_EDEVICE 3

#define GPMI_MAX_OPS 32     /* nodes in a kernel tree (leaves + SUM / PROD); the evaluation stack may be 6 deep.  The input
                             * dimension d, the number of stored parameters and of masked active-dim entries are NOT limited:
                             * the library sizes its tables from the descriptor (the reference's loops take any `dim`). */

/* Kernel-tree node codes.  A kernel is a POSTFIX program over these. */
enum gpmi_op {
    GPMI_K_SE_ISO = 1,     /* src/kernels/se_iso.jl:39     s2*exp(-0.5*r/l2),   r = sq-euclid        params [l2, s2]        */
    GPMI_K_SE_ARD = 2,     /* src/kernels/se_ard.jl:43     s2*exp(-r/2),        r = weighted sq      params [il2[nd], s2]   */
    GPMI_K_MAT12_ISO = 3,  /* src/kernels/mat12_iso.jl:41  s2*exp(-r/l),        r = euclid           params [l, s2]         */
    GPMI_K_MAT12_ARD = 4,  /* src/kernels/mat12_ard.jl:43  s2*exp(-r),          r = weighted euclid  params [il2[nd], s2]   */
    GPMI_K_MAT32_ISO = 5,  /* src/kernels/mat32_iso.jl:41  s=sqrt3 r/l; s2(1+s)e^-s                  params [l, s2]         */
    GPMI_K_MAT32_ARD = 6,  /* src/kernels/mat32_ard.jl:43  s=sqrt3 r                                 params [il2[nd], s2]   */
    GPMI_K_MAT52_ISO = 7,  /* src/kernels/mat52_iso.jl:40  s=sqrt5 r/l; s2(1+s+s^2/3)e^-s            params [l, s2]         */
    GPMI_K_MAT52_ARD = 8,  /* src/kernels/mat52_ard.jl:43  s=sqrt5 r                                 params [il2[nd], s2]   */
    GPMI_K_RQ_ISO = 9,     /* src/kernels/rq_iso.jl:44     s2*(1+r/(2 a l2))^-a                      params [l2, s2, a]     */
    GPMI_K_RQ_ARD = 10,    /* src/kernels/rq_ard.jl:47     s2*(1+0.5 r/a)^-a                         params [il2[nd], s2, a]*/
    GPMI_K_NOISE = 11,     /* src/kernels/noise.jl:29-39   s2*[all_z x_z ~= y_z]  (isapprox, rtol sqrt(eps)) params [s2]    */
    GPMI_K_CONST = 12,     /* src/kernels/const.jl:36      s2                                        params [s2]            */
    GPMI_K_SUM = 100,      /* src/kernels/sum_kernel.jl:15  pops right, left; pushes left+right                             */
    GPMI_K_PROD = 101      /* src/kernels/prod_kernel.jl:14 pops right, left; pushes left*right                             */
};

/* Flattened kernel tree (what a Julia `Kernel` serialises to).
 *   ops[n_ops]          postfix program (leaves push, SUM/PROD combine).
 *   dims_off[n_ops+1]   leaf i acts on input rows dims[dims_off[i] .. dims_off[i+1])
 *                       (0-based; this is how Masked, src/kernels/masked_kernel.jl:44-49,
 *                       is expressed).  An EMPTY range means "all d rows".  FixedKernel
 *                       (fixed_kernel.jl:69) does not change cov and needs no encoding.
 *   params[n_params]    the kernels' STORED (transformed) fields in program order,
 *                       exactly the struct fields of the reference types (l2 | l |
 *                       il2[], s2, a) so host and device evaluate the same expression.
 *                       nd = size of the leaf's active-dim range (d when empty). */
typedef struct gpmi_kernel {
    int32_t n_ops;
    const int32_t* ops;
    const int32_t* dims_off;
    const int32_t* dims;
    const double* params;
    int32_t n_params;
} gpmi_kernel;

typedef struct gpmi_ctx gpmi_ctx; /* one per process: device(s), streams, error text */
typedef struct gpmi_gp gpmi_gp;   /* one per GPE: resident x, factor, alpha          */

/* ---- context ---------------------------------------------------------- */
/* n_devices == 1: one GPU (device_ids may be NULL -> device 0).
 * n_devices  > 1: an in-process DEVICE GROUP — one context per entry of device_ids (an id may repeat), joined by an
 * in-process communicator (peer copies ordered by events: no library, no launcher).  gpmi_gp_create_blocked(ctx, NULL, ...)
 * on such a context shards the model row-block-wise over the group's devices and runs one worker thread per member inside
 * every gpmi_fit / gpmi_predict / gpmi_grad on it; every other entry point runs on device_ids[0].  This is the
 * single-process multi-GPU form (a Julia session driving all the GPUs of a node); the one-process-per-GPU form is a
 * communicator (gpmi_comm_*) on single-device contexts.
 * CU PARTITIONS: a device id may be given as  id + 256 * (1 + p),  p = 0 | 1: the context's streams are then confined to half p
 * of every XCD (16 of its 32 compute units, 128 in all), so two such contexts — or the two members of a device group
 * {id + 256, id + 512} — run side by side on one MI355X without sharing a compute unit: the software stand-in for the driver's
 * compute partitioning where that is not available (two models on one GPU; two logical devices for the multi-device path).  */
GPMI_API int gpmi_ctx_create(int n_devices, const int* device_ids, gpmi_ctx** out);
GPMI_API void gpmi_ctx_destroy(gpmi_ctx*);
/* waits for ALL work on the context's device(s) (every entry point already returns with its results on the host; this is the
 * explicit bracket a timing harness puts around a measured region: hipDeviceSynchronize)                                  */
GPMI_API int gpmi_ctx_synchronize(gpmi_ctx*);
GPMI_API const char* gpmi_last_error(gpmi_ctx*);
GPMI_API const char* gpmi_version(void);

/* ---- model object: replaces alloc_cK (src/GP.jl:14-20) + KernelData ----- */
/* Copies x (d x n col-major, element type given by dtype) to the device and
 * allocates ONE n x n factor buffer (the reference keeps two, GP.jl:16-17).   */
GPMI_API int gpmi_gp_create(gpmi_ctx*, int dtype /*64|32*/, int d, int64_t n, const void* x, gpmi_gp** out);
GPMI_API void gpmi_gp_destroy(gpmi_gp*);

/* ---- fit: replaces update_cK! + update_mll! (src/GPE.jl:169-212) and
 *      make_posdef! (src/GP.jl:101-112) -------------------------------------
 * log_noise: n_noise == 1 -> scalar logNoise (nugget exp(2 logNoise), GPE.jl:173)
 *            n_noise == n -> heteroscedastic vector (GPE.jl:177-186).
 * y_minus_mu: y - mean(m, x) (GPE.jl:206-207; the mean stays on the host).
 * mll_out:   -(y'alpha + logdet + n log 2pi)/2  (GPE.jl:210), always double.
 * alpha_out: n elements of dtype, may be NULL.
 * info_out:  0, or the 1-based failing pivot when GPMI_ENOTPD is returned.    */
GPMI_API int gpmi_fit(gpmi_gp*, const gpmi_kernel*, const double* log_noise, int64_t n_noise,
             const void* y_minus_mu, double* mll_out, void* alpha_out, int64_t* info_out);

/* ---- update_mll!(gp; noise = false, kern = false): only the mean changed (src/GPE.jl:203-211 with update_cK! skipped) ----
 * The factor of the last gpmi_fit is KEPT:  alpha = cK \ (y - mu)  through it,  mll = -(y'alpha + logdet + n log 2pi)/2  from the
 * stored logdet.  The DEVICE copy of alpha (what gpmi_predict and gpmi_grad read) is replaced too, on dense, blocked and
 * device-group handles alike (every rank of a sharded model passes the same y_minus_mu).  alpha_out: n elements, may be NULL.  */
GPMI_API int gpmi_update_alpha(gpmi_gp*, const void* y_minus_mu, double* mll_out, void* alpha_out);

/* ---- predict: replaces predict_f / predictMVN (src/GP.jl:25-84) ----------
 * xpred: d x p col-major.  mean_pred: mean(m, xpred), p elements.
 * full_cov == 0: var_out[p] = max(k(x*,x*) - |L^-1 k*|^2, 0)   (GP.jl:69-77, batched)
 * full_cov != 0: var_out[p x p] = Kpred - (L^-1 K*)'(L^-1 K*), no clamp (GP.jl:25-30,51-54) */
GPMI_API int gpmi_predict(gpmi_gp*, const gpmi_kernel*, int64_t p, const void* xpred, const void* mean_pred,
                 int full_cov, void* mu_out, void* var_out);

/* ---- gradient: replaces update_dmll! (src/GPE.jl:298-324) for the kernel and noise parts ----
 * After a successful gpmi_fit with the SAME kernel / log_noise:
 *   dkern_out[p] = d mll / d theta_p for the kernel's log-scale parameters in get_params order
 *                  (leaf files' get_params; composites left then right, src/kernels/pair_kernel.jl:15):
 *                  1/2 sum_ij (alpha alpha' - K^-1)_ij dK_ij/dtheta_p   (dmll_kern!, GPE.jl:219-241)
 *   dnoise_out   = exp(2 logNoise) tr(alpha alpha' - K^-1)             (dmll_noise, GPE.jl:273-275; may be NULL)
 * The mean part, dot(grad_mean, alpha) (GPE.jl:282-288), is O(N d) host work on alpha.
 * n_kern must equal the kernel's parameter count.  Allocates two more n x n device buffers on
 * first use.  No limit on d (up to 2^20; cov! reads its operands from global memory beyond d = 64); beyond d = 32 / 64 parameters
 * a slower form of the trace kernel runs whose per-wave table of (n_kern + 1) x 4 doubles lives in LDS: kernels with more than
 * GPMI_GRAD_MAX_PARAMS hyper-parameters are refused with GPMI_EARG (also by gpmi_fitc_grad and on blocked handles).          */
#define GPMI_GRAD_MAX_PARAMS 5000
GPMI_API int gpmi_grad(gpmi_gp*, const gpmi_kernel*, const double* log_noise, int64_t n_noise, double* dkern_out, int32_t n_kern,
              double* dnoise_out);

/* ---- FITC sparse approximation (BASELINE.json configs[4]) ----------------------
 * Replaces, for covstrat = FullyIndepStrat(inducing):
 *   alloc_cK / update_cK!(::FullyIndepPDMat, ...)   src/sparse/fully_indep_train_conditional.jl:118-156
 *   `\`, logdet of FullyIndepPDMat                   :38-41, :80            (inside update_mll!, src/GPE.jl:202-212)
 *   get_alpha_u, predictMVN(::FullyIndepStrat)      :279-286, :321-329 (DTC: determ_train_conditional.jl:41-59,
 *                                                   SoR: subsetofregressors.jl:303-321)
 * x: n x d row-major (== Julia's d x n), xu: m x d row-major (== the d x m `inducing` matrix).  Both make_posdef!
 * nuggets (1e-10 on Kuu and on SigmaQR) are applied as the reference does.  log_noise is a scalar (the reference's
 * FITC takes logNoise::Real).  alpha_out (n) = cK \ (y - mu), mll_out as GPE.jl:210 with the determinant lemma.
 * GPMI_ENOTPD when Kuu / SigmaQR fail to factor or a Lambda_i is not positive (info = pivot / 1-based index).
 * Device memory: three n x m matrices (Kfu, its whitened image, Kuf) -- sized for 288 GB, not for a host. */
typedef struct gpmi_fitc gpmi_fitc;
GPMI_API int gpmi_fitc_create(gpmi_ctx*, int dtype, int d, int64_t n, const void* x, int64_t m, const void* xu, gpmi_fitc** out);
GPMI_API void gpmi_fitc_destroy(gpmi_fitc*);
GPMI_API int gpmi_fitc_fit(gpmi_fitc*, const gpmi_kernel*, double log_noise, const void* y_minus_mu, double* mll_out, void* alpha_out,
                  int64_t* info_out);
/* mu_out[p] = mean_pred[p] + Kxu alpha_u ; var_out: p variances clamped at 0 (GP.jl:75), or the p x p matrix
 * Kxx - Qxx + Kxu SigmaQR^-1 Kux (col-major == row-major, symmetric) when full_cov != 0 */
GPMI_API int gpmi_fitc_predict(gpmi_fitc*, const gpmi_kernel*, int64_t p, const void* xpred, const void* mean_pred, int full_cov,
                      void* mu_out, void* var_out);
/* alpha_u = SigmaQR \ (Kuf (Lambda \ (y - mu))), m elements (get_alpha_u) */
GPMI_API int gpmi_fitc_alpha_u(gpmi_fitc*, void* out);
/* update_dmll! on the FITC model of the last gpmi_fitc_fit (same kernel, same log_noise): dmll_kern!
 * (ASSUMES stationary leaves — every gpmi_op above is: the per-point term sum_i q_i dk(x_i,x_i)/dtheta of :218 is taken as
 * (sum_i q_i) dk/dtheta at r = 0; a kernel with any other leaf is refused with GPMI_EARG.)
 * (src/sparse/fully_indep_train_conditional.jl:200-234 over subsetofregressors.jl:219-256) -> dkern_out[n_kern] in
 * get_params order, dmll_noise (:243-257) -> *dnoise_out.  The mean part (GPE.jl:282-288) is grad_stack' * alpha on the host. */
GPMI_API int gpmi_fitc_grad(gpmi_fitc*, const gpmi_kernel*, double log_noise, double* dkern_out, int32_t n_kern, double* dnoise_out);

/* ---- blocked model object: packed storage on one device, row-block sharding over several ---------
 * The same gpmi_gp handle type and the same gpmi_fit / gpmi_predict / gpmi_grad / gpmi_logdet / gpmi_factor_diag, with the
 * factor of K + noise held as block-rows of block_rows = 256 * 2^s rows (0: 1024 from 16 384 points — 2048 from 131 072 on one rank —, 512 from 4096, 256 below)
 *   - dealt round-robin over the ranks of `comm` (one process per GPU; every rank makes the same calls with the same
 *     arguments and receives the same results: mll, alpha, mu, var, gradient are replicated), and
 *   - per rank, in stripes of stripe_blocks local blocks that stop at their own diagonal (0: one stripe = full rows), so the
 *     upper triangle is never allocated: N^2 (1 + 1/S) / 2 elements instead of alloc_cK's two N x N (src/GP.jl:14-20).
 * comm == NULL: one rank (a single device past the N x N ceiling: N = 250 000 fp64 on 288 GB) — or, when ctx is a device
 * group (gpmi_ctx_create with n_devices > 1), one rank per device of the group.  gpmi_grad on a blocked
 * handle needs ONE more own-rows x N matrix (N^2 / world) instead of two N x N; gpmi_inv_diag uses the same scratch.  The
 * AbstractPDMat surface (gpmi_solve / gpmi_whiten / gpmi_inv_diag / gpmi_factor_to_host) answers on a blocked handle too: every
 * rank passes the same right-hand sides and receives the same results (gpmi_factor_to_host gathers the whole n x n factor on
 * every rank's host: for inspection at sizes where that is affordable).                                                  */
typedef struct gpmi_comm gpmi_comm;
/* Collectives on DEVICE buffers, to be enqueued on the HIP stream passed as `stream` (ordered after the work already on it;
 * later work on it must see the result); host_allreduce reduces n host doubles in place (op 0 sum, 1 min, 2 max).
 * Every function returns 0 on success.                                                                                    */
typedef struct gpmi_comm_callbacks {
    void* user;
    int (*broadcast)(void* user, void* buf, int64_t bytes, int root, void* stream);
    int (*all_gather)(void* user, const void* send, void* recv /* world x bytes_each */, int64_t bytes_each, void* stream);
    int (*all_reduce_sum)(void* user, void* buf, int64_t count, int elem_bytes /* 8: double, 4: float */, void* stream);
    int (*host_allreduce)(void* user, double* vals, int32_t n, int32_t op);
} gpmi_comm_callbacks;
GPMI_API int gpmi_comm_create_callbacks(const gpmi_comm_callbacks* cb, int rank, int world, gpmi_comm** out);
/* RCCL directly (librccl.so is opened at run time): rank 0 calls gpmi_comm_unique_id and hands the 128 bytes to every rank
 * (any out-of-band channel: the launcher's store, a file, MPI); every rank then calls gpmi_comm_create_rccl on its context.  */
GPMI_API int gpmi_comm_unique_id(void* id128_out);
GPMI_API int gpmi_comm_create_rccl(gpmi_ctx*, const void* id128, int rank, int world, gpmi_comm** out);
GPMI_API void gpmi_comm_destroy(gpmi_comm*);
/* every collective of `comm` on small device buffers with rank-dependent patterns, verified on the host (collective call:
 * all ranks).  A launcher runs it once before its first fit so that a broken transport fails with a message of its own.  */
GPMI_API int gpmi_comm_selftest(gpmi_ctx*, gpmi_comm*);
GPMI_API int gpmi_gp_create_blocked(gpmi_ctx*, gpmi_comm* comm /* NULL: one rank */, int dtype, int d, int64_t n, const void* x,
                           int64_t block_rows, int stripe_blocks, gpmi_gp** out);
/* what the handle occupies: rows per block, stripes on this rank, bytes of factor storage on this rank */
GPMI_API int gpmi_gp_blocked_info(gpmi_gp*, int64_t* block_rows, int32_t* n_stripes, int64_t* factor_bytes);

/* ---- cov: replaces cov / cov! (src/kernels/kernels.jl:31-71) -------------
 * out is n1 x n2 col-major; x2 == NULL selects the symmetric X1 === X2 form. */
GPMI_API int gpmi_cov(gpmi_ctx*, const gpmi_kernel*, int dtype, int d, int64_t n1, const void* x1,
             int64_t n2, const void* x2, void* out);

/* ---- AbstractPDMat surface of the fitted factor (PDMats `\`, whiten!, logdet;
 *      src/GPE.jl:208,210, src/GP.jl:27,136, src/GPE.jl:162) ----------------
 * b_inout is n x nrhs col-major, overwritten with the result.               */
GPMI_API int gpmi_solve(gpmi_gp*, int64_t nrhs, void* b_inout);  /* (K + noise)^-1 b     */
GPMI_API int gpmi_whiten(gpmi_gp*, int64_t nrhs, void* b_inout); /* L^-1 b,  L = U'      */
/* out[i] = ((K + noise)^-1)_ii, n elements: what predict_LOO needs — inv(Σ) / diag in src/crossvalidation.jl:8-13
 * (SURVEY 8f rank 4).  n^3/3 flops on the device instead of the reference's dense inverse. */
GPMI_API int gpmi_inv_diag(gpmi_gp*, void* out);
GPMI_API int gpmi_logdet(gpmi_gp*, double* out);                 /* 2 sum log U_ii       */
/* U_out: n x n col-major with the upper factor in its upper triangle and zeros
 * below (== Cholesky(factors,'U',0), src/GPE.jl:60).                        */
GPMI_API int gpmi_factor_to_host(gpmi_gp*, void* U_out);
/* diag_out[i] = U_ii, n elements of the model's dtype: `diag(cholfactors(cK))` without moving the n x n factor
 * (what PDMats.logdet sums, GPE.jl:210; lets a caller re-derive logdet on the host). */
GPMI_API int gpmi_factor_diag(gpmi_gp*, void* diag_out);

/* ---- measurement hooks (bench.py; no reference counterpart) --------------
 * When enabled, every launch of a profiled kernel class is bracketed by HIP
 * events on the stream it is launched on.  gpmi_profile_get drains them.    */
enum gpmi_prof_class {
    GPMI_PROF_SYRK = 0,  /* Cholesky trailing update (MFMA)        */
    GPMI_PROF_COV = 1,   /* covariance assembly                    */
    GPMI_PROF_PANEL = 2, /* potf2 + trsm + in-panel update         */
    GPMI_PROF_SOLVE = 3, /* alpha solves, logdet                   */
    GPMI_PROF_PREDICT = 4,
    /* PHASES of one step of the blocked / sharded factorisation (csrc/blocked.cpp; one event pair per phase and step on the stream
     * the phase runs on: ~7 pairs per block step).  `launches` counts steps, `total_ms` sums the phase over the steps of the fits
     * since the last call: what a multi-GPU bench line reports as per-step chain / broadcast / solve / gather / update times.   */
    GPMI_PROF_STEP_U1 = 5,      /* next diagonal block + block column k+1 (update stream)                                       */
    GPMI_PROF_STEP_CHAIN = 6,   /* factor + explicit inverse of the next diagonal block (chain stream; its owner only)          */
    GPMI_PROF_STEP_BCAST = 7,   /* broadcast of that inverse (exchange stream)                                                  */
    GPMI_PROF_STEP_U2A = 8,     /* the part of the trailing update that hides chain + broadcast                                  */
    GPMI_PROF_STEP_SOLVE = 9,   /* next panel X LW' + copy back                                                                 */
    GPMI_PROF_STEP_GATHER = 10, /* all-gather of the solved panel (exchange stream)                                             */
    GPMI_PROF_STEP_U2B = 11,    /* the rest of the trailing update, which hides the gather                                      */
    GPMI_PROF_NCLASS = 12
};
/* on = 0: off; 1: every class; 2 + cls: the launches of class cls ONLY (a few dozen event pairs per fit for GPMI_PROF_SYRK,
 * against thousands for the panel kernels: what a timing harness leaves on inside its timed region); 64: every class but the
 * chain's tiny kernels; 65: the GPMI_PROF_STEP_* phases of blocked handles only.                                           */
GPMI_API int gpmi_profile_enable(gpmi_ctx*, int on);
/* launches, total milliseconds and algorithmic work (flops for SYRK/PANEL/
 * PREDICT, bytes for COV/SOLVE) accumulated since the last call for `cls`.  */
GPMI_API int gpmi_profile_get(gpmi_ctx*, int cls, int64_t* launches, double* total_ms, double* work);
/* algorithmic HBM bytes of the MFMA products of `cls` accumulated since the last call (every output entry read and
 * written once, the operand panels read once; K varies per launch since the two-level factorisation).             */
GPMI_API int gpmi_profile_get_bytes(gpmi_ctx*, int cls, double* bytes);
/* Peak-rate micro-benchmark of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32:
 * returns measured TFLOP/s with every SIMD issuing back-to-back MFMAs.      */
GPMI_API int gpmi_mfma_peak(gpmi_ctx*, int dtype, double* tflops_out);
/* Isolated timing of the trailing-update kernel  C[M x N] -= A[M x K] A[0:N, 0:K]'  on random
 * operands (lower != 0: SYRK tile set).  variant 0 is the 128 x 128 product kernel, 256 the
 * 256 x 128 one (csrc/update256.hip; needs lower != 0, falls back to 0 where it does not apply);
 * other values are the ablations of tools/gemm_ablate.py / tools/update256_ablate.py, compiled
 * only into a GPMI_TOOLS build of the library (make TOOLS=1; GPMI_EARG otherwise).  Returns
 * milliseconds per launch.                                                                     */
GPMI_API int gpmi_bench_gemm(gpmi_ctx*, int dtype, int64_t M, int64_t N, int64_t K, int lower, int variant, int iters,
                    double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* GPMI_H */
