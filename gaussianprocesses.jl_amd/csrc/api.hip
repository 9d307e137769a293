// api.hip — the C ABI of include/gpmi.h and the host-side drivers (blocked Cholesky, solves,
// predict).  All device work is enqueued on ctx->stream; every export returns after the
// results are on the host (the reference's callers are synchronous).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "chol.h"
#include "blocked.h"

using namespace gpmi;

namespace gpmi {

// ---------------------------------------------------------------------------------------------
// kernel descriptor -> device program
// ---------------------------------------------------------------------------------------------
static bool is_ard(int op) {
    return op == GPMI_K_SE_ARD || op == GPMI_K_MAT12_ARD || op == GPMI_K_MAT32_ARD || op == GPMI_K_MAT52_ARD ||
           op == GPMI_K_RQ_ARD;
}
static bool is_iso(int op) {
    return op == GPMI_K_SE_ISO || op == GPMI_K_MAT12_ISO || op == GPMI_K_MAT32_ISO || op == GPMI_K_MAT52_ISO ||
           op == GPMI_K_RQ_ISO;
}

int digest_kernel(const gpmi_kernel* k, int d, std::vector<unsigned char>* buf, std::string* err) {
    if (!k || !k->ops || !k->dims_off || !k->params || k->n_ops <= 0) {
        *err = "kernel descriptor: null field or empty program";
        return GPMI_EARG;
    }
    if (k->n_ops > GPMI_MAX_OPS) {
        *err = "kernel descriptor: more than GPMI_MAX_OPS nodes";
        return GPMI_EARG;
    }
    if (d <= 0 || d > (1 << 20)) {
        *err = "input dimension must be in 1 .. 2^20";
        return GPMI_EARG;
    }
    int64_t nleaf = 0;
    for (int o = 0; o < k->n_ops; ++o) nleaf += (k->ops[o] != GPMI_K_SUM && k->ops[o] != GPMI_K_PROD) ? 1 : 0;
    const int64_t w_count = nleaf * d;
    buf->assign((size_t)program_bytes(w_count), 0);
    DevProgram* out = reinterpret_cast<DevProgram*>(buf->data());
    out->n_ops = k->n_ops;
    out->d = d;
    out->w_count = w_count;
    int pp = 0, depth = 0, wcur = 0, hyp = 0;
    double kst[GPMI_MAX_OPS];
    int nst[GPMI_MAX_OPS];  // node index of each stack entry
    for (int64_t q = 0; q < w_count; ++q) out->pmtab()[q] = -1;
    for (int o = 0; o < k->n_ops; ++o) {
        const int op = k->ops[o];
        DevLeaf& lf = out->leaf[o];
        lf.op = op;
        if (op == GPMI_K_SUM || op == GPMI_K_PROD) {
            if (depth < 2) {
                *err = "kernel descriptor: SUM/PROD without two operands";
                return GPMI_EARG;
            }
            lf.right = nst[depth - 1];
            lf.left = nst[depth - 2];
            const double r = kst[--depth], l = kst[--depth];
            nst[depth] = o;
            kst[depth++] = (op == GPMI_K_SUM) ? l + r : l * r;
            continue;
        }
        if (!(is_ard(op) || is_iso(op) || op == GPMI_K_NOISE || op == GPMI_K_CONST)) {
            *err = "kernel descriptor: unknown op code";
            return GPMI_EARG;
        }
        const int d0 = k->dims_off[o], d1 = k->dims_off[o + 1];
        if (d1 < d0 || (d1 > d0 && !k->dims)) {
            *err = "kernel descriptor: bad dims_off";
            return GPMI_EARG;
        }
        const int nd = (d1 > d0) ? d1 - d0 : d;
        const int npar = is_ard(op) ? nd + 1 + (op == GPMI_K_RQ_ARD) : is_iso(op) ? 2 + (op == GPMI_K_RQ_ISO) : 1;
        if (pp + npar > k->n_params) {
            *err = "kernel descriptor: params shorter than the program needs";
            return GPMI_EARG;
        }
        const double* par = k->params + pp;
        pp += npar;
        lf.woff = wcur;
        double* w = out->wtab() + wcur;
        wcur += d;
        for (int z = 0; z < nd; ++z) {
            const int kk = (d1 > d0) ? k->dims[d0 + z] : z;
            if (kk < 0 || kk >= d) {
                *err = "kernel descriptor: active dim out of range";
                return GPMI_EARG;
            }
            w[kk] += is_ard(op) ? par[z] : 1.0;
            if (is_ard(op)) out->pmtab()[lf.woff + kk] = (int32_t)z;
        }
        lf.poff = hyp;
        lf.nd = nd;
        hyp += npar;
        if (op == GPMI_K_NOISE || op == GPMI_K_CONST) {
            lf.s2 = par[0];
            if (op == GPMI_K_NOISE) out->has_noise_leaf = 1;
        } else if (is_iso(op)) {
            lf.s2 = par[1];
            if (op == GPMI_K_RQ_ISO) {
                lf.p1 = par[2];
                lf.p0 = 1.0 / (2.0 * par[2] * par[0]);
            } else {
                lf.p0 = 1.0 / par[0];
            }
        } else {
            lf.s2 = par[nd];
            if (op == GPMI_K_RQ_ARD) {
                lf.p1 = par[nd + 1];
                lf.p0 = 0.5 / par[nd + 1];
            } else {
                lf.p0 = 1.0;
            }
        }
        if (depth >= 6) {
            *err = "kernel descriptor: expression stack deeper than 6";
            return GPMI_EARG;
        }
        nst[depth] = o;
        kst[depth++] = lf.s2;  // every leaf evaluates to s2 at coincident points
    }
    if (depth != 1 || pp != k->n_params) {
        *err = "kernel descriptor: malformed program (stack/params left over)";
        return GPMI_EARG;
    }
    out->kdiag = kst[0];
    out->n_hyp = hyp;
    // multi-leaf programs of shallow depth get the specialised interior-tile kernel (cov.hip)
    out->fast_class = -1;
    if (k->n_ops > 1) {
        int dep = 0, maxdep = 0, cls = 0;
        for (int o = 0; o < k->n_ops; ++o) {
            const int op = k->ops[o];
            if (op == GPMI_K_SUM || op == GPMI_K_PROD) {
                --dep;
            } else {
                ++dep;
                if (op == GPMI_K_RQ_ISO || op == GPMI_K_RQ_ARD) cls |= 1;
                if (op == GPMI_K_NOISE) cls |= 2;
            }
            maxdep = std::max(maxdep, dep);
        }
        if (maxdep <= 3) out->fast_class = cls;
    }
    return GPMI_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling scope
// ---------------------------------------------------------------------------------------------
static hipEvent_t take_event(gpmi_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(gpmi_ctx* ctx, int cls, double work, double bytes, bool attach_to_launch, bool chain_kernel) : c(ctx), attach(attach_to_launch) {
    if (!c->prof_on || (c->prof_only >= 0 && cls != c->prof_only) || (chain_kernel && c->prof_skip_chain)) return;
    if (c->prof_phases_only && cls < GPMI_PROF_STEP_U1) return;
    ProfRec r;
    r.a = take_event(c);
    r.b = take_event(c);
    r.cls = cls;
    r.work = work;
    r.bytes = bytes;
    r.attached = attach;
    if (attach) {  // the launch itself carries the two events (start / stop of the dispatch): nothing else enters the queue
        c->attach_a = r.a;
        c->attach_b = r.b;
    } else {
        hipEventRecord(r.a, c->stream);
    }
    idx = (int)c->prof.size();
    c->prof.push_back(r);
}
ProfScope::~ProfScope() {
    if (idx < 0) return;
    if (attach) {
        if (c->attach_a) {  // nothing was launched (empty product): give the pair a zero-length interval
            hipEventRecord(c->prof[idx].a, c->stream);
            hipEventRecord(c->prof[idx].b, c->stream);
            c->attach_a = c->attach_b = nullptr;
        }
        return;
    }
    hipEventRecord(c->prof[idx].b, c->stream);
}

static int drain_profile(gpmi_ctx* c) {
    if (c->prof.empty()) return GPMI_OK;
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& r : c->prof) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        c->prof_n[r.cls] += 1;
        c->prof_ms[r.cls] += ms;
        c->prof_work[r.cls] += r.work;
        c->prof_bytes[r.cls] += r.bytes;
        if (r.attached) {
            // A dispatch's start / stop events keep the kernel command alive, and the command keeps its argument buffers: while such
            // an event sits in the pool, hipFree of the factor does not return its memory (measured under ROCm 7.2's runtime: a
            // destroyed N = 200 000 fp32 model left 165 GB held until the events were recorded again).  These are few (one pair per
            // trailing update): destroy them instead of pooling them.
            hipEventDestroy(r.a);
            hipEventDestroy(r.b);
        } else {
            c->ev_pool.push_back(r.a);
            c->ev_pool.push_back(r.b);
        }
    }
    c->prof.clear();
    return GPMI_OK;
}

// Both look-ahead stream sets are created WITH the context: the CU-masked pair here, right after the context's own stream (the
// priority side stream comes before it: gpmi_ctx_create).
// words of a 256-bit CU mask: bit b = CU (b / 8) of XCC (b % 8) (tools/cumask_probe.hip) — half p of EVERY XCD is words [4 p, 4 p + 4)
static void partition_mask(int part, uint32_t m[8]) {
    for (int w = 0; w < 8; ++w) m[w] = (part == 0 || w / 4 == part - 1) ? ~0u : 0u;
}

static void create_lookahead_streams(gpmi_ctx* c) {
    if (!c->mask_ok) return;
    // one CU per XCD for the chain: mask bit k * 33 (k = 0..7) lands on XCC k, one CU each (tools/cumask_probe.hip,
    // profiles/r01_coresidency_cumask_probe.log); the update stream gets the other 248
    uint32_t side_m[8] = {0}, upd_m[8], part_m[8];
    partition_mask(c->cu_part, part_m);
    // (a CU partition reserves its own first CU of every XCD: bits 128 (p - 1) + k)
    for (int k = 0; k < 8; ++k) {
        const int b = c->cu_part ? 128 * (c->cu_part - 1) + k : k * 33;
        side_m[b / 32] |= 1u << (b % 32);
    }
    for (int w = 0; w < 8; ++w) upd_m[w] = part_m[w] & ~side_m[w];
    if (hipExtStreamCreateWithCUMask(&c->side_masked, 8, side_m) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&c->upd_stream, 8, upd_m) != hipSuccess) {
        (void)hipGetLastError();
        for (hipStream_t* st : {&c->upd_stream, &c->side_masked})
            if (*st) {
                (void)hipStreamDestroy(*st);
                *st = nullptr;
            }
        c->mask_ok = false;  // not on this device / runtime: free slots everywhere
    }
}

int set_lookahead_mode(gpmi_ctx* c, bool whole) {
    const int want = (whole && c->mask_ok && c->side_masked && c->upd_stream) ? 1 : 0;
    c->reserved_cus = want == 1 ? 8 : 0;
    c->la_mode = want;
    return want;
}

// ---------------------------------------------------------------------------------------------
// drivers
// ---------------------------------------------------------------------------------------------
int upload_program(gpmi_ctx* c, const gpmi_kernel* k, int d) {
    int rc = digest_kernel(k, d, &c->prog_buf, &c->err);
    if (rc != GPMI_OK) return rc;
    const int64_t bytes = (int64_t)c->prog_buf.size();
    if (bytes > c->prog_cap) {  // header + weight tables: grown to the largest program seen
        GPMI_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_prog) (void)hipFree(c->d_prog);
        if (c->h_prog) (void)hipHostFree(c->h_prog);
        c->d_prog = c->h_prog = nullptr;
        c->prog_cap = 0;
        const int64_t cap = std::max<int64_t>(bytes, program_bytes(1024));
        GPMI_HIP(c, hipMalloc(&c->d_prog, (size_t)cap));
        GPMI_HIP(c, hipHostMalloc(&c->h_prog, (size_t)cap));
        c->prog_cap = cap;
    }
    memcpy(c->h_prog, c->prog_buf.data(), (size_t)bytes);
    GPMI_HIP(c, hipMemcpyAsync(c->d_prog, c->h_prog, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    // h_prog is reused by the next call: make sure the copy has left the staging buffer
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    return GPMI_OK;
}

// ---------------------------------------------------------------------------------------------
template <typename T>
static int fit_t(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, int64_t n_noise, const void* y_minus_mu,
                 double* mll_out, void* alpha_out, int64_t* info_out) {
    gpmi_ctx* c = gp->ctx;
    const int64_t n = gp->n, npad = gp->npad, ld = gp->ld;
    T* A = (T*)gp->A;
    gp->fitted = false;
    la_reset(c);
    int rc = upload_program(c, k, gp->d);
    if (rc != GPMI_OK) return rc;

    double nugget = 0.0, min_nugget = 0.0;
    const double* d_noise = nullptr;
    if (n_noise == 1) {
        nugget = min_nugget = exp(2.0 * log_noise[0]);  // GPE.jl:173
    } else {
        std::vector<double> nv((size_t)n);
        for (int64_t i = 0; i < n; ++i) nv[(size_t)i] = exp(2.0 * log_noise[i]);  // GPE.jl:181-183
        min_nugget = *std::min_element(nv.begin(), nv.end());
        if (!gp->noise) GPMI_HIP(c, hipMalloc(&gp->noise, (size_t)n * sizeof(double)));
        GPMI_HIP(c, hipMemcpy(gp->noise, nv.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        d_noise = gp->noise;
    }
    // y - mu -> device (zero padded) and into the carried row npad of A
    GPMI_HIP(c, hipMemsetAsync(gp->ymu, 0, (size_t)npad * sizeof(T), c->stream));
    GPMI_HIP(c, hipMemcpyAsync(gp->ymu, y_minus_mu, (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    GPMI_HIP(c, hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));

    // (Measured and not kept, GPU call P: cov! split in two launches so that the first diagonal block — the one chain with nothing to hide
    //  behind — is factored on the side stream under the second: correct, but 701 against 697 ms at N = 50 000 and 70.5 against 68.6 at
    //  N = 20 000: beside a launch that saturates every CU with short workgroups the chain's single-workgroup kernels crawl, and what was
    //  3.4 ms alone costs more than it hides.  profiles/r04_p_first_block_under_cov.log)
    launch_cov<T>(c, (const T*)gp->x, n, (const T*)gp->x, n, gp->d, A, ld, npad, npad,
                  COV_LOWER | COV_NUGGET | COV_PAD_IDENTITY, nugget, d_noise);
    GPMI_HIP(c, hipMemcpyAsync(A + npad * ld, gp->ymu, (size_t)npad * sizeof(T), hipMemcpyDeviceToDevice, c->stream));

    // When the noise is tiny next to the prior variance the matrix is regularised like FITC's Kuu and the products with
    // stored block inverses need the refinement step to keep LAPACK's accuracy (tools/small_noise_check.py: at
    // logNoise = -8 the mll error is -0.46 without it, -4e-6 with it, LAPACK's own +0.04).  The usual case does not pay.
    const bool refine = min_nugget < 1e-5 * c->h_prog->kdiag || c->refine_default;
    c->refine_solves = refine;
    // the super-block inverses the factorisation builds are kept for predict_f / the gradient / predict_LOO (chol.h SuperStore)
    SuperStore<T> store;
    gp->sup_parts.clear();
    {
        const int64_t w0 = super_width_max(c, npad);
        if (w0 > NB && !refine && c->super_inverse) {
            const int rc_s = grow(c, &gp->supinv, &gp->supinv_cap, npad * (w0 + IB) * (int64_t)sizeof(T));
            if (rc_s) return rc_s;
            store.buf = (T*)gp->supinv;
            store.cap = npad * (w0 + IB);
            store.parts = &gp->sup_parts;
        }
    }
    const int rc_chol = cholesky_lower<T>(c, A, ld, (T*)gp->linv, (T*)gp->invdiag, npad, 1, c->d_info, store.buf ? &store : nullptr);
    c->refine_solves = c->refine_default;
    if (rc_chol) return rc_chol;

    {
        ProfScope ps(c, GPMI_PROF_SOLVE, (double)npad * (double)npad * 0.5 * sizeof(T));
        launch_linv256<T>(c, A, ld, (const T*)gp->linv, (T*)gp->linv256, npad, c->d_info);
        for (int64_t k0 = (npad - 1) / NB * NB; k0 >= 0; k0 -= NB)
            launch_bsolve256<T>(c, A + k0 * ld, ld, k0, (int)std::min<int64_t>(NB, npad - k0),
                                (const T*)gp->linv256 + (k0 / NB) * NB * NB, A + npad * ld, (T*)gp->alpha);
        launch_finalize<T>(c, A, ld, n, (const T*)gp->ymu, (const T*)gp->alpha, c->d_scal);
    }
    int h_info = 0;
    GPMI_HIP(c, hipMemcpyAsync(c->h_scal, c->d_scal, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GPMI_HIP(c, hipMemcpyAsync(&h_info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    GPMI_HIP(c, hipGetLastError());
    if (h_info < 0) {  // chain.hip: a dependency wait ran into its bound (a workgroup of the chain launch never came up)
        c->err = "chain kernel: a dependency wait timed out (GPMI_CHAIN=0 selects the multi-launch chain)";
        return GPMI_EDEVICE;
    }
    if (info_out) *info_out = h_info;
    if (h_info != 0) {
        c->err = "matrix is not positive definite; Cholesky factorization failed";
        return GPMI_ENOTPD;
    }
    gp->mll = c->h_scal[0];
    gp->logdet = c->h_scal[1];
    gp->fitted = true;
    if (mll_out) *mll_out = gp->mll;
    if (alpha_out) GPMI_HIP(c, hipMemcpy(alpha_out, gp->alpha, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    return GPMI_OK;
}

// gradient of the mll with respect to the kernel hyper-parameters and log-noise (update_dmll!, GPE.jl:298-324)
// the gradient / predict_LOO scratch: two more npad x ld matrices (L^-T rows and K^-1), kept until gpmi_gp_destroy.  They
// triple the model's footprint — N = 150 000 fp64 would need 540 GB — so failure here gets a message of its own.
static int alloc_grad_scratch(gpmi_gp* gp, size_t bytes) {
    gpmi_ctx* c = gp->ctx;
    for (void** p : {&gp->g1, &gp->g2}) {
        if (*p) continue;
        if (hipMalloc(p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            *p = nullptr;
            c->err = "update_dmll / predict_LOO need two more N x N device matrices (" + std::to_string(2 * bytes >> 30) +
                     " GiB at this size) next to the factor: out of device memory";
            return GPMI_EDEVICE;
        }
    }
    return GPMI_OK;
}

// the column ranges of the factor that still have their explicit super-block inverse from the last fit (chol.h)
template <typename T>
static std::vector<WhitenSeg<T>> whiten_segments(const gpmi_gp* gp) {
    std::vector<WhitenSeg<T>> v;
    if (gp->ctx->whiten_by_super_inverse)
        for (const auto& p : gp->sup_parts) v.push_back(WhitenSeg<T>{p.ks, p.ks + p.w, (const T*)gp->supinv + p.off});
    return v;
}

template <typename T>
static int grad_t(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, double* dkern_out, double* dnoise_out) {
    gpmi_ctx* c = gp->ctx;
    const int64_t n = gp->n, npad = gp->npad, ld = gp->ld;
    const T* A = (const T*)gp->A;
    la_reset(c);
    int rc = upload_program(c, k, gp->d);
    if (rc != GPMI_OK) return rc;
    const int n_hyp = c->h_prog->n_hyp;
    const size_t bytes = (size_t)(npad * ld) * sizeof(T);
    if (const int rc_g = alloc_grad_scratch(gp, bytes)) return rc_g;
    T* G1 = (T*)gp->g1;
    T* G2 = (T*)gp->g2;
    const int64_t nt = (n + 63) / 64;
    const int64_t need = nt * nt * (n_hyp + 1) * (int64_t)sizeof(double);
    if (gp->gpart_cap < need) {
        if (gp->gpart) hipFree(gp->gpart);
        gp->gpart = nullptr;
        gp->gpart_cap = 0;
        GPMI_HIP(c, hipMalloc(&gp->gpart, (size_t)need));
        gp->gpart_cap = need;
    }
    {
        ProfScope ps(c, GPMI_PROF_SOLVE, 2.0 * (double)npad * (double)npad * (double)npad / 3.0);
        // rows of L^-T: the whiten sequence applied to an identity (in G2, consumed); row i is zero left of column i,
        // so block k only has to process rows < kend
        launch_set_identity<T>(c, G2, ld, npad);
        const auto segs = whiten_segments<T>(gp);
        whiten_rows_inv<T>(c, A, ld, (const T*)gp->linv256, npad, G2, ld, G1, ld, [](int64_t kend) { return kend; }, &segs);
        // K^-1 = L^-T L^-1 = G1 G1'  (lower tiles).  Small n: one product whose K loop starts at the tile's first row (G1 is
        // upper triangular by rows).  Large n: the K dimension in chunks of 2048 columns — a tile's K loop over the whole
        // row length streams two 128 x n panels (25 MB each at n = 50 000) through a 4 MB L2, chunked it is the K = 2048
        // update of the factorisation: chunk c first WRITES the rows that start inside it (rows [k0, k1), K from the tile's
        // first row), then accumulates onto the k0 x k0 block above them.  The accumulation subtracts, so G2 holds -K^-1.
        const int64_t WK = c->grad_chunk;
        const bool chunked = WK > 0 && npad >= 4 * WK;
        if (!chunked) {
            launch_gemm_shape<T>(c, G2, ld, G1, ld, G1, ld, npad, npad, npad, TileShape{0, 0, 1, 0, 1, 0}, nullptr,
                                 GEMM_OVERWRITE | GEMM_KSTART_ROW);
        } else {
            for (int64_t k0 = 0; k0 < npad; k0 += WK) {
                const int64_t kw = std::min<int64_t>(WK, npad - k0), k1 = k0 + kw;
                launch_gemm_shape<T>(c, G2 + k0 * ld, ld, G1 + k0 * ld + k0, ld, G1 + k0, ld, kw, k1, kw,
                                     TileShape{0, 0, 1, (int)(k0 / GEMM_BM), 1, 0}, nullptr, GEMM_OVERWRITE | GEMM_NEGOUT | GEMM_KSTART_ROW);
                if (k0 > 0)
                    launch_gemm_shape<T>(c, G2, ld, G1 + k0, ld, G1 + k0, ld, k0, k0, kw, TileShape{0, 0, 1, 0, 1, 0}, nullptr, GEMM_AUX);
            }
        }
        const int64_t nblocks = launch_dmll<T>(c, (const T*)gp->x, n, gp->d, (const T*)gp->alpha, G2, ld, gp->gpart, n_hyp, chunked);
        launch_reduce_partials(c, gp->gpart, nblocks, n_hyp + 1, (double*)gp->g1);  // g1 is free again: result vector
    }
    std::vector<double> h((size_t)n_hyp + 1);
    GPMI_HIP(c, hipMemcpyAsync(h.data(), gp->g1, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    GPMI_HIP(c, hipGetLastError());
    for (int p = 0; p < n_hyp; ++p) dkern_out[p] = h[(size_t)p];
    if (dnoise_out) *dnoise_out = exp(2.0 * log_noise[0]) * h[(size_t)n_hyp];  // GPE.jl:273-275
    return GPMI_OK;
}

// diag((K + noise)^-1) = squared row norms of L^-T (whitened identity): what predict_LOO needs (crossvalidation.jl:8-13)
template <typename T>
static int inv_diag_t(gpmi_gp* gp, void* out) {
    gpmi_ctx* c = gp->ctx;
    const int64_t n = gp->n, npad = gp->npad, ld = gp->ld;
    la_reset(c);
    const size_t bytes = (size_t)(npad * ld) * sizeof(T);
    if (const int rc_g = alloc_grad_scratch(gp, bytes)) return rc_g;
    T* G1 = (T*)gp->g1;
    T* G2 = (T*)gp->g2;
    launch_set_identity<T>(c, G2, ld, npad);
    const auto segs = whiten_segments<T>(gp);
    whiten_rows_inv<T>(c, (const T*)gp->A, ld, (const T*)gp->linv256, npad, G2, ld, G1, ld, [](int64_t kend) { return kend; }, &segs);
    // row i of G1 is written from the first column of its own NB-block on (zero left of that in exact arithmetic)
    launch_row_sumsq<T>(c, G1, ld, n, npad, NB, G2);  // G2 is free again: result vector
    GPMI_HIP(c, hipMemcpyAsync(out, G2, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    GPMI_HIP(c, hipGetLastError());
    return GPMI_OK;
}

int grow(gpmi_ctx* c, void** p, int64_t* cap, int64_t need_bytes) {
    if (*cap >= need_bytes) return GPMI_OK;
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    GPMI_HIP(c, hipMalloc(p, (size_t)need_bytes));
    *cap = need_bytes;
    return GPMI_OK;
}

template <typename T>
static int predict_t(gpmi_gp* gp, const gpmi_kernel* k, int64_t P, const void* xpred, const void* mean_pred,
                     int full_cov, void* mu_out, void* var_out) {
    gpmi_ctx* c = gp->ctx;
    const int64_t n = gp->n, npad = gp->npad, ld = gp->ld;
    const int d = gp->d;
    const T* A = (const T*)gp->A;
    la_reset(c);
    int rc = upload_program(c, k, d);
    if (rc != GPMI_OK) return rc;
    const double kdiag = c->h_prog->kdiag;

    int64_t rows_bytes = 2 * P * ld * (int64_t)sizeof(T);  // K*' rows and, out of place, their whitened image
    if ((rc = grow(c, &gp->rows, &gp->rows_cap, rows_bytes)) != GPMI_OK) return rc;
    if ((rc = grow(c, &gp->xp, &gp->xp_cap, P * d * (int64_t)sizeof(T))) != GPMI_OK) return rc;
    if ((rc = grow(c, &gp->small, &gp->small_cap, std::max<int64_t>(3 * P, npad) * (int64_t)sizeof(T))) != GPMI_OK) return rc;
    T* R = (T*)gp->rows;
    T* V = R + P * ld;
    T* xp = (T*)gp->xp;
    T* d_mean = (T*)gp->small;
    T* d_mu = d_mean + P;
    T* d_var = d_mu + P;
    GPMI_HIP(c, hipMemcpyAsync(xp, xpred, (size_t)(P * d) * sizeof(T), hipMemcpyHostToDevice, c->stream));
    GPMI_HIP(c, hipMemcpyAsync(d_mean, mean_pred, (size_t)P * sizeof(T), hipMemcpyHostToDevice, c->stream));

    {
        ProfScope ps(c, GPMI_PROF_PREDICT, (double)npad * (double)npad * (double)P);
        // K*' (P x npad, one test point per row): cov(k, xtrain, xpred)', GP.jl:44
        launch_cov<T>(c, xp, P, (const T*)gp->x, n, d, R, ld, P, npad, 0, 0.0, nullptr);
        launch_row_gemv<T>(c, R, ld, P, n, (const T*)gp->alpha, d_mean, d_mu);  // mu = mx + Kfx' alpha, GP.jl:26
        // Lck = whiten!(Kff, Kfx), GP.jl:27
        const auto segs = whiten_segments<T>(gp);
        // (Measured and not kept, round 5 call F: the two halves of the test points whitened on two streams so that one half's launches fill the
        //  partial last round of the other's updates — 44.6 against 44.2 ms at N = 50 000, 8.9 against 8.7 at N = 20 000: no gain, each half
        //  re-reads the factor's panels.  profiles/r05_f_predict_split.log)
        whiten_rows_inv<T>(c, A, ld, (const T*)gp->linv256, npad, R, ld, V, ld, [P](int64_t) { return P; }, &segs);
        if (!full_cov) launch_row_var<T>(c, V, ld, P, npad, kdiag, d_var);
    }
    GPMI_HIP(c, hipMemcpyAsync(mu_out, d_mu, (size_t)P * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    if (!full_cov) {
        GPMI_HIP(c, hipMemcpyAsync(var_out, d_var, (size_t)P * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        GPMI_HIP(c, hipStreamSynchronize(c->stream));
        GPMI_HIP(c, hipGetLastError());
        return GPMI_OK;
    }
    // full covariance: Kpred - Lck'Lck (GP.jl:45,51-54); no clamp in this branch
    const int64_t ldp = (P + 63) / 64 * 64;
    T* Kpp = nullptr;
    GPMI_HIP(c, hipMalloc(&Kpp, (size_t)(P * ldp) * sizeof(T)));
    launch_cov<T>(c, xp, P, xp, P, d, Kpp, ldp, P, ldp, 0, 0.0, nullptr);
    launch_gemm_nt<T>(c, Kpp, ldp, V, ld, V, ld, P, P, npad, 0, nullptr);
    hipError_t e = hipMemcpy2DAsync(var_out, (size_t)P * sizeof(T), Kpp, (size_t)ldp * sizeof(T), (size_t)P * sizeof(T),
                                    (size_t)P, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(Kpp);
    GPMI_HIP(c, e);
    GPMI_HIP(c, hipGetLastError());
    return GPMI_OK;
}

template <typename T>
static int cov_t(gpmi_ctx* c, const gpmi_kernel* k, int d, int64_t n1, const void* x1, int64_t n2, const void* x2,
                 void* out) {
    int rc = upload_program(c, k, d);
    if (rc != GPMI_OK) return rc;
    const bool sym = (x2 == nullptr);
    if (sym) n2 = n1;
    // out is n1 x n2 col-major == row-major [n2][n1]: rows index x2, columns index x1
    const int64_t ldc = (n1 + 63) / 64 * 64;
    T *dx1 = nullptr, *dx2 = nullptr, *dC = nullptr;
    hipError_t e = hipMalloc(&dx1, (size_t)(n1 * d) * sizeof(T));
    if (e == hipSuccess && !sym) e = hipMalloc(&dx2, (size_t)(n2 * d) * sizeof(T));
    if (e == hipSuccess) e = hipMalloc(&dC, (size_t)(n2 * ldc) * sizeof(T));
    if (e == hipSuccess) e = hipMemcpyAsync(dx1, x1, (size_t)(n1 * d) * sizeof(T), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && !sym) e = hipMemcpyAsync(dx2, x2, (size_t)(n2 * d) * sizeof(T), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_cov<T>(c, sym ? dx1 : dx2, n2, dx1, n1, d, dC, ldc, n2, ldc, 0, 0.0, nullptr);
        e = hipMemcpy2DAsync(out, (size_t)n1 * sizeof(T), dC, (size_t)ldc * sizeof(T), (size_t)n1 * sizeof(T), (size_t)n2,
                             hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    hipFree(dx1);
    hipFree(dx2);
    hipFree(dC);
    GPMI_HIP(c, e);
    return GPMI_OK;
}

// b (n x nrhs col-major) -> rows of R, whiten (and optionally back-substitute), copy back
template <typename T>
static int solve_t(gpmi_gp* gp, int64_t nrhs, void* b_inout, bool backward) {
    gpmi_ctx* c = gp->ctx;
    const int64_t n = gp->n, npad = gp->npad, ld = gp->ld;
    const T* A = (const T*)gp->A;
    int rc;
    if ((rc = grow(c, &gp->rows, &gp->rows_cap, nrhs * ld * (int64_t)sizeof(T))) != GPMI_OK) return rc;
    if ((rc = grow(c, &gp->small, &gp->small_cap, npad * (int64_t)sizeof(T))) != GPMI_OK) return rc;
    T* R = (T*)gp->rows;
    GPMI_HIP(c, hipMemsetAsync(R, 0, (size_t)(nrhs * ld) * sizeof(T), c->stream));
    GPMI_HIP(c, hipMemcpy2DAsync(R, (size_t)ld * sizeof(T), b_inout, (size_t)n * sizeof(T), (size_t)n * sizeof(T),
                                 (size_t)nrhs, hipMemcpyHostToDevice, c->stream));
    whiten_rows<T>(c, A, ld, (const T*)gp->linv, npad, R, ld, nrhs);
    if (backward) {
        T* tmp = (T*)gp->small;
        for (int64_t r = 0; r < nrhs; ++r) {
            for (int64_t j0 = npad - IB; j0 >= 0; j0 -= IB)
                launch_bsolve_step<T>(c, A + j0 * ld, ld, j0, (const T*)gp->linv + (j0 / IB) * IB * IB, R + r * ld, tmp);
            GPMI_HIP(c, hipMemcpyAsync(R + r * ld, tmp, (size_t)npad * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        }
    }
    GPMI_HIP(c, hipMemcpy2DAsync(b_inout, (size_t)n * sizeof(T), R, (size_t)ld * sizeof(T), (size_t)n * sizeof(T),
                                 (size_t)nrhs, hipMemcpyDeviceToHost, c->stream));
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    GPMI_HIP(c, hipGetLastError());
    return GPMI_OK;
}

}  // namespace gpmi

// =============================================================================================
// C ABI
// =============================================================================================
// ---- two-level sharded factorisation building blocks (gpmi_dev_super_*) ----
template <typename T>
static int super_factor_t(gpmi_ctx* c, T* blk, int64_t ld, int64_t w, T* linv, T* invdiag, T* lw, int64_t pivot_base) {
    int rc;  // scratch of build_super_inverse: the transposed inverse, the packed 256-inverses, one product buffer
    if ((rc = grow(c, &c->sup_lwt, &c->sup_lwt_cap, w * w * (int64_t)sizeof(T)))) return rc;
    if ((rc = grow(c, &c->sup_l256, &c->sup_l256_cap, w * NB * (int64_t)sizeof(T)))) return rc;
    if ((rc = grow(c, &c->sup_ut, &c->sup_ut_cap, std::max<int64_t>(1, (w / 2) * (w / 2)) * (int64_t)sizeof(T)))) return rc;
    // factor_diag_block / build_super_inverse address the block as (k, k) of a matrix and the inverses by global index:
    // shift the bases so that k = pivot_base lands on the caller's buffers (nothing outside them is dereferenced)
    const int64_t k = pivot_base;
    if (launch_chain_block<T>(c, blk, ld, w, linv, invdiag, lw, w, c->d_info, pivot_base)) return GPMI_OK;  // one persistent launch (chain.hip)
    T* Av = blk - (k * ld + k);
    T* linv_v = linv - (k / IB) * IB * IB;
    T* invd_v = invdiag - k;
    factor_diag_block<T>(c, Av, ld, linv_v, invd_v, k, w, c->d_info);
    build_super_inverse<T>(c, Av, ld, linv_v, k, w, lw, w, c->d_info);
    return GPMI_OK;
}
namespace gpmi {
template <typename T>
int super_factor_block(gpmi_ctx* c, T* blk, int64_t ld, int64_t w, T* linv, T* invdiag, T* lw, int64_t pivot_base) {
    return super_factor_t<T>(c, blk, ld, w, linv, invdiag, lw, pivot_base);
}
template int super_factor_block<double>(gpmi_ctx*, double*, int64_t, int64_t, double*, double*, double*, int64_t);
template int super_factor_block<float>(gpmi_ctx*, float*, int64_t, int64_t, float*, float*, float*, int64_t);
}  // namespace gpmi

// GPMI_EARG with a message of its own (gpmi_last_error must never return the text of an unrelated earlier failure)
static int earg(gpmi_ctx* c, const char* msg) {
    if (c) c->err = msg;
    return GPMI_EARG;
}

extern "C" {

const char* gpmi_version(void) { return "gpmi 0.3 (gfx950)"; }

static int create_one_context(int dev, gpmi_ctx** out);

int gpmi_ctx_create(int n_devices, const int* device_ids, gpmi_ctx** out) {
    if (!out) return GPMI_EARG;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || (n_devices > 1 && !device_ids)) return GPMI_EARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GPMI_EDEVICE;  // no GPU: no fallback
    for (int i = 0; i < n_devices; ++i) {
        const int code = device_ids ? device_ids[i] : 0;
        const int dev = code & 255, part = code >> 8;  // id + 256 (1 + p): CU partition p of device id (include/gpmi.h)
        if (code < 0 || dev >= count || part > 2) return GPMI_EARG;
    }
    gpmi_ctx* c = nullptr;
    int rc = create_one_context(device_ids ? device_ids[0] : 0, &c);
    if (rc != GPMI_OK) return rc;
    if (n_devices > 1 && (rc = group_create(c, n_devices, device_ids)) != GPMI_OK) {
        gpmi_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return GPMI_OK;
}

static int create_one_context(int dev_code, gpmi_ctx** out) {
    *out = nullptr;
    gpmi_ctx* c = new gpmi_ctx();
    const int dev = dev_code & 255;
    c->device = dev;
    c->cu_part = dev_code >> 8;
    if (c->cu_part) {
        // A CU PARTITION of the device: every stream of this context is confined to 16 of the 32 CUs of every XCD (the same 8-XCD
        // round-robin, half the width), so two such contexts run side by side without sharing a compute unit — the software
        // stand-in for the driver's compute partitioning (CPX was refused on the leased device: profiles/r04_a_cpx_refused.log).
        uint32_t pm[8];
        partition_mask(c->cu_part, pm);
        if (hipSetDevice(dev) != hipSuccess || hipExtStreamCreateWithCUMask(&c->side_stream, 8, pm) != hipSuccess ||
            hipExtStreamCreateWithCUMask(&c->own_stream, 8, pm) != hipSuccess) {
            (void)hipGetLastError();
            c->stream = c->own_stream;
            gpmi_ctx_destroy(c);
            return GPMI_EDEVICE;
        }
    } else
    // Stream creation ORDER matters on this runtime (profiles/r03_d_stream_order.log, r03_e_*): priority side stream, the
    // context's own stream, then the two CU-masked streams.  The masked streams must directly follow the own stream (created
    // after the priority stream, or lazily after a large factorisation, they cost N = 20 000 ten ms per step), and the priority
    // stream must not be the fourth (as the fourth it serialised behind the own stream: 940 instead of 711 ms at N = 50 000).
    if (hipSetDevice(dev) == hipSuccess) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = greatest priority (numerically lowest)
        if (hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            c->side_stream = nullptr;
        }
    }
    if (hipSetDevice(dev) != hipSuccess || (!c->own_stream && hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) ||
        hipMalloc(&c->d_prog, (size_t)program_bytes(1024)) != hipSuccess ||
        hipHostMalloc(&c->h_prog, (size_t)program_bytes(1024)) != hipSuccess || hipMalloc(&c->d_info, sizeof(int)) != hipSuccess ||
        hipMalloc(&c->d_scal, 8 * sizeof(double)) != hipSuccess ||
        hipHostMalloc(&c->h_scal, 8 * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->d_queue, (64 + 4 * 1024) * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->d_queue, 0, (64 + 4 * 1024) * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_queue_side, 64 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->d_queue_side, 0, 64 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->chain_sync, (size_t)chain_sync_bytes(c->chain_nb_max)) != hipSuccess ||
        // the whole area, INCLUDING the never-reset `started` word behind the part every launch zeroes: it counts up from 0 together with
        // chain_started_expect (a recycled allocation would hand chain_wait_kernel a stale count: no wait, or a full timeout, per update)
        hipMemset(c->chain_sync, 0, (size_t)chain_sync_bytes(c->chain_nb_max)) != hipSuccess) {
        c->stream = c->own_stream;
        gpmi_ctx_destroy(c);
        return GPMI_EDEVICE;
    }
    c->stream = c->own_stream;
    c->prog_cap = program_bytes(1024);
    memset(c->h_prog, 0, sizeof(DevProgram));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    const int device_cus = c->num_cus;
    if (c->cu_part) c->num_cus = device_cus / 2;
    // look-ahead Cholesky: a high-priority side stream for the next panel's serial chain, and how many of the chip's
    // workgroup slots the main trailing-update launch leaves free for it (GPMI_LOOKAHEAD=0 switches it off)
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = greatest priority (numerically lowest)
        const char* cm = getenv("GPMI_CUMASK");
        c->mask_ok = !(cm && atoi(cm) == 0) && device_cus == 256;
        (void)hi;
        create_lookahead_streams(c);
        c->lookahead_slots = 16;  // 8 in round 1 (a 7-launch chain per panel); the super-block factorisation has launches of up to 28 workgroups
        if (const char* e = getenv("GPMI_LOOKAHEAD")) c->lookahead_slots = atoi(e) / 8 * 8;
        if (const char* e = getenv("GPMI_LOOKAHEAD_MIN")) {  // given as a trailing size, as in round 1
            const double t = atof(e);
            c->lookahead_min_tiles = c->lookahead_min_tiles_masked = (int64_t)(0.5 * t * t / (GEMM_BM * GEMM_BN));
        }
    }
    // ---- environment, read once per context.  Five knobs steer the factorisation (defaults are the measured optimum):
    //   GPMI_SUPER=a,b,c       rows remaining from which super-panels of 512 / 1024 / 2048 columns are used
    //   GPMI_LOOKAHEAD=slots   workgroup slots left free beside the update (0 = serial factorisation)
    //   GPMI_LOOKAHEAD_MIN=n   smallest trailing size whose update still hides a 256-block chain
    //   GPMI_CUMASK=0          never reserve whole compute units for the chain (round 2's free slots everywhere)
    //   GPMI_UPDATE256=0       the 128 x 128 update kernel everywhere (update256.hip off)
    // and five TEST HOOKS select the alternative code paths (NB-block substitution instead of the stored super-block inverses,
    // the one-product K^-1, the 256 x 128 update for small launches) at sizes a test can afford — tests/test_gpu_twolevel.py:
    //   GPMI_SUPER_INV=0  GPMI_WHITEN_INV=0  GPMI_WHITEN_SUPER=w  GPMI_GRAD_CHUNK=k  GPMI_UPDATE256_MIN=tiles
    //   (round 4) GPMI_UPDATE256_ATOMIC=1   (update256's C tile as no-return atomic adds: measured +0.3 %, off by default; tested)
    //             GPMI_TEST_COMM_DELAY_US / GPMI_TEST_COMM_DELAY_ON  (dev_hip.hip: latency injected in front of the in-process
    //             communicator's collectives — the overlap measurement of tests/test_gpu_dist.py)
    //   tools builds only: GPMI_UPDATE256_RECT=rows, GPMI_BLOCKED_U2A=q (blocked.cpp) besides the older ones below
    // Everything else (tile-shape overrides, the phase lock, refinement everywhere, C access width) is bring-up tooling and only
    // exists in a GPMI_TOOLS build (make TOOLS=1).
    if (const char* e = getenv("GPMI_SUPER")) {  // "min512,min1024,min2048" (remaining rows from which each width is used)
        long long a = 0, b = 0, d = 0;
        sscanf(e, "%lld,%lld,%lld", &a, &b, &d);
        c->super_min[0] = a; c->super_min[1] = b; c->super_min[2] = d;
    }
    //   (round 5) GPMI_CHAIN=0  the multi-launch diagonal-block chain instead of the persistent chain kernel (chain.hip);
    //             GPMI_CHAIN_WGS=g  workgroups of every chain launch (test hook: 1 = a serial walk of the task list)
    if (const char* e = getenv("GPMI_CHAIN")) c->chain_kernel = atoi(e) != 0;
    if (const char* e = getenv("GPMI_CHAIN_WGS")) c->chain_wgs = std::max(0, atoi(e));
    if (const char* e = getenv("GPMI_CHAIN_BESIDE_WGS")) c->chain_beside_wgs = std::max(8, atoi(e) / 8 * 8);
    //             GPMI_TAIL_FUSE=rows  the last `rows` rows (<= 2048) of a factorisation as ONE diagonal block (0 = off)
    //             GPMI_CUMASK_BELOW=rows  factorisations of fewer rows reserve whole compute units for the chain (default 32768)
    if (const char* e = getenv("GPMI_TAIL_FUSE")) c->tail_fuse = std::min<long long>(std::max<long long>(0, atoll(e)) / IB * IB, (long long)c->chain_nb_max * IB);
    if (const char* e = getenv("GPMI_CUMASK_BELOW")) c->whole_cus_below = atoll(e);
    if (const char* e = getenv("GPMI_UPDATE256")) c->update256 = atoi(e) != 0;
    if (const char* e = getenv("GPMI_UPDATE_FULL_GRID")) c->update_full_grid = atoi(e) != 0;
    if (const char* e = getenv("GPMI_KEND_HEAVY_FIRST")) c->kend_heavy_first = atoi(e) != 0;
    if (const char* e = getenv("GPMI_UPDATE256_KEND")) c->update256_kend = atoi(e);
    if (const char* e = getenv("GPMI_SUPER_W")) c->super_wide = atoll(e) / NB * NB;
    if (const char* e = getenv("GPMI_UPDATE256_MIN")) c->update256_min_tiles = std::max<long long>(1, atoll(e));
    if (const char* e = getenv("GPMI_UPDATE256_ATOMIC")) c->update256_atomic = atoi(e) != 0;

    if (const char* e = getenv("GPMI_GRAD_CHUNK")) c->grad_chunk = std::max<long long>(0, atoll(e) / NB * NB);
    if (const char* e = getenv("GPMI_SUPER_INV")) c->super_inverse = atoi(e) != 0;
    if (const char* e = getenv("GPMI_WHITEN_INV")) c->whiten_by_super_inverse = atoi(e) != 0;
    if (const char* e = getenv("GPMI_WHITEN_SUPER")) c->whiten_super = std::max<long long>(NB, atoll(e) / NB * NB);
#ifdef GPMI_TOOLS
    if (const char* e = getenv("GPMI_UPDATE256_RECT")) c->update256_rect_min_m = std::max<long long>(1, atoll(e));  // rows from which tall products take the 256 x 128 kernel
    if (const char* e = getenv("GPMI_PHASE_LOCK")) c->phase_lock_min_k = atoll(e);
    if (const char* e = getenv("GPMI_REFINE")) c->refine_default = atoi(e) != 0;
    if (const char* e = getenv("GPMI_GEMM_NI")) c->gemm_ni = atoi(e) == 2 ? 2 : atoi(e) == 4 ? 4 : 0;
    if (const char* e = getenv("GPMI_GEMM_WGS")) c->gemm_wgs_per_cu = atoi(e) == 1 ? 1 : 2;
#endif
    c->refine_solves = c->refine_default;
    if (getenv("GPMI_DEBUG"))
        fprintf(stderr, "[gpmi] device %d: %d CUs, look-ahead slots %d, CUs reserved for the chain %d\n", dev, c->num_cus, c->lookahead_slots,
                c->reserved_cus);
    *out = c;
    return GPMI_OK;
}

void gpmi_ctx_destroy(gpmi_ctx* c) {
    if (!c) return;
    if (c->group && c->group_rank == 0) group_destroy(c);  // the other members of a device group go with their primary
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& r : c->prof) {
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    for (auto e : c->ev_pool) hipEventDestroy(e);
    for (auto e : c->la_events) hipEventDestroy(e);
    for (hipStream_t st : {c->side_stream, c->upd_stream, c->side_masked})
        if (st) {
            hipStreamSynchronize(st);
            hipStreamDestroy(st);
        }
    for (void* p : {c->sup_lw, c->sup_lwt, c->sup_l256, c->sup_ut, c->sup_s, c->cov_scaled})
        if (p) hipFree(p);
    if (c->d_prog) hipFree(c->d_prog);
    if (c->h_prog) hipHostFree(c->h_prog);
    if (c->d_info) hipFree(c->d_info);
    if (c->d_scal) hipFree(c->d_scal);
    if (c->h_scal) hipHostFree(c->h_scal);
    if (c->d_queue) hipFree(c->d_queue);
    if (c->d_queue_side) hipFree(c->d_queue_side);
    if (c->chain_sync) hipFree(c->chain_sync);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

const char* gpmi_last_error(gpmi_ctx* c) { return c ? c->err.c_str() : "null context"; }

int gpmi_ctx_synchronize(gpmi_ctx* c) {
    if (!c) return GPMI_EARG;
    for (gpmi_ctx* m : group_members(c)) {
        GPMI_HIP(c, hipSetDevice(m->device));
        GPMI_HIP(c, hipDeviceSynchronize());
    }
    GPMI_HIP(c, hipSetDevice(c->device));
    return GPMI_OK;
}

int gpmi_gp_create(gpmi_ctx* c, int dtype, int d, int64_t n, const void* x, gpmi_gp** out) {
    if (!c) return earg(c, "gpmi_gp_create: bad argument");
    if (!out || !x || (dtype != 64 && dtype != 32) || d <= 0 || n <= 0) {
        c->err = "gpmi_gp_create: bad argument (dtype must be 64|32, 1 <= d <= 2^20, n >= 1)";
        return GPMI_EARG;
    }
    *out = nullptr;
    GPMI_HIP(c, hipSetDevice(c->device));
    gpmi_gp* gp = new gpmi_gp();
    gp->ctx = c;
    gp->dtype = dtype;
    gp->d = d;
    gp->n = n;
    gp->npad = (n + IB - 1) / IB * IB;
    const size_t es = dtype == 64 ? 8 : 4;
    // never a row stride that is a multiple of 4 KiB: a power-of-two stride parks every row of a tile on the same HBM
    // channels (measured on the FITC whitening GEMMs, ld = 4096 doubles: 23 instead of 45 TFLOP/s)
    gp->ld = ((gp->npad * (int64_t)es) % 4096 == 0) ? gp->npad + 64 : gp->npad;
    hipError_t e = hipMalloc(&gp->x, (size_t)(n * d) * es);
    if (e == hipSuccess) e = hipMalloc(&gp->A, (size_t)((gp->npad + 8) * gp->ld) * es);
    if (e == hipSuccess) e = hipMalloc(&gp->ymu, (size_t)gp->npad * es);
    if (e == hipSuccess) e = hipMalloc(&gp->alpha, (size_t)gp->npad * es);
    if (e == hipSuccess) e = hipMalloc(&gp->invdiag, (size_t)gp->npad * es);
    if (e == hipSuccess) e = hipMalloc(&gp->linv, (size_t)(gp->npad * IB) * es);
    if (e == hipSuccess) e = hipMalloc(&gp->linv256, (size_t)((gp->npad + NB - 1) / NB * NB * NB) * es);
    if (e == hipSuccess) e = hipMemcpy(gp->x, x, (size_t)(n * d) * es, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset((char*)gp->A + (size_t)(gp->npad * gp->ld) * es, 0, (size_t)(8 * gp->ld) * es);
    if (e != hipSuccess) {
        c->err = std::string("gpmi_gp_create: ") + hipGetErrorString(e);
        gpmi_gp_destroy(gp);
        return GPMI_EDEVICE;
    }
    *out = gp;
    return GPMI_OK;
}

void gpmi_gp_destroy(gpmi_gp* gp) {
    if (!gp) return;
    if (gp->ctx) {
        hipSetDevice(gp->ctx->device);
        hipStreamSynchronize(gp->ctx->stream);
    }
    if (gp->group) {
        group_gp_destroy(gp);
        delete gp;
        return;
    }
    if (gp->blocked) {
        blocked_destroy(gp->blocked);
        delete gp;
        return;
    }
    void* ptrs[] = {gp->x, gp->A, gp->ymu, gp->alpha, gp->invdiag, gp->linv, gp->linv256, gp->noise, gp->rows, gp->xp, gp->small, gp->supinv,
                    gp->g1, gp->g2, gp->gpart};
    for (void* p : ptrs)
        if (p) hipFree(p);
    delete gp;
}

int gpmi_fit(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, int64_t n_noise, const void* y_minus_mu,
             double* mll_out, void* alpha_out, int64_t* info_out) {
    if (!gp) return earg((gp ? gp->ctx : nullptr), "gpmi_fit: bad argument");
    gpmi_ctx* c = gp->ctx;
    if (info_out) *info_out = 0;
    if (!k || !log_noise || !y_minus_mu || (n_noise != 1 && n_noise != gp->n)) {
        c->err = "gpmi_fit: bad argument (logNoise must have length 1 or nobs)";
        return GPMI_EARG;
    }
    GPMI_HIP(c, hipSetDevice(c->device));
    if (gp->group) return group_fit(gp, k, log_noise, n_noise, y_minus_mu, mll_out, alpha_out, info_out);
    if (BlockedGP* b = blocked_of(gp)) {
        const int rc = b->fit(k, log_noise, n_noise, y_minus_mu, mll_out, alpha_out, info_out);
        if (rc != GPMI_OK) c->err = b->error();
        return rc;
    }
    return gp->dtype == 64 ? fit_t<double>(gp, k, log_noise, n_noise, y_minus_mu, mll_out, alpha_out, info_out)
                           : fit_t<float>(gp, k, log_noise, n_noise, y_minus_mu, mll_out, alpha_out, info_out);
}

int gpmi_predict(gpmi_gp* gp, const gpmi_kernel* k, int64_t p, const void* xpred, const void* mean_pred, int full_cov,
                 void* mu_out, void* var_out) {
    if (!gp) return earg((gp ? gp->ctx : nullptr), "gpmi_predict: bad argument");
    gpmi_ctx* c = gp->ctx;
    if (!k || p <= 0 || !xpred || !mean_pred || !mu_out || !var_out) {
        c->err = "gpmi_predict: bad argument";
        return GPMI_EARG;
    }
    if (gp->group) return group_predict(gp, k, p, xpred, mean_pred, full_cov, mu_out, var_out);
    if (BlockedGP* b = blocked_of(gp)) {
        GPMI_HIP(c, hipSetDevice(c->device));
        const int rc = b->predict(k, p, xpred, mean_pred, full_cov, mu_out, var_out);
        if (rc != GPMI_OK) c->err = b->error();
        return rc;
    }
    if (!gp->fitted) {
        c->err = "gpmi_predict: no valid factorisation (call gpmi_fit first)";
        return GPMI_EARG;
    }
    GPMI_HIP(c, hipSetDevice(c->device));
    return gp->dtype == 64 ? predict_t<double>(gp, k, p, xpred, mean_pred, full_cov, mu_out, var_out)
                           : predict_t<float>(gp, k, p, xpred, mean_pred, full_cov, mu_out, var_out);
}

int gpmi_grad(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, int64_t n_noise, double* dkern_out, int32_t n_kern,
              double* dnoise_out) {
    if (!gp) return earg((gp ? gp->ctx : nullptr), "gpmi_grad: bad argument");
    gpmi_ctx* c = gp->ctx;
    if (!k || !log_noise || !dkern_out) {
        c->err = "gpmi_grad: bad argument";
        return GPMI_EARG;
    }
    if (n_noise != 1 && dnoise_out) {
        c->err = "gpmi_grad: the noise gradient is defined for scalar logNoise only (GPE.jl:313)";
        return GPMI_EARG;
    }
    if (gp->group) return group_grad(gp, k, log_noise, n_noise, dkern_out, n_kern, dnoise_out);
    if (BlockedGP* b = blocked_of(gp)) {
        GPMI_HIP(c, hipSetDevice(c->device));
        const int rc = b->grad(k, log_noise, n_noise, dkern_out, n_kern, dnoise_out);
        if (rc != GPMI_OK) c->err = b->error();
        return rc;
    }
    if (!gp->fitted) {
        c->err = "gpmi_grad: no valid factorisation (call gpmi_fit first)";
        return GPMI_EARG;
    }
    {   // the caller's buffer must match the kernel's parameter count
        std::string err;
        std::vector<unsigned char> tmpb;
        if (digest_kernel(k, gp->d, &tmpb, &err) != GPMI_OK || reinterpret_cast<const DevProgram*>(tmpb.data())->n_hyp != n_kern) {
            c->err = err.empty() ? "gpmi_grad: dkern_out length differs from the kernel's number of parameters" : err;
            return GPMI_EARG;
        }
        if (n_kern > GPMI_GRAD_MAX_PARAMS)
            return earg(c, "gpmi_grad: more than GPMI_GRAD_MAX_PARAMS (5000) kernel hyper-parameters: the trace kernel's per-wave table does not fit the LDS");
    }
    GPMI_HIP(c, hipSetDevice(c->device));
    return gp->dtype == 64 ? grad_t<double>(gp, k, log_noise, dkern_out, dnoise_out)
                           : grad_t<float>(gp, k, log_noise, dkern_out, dnoise_out);
}

int gpmi_cov(gpmi_ctx* c, const gpmi_kernel* k, int dtype, int d, int64_t n1, const void* x1, int64_t n2, const void* x2,
             void* out) {
    if (!c) return earg(c, "gpmi_cov: bad argument");
    if (!k || !x1 || !out || n1 <= 0 || (x2 && n2 <= 0) || (dtype != 64 && dtype != 32)) {
        c->err = "gpmi_cov: bad argument";
        return GPMI_EARG;
    }
    GPMI_HIP(c, hipSetDevice(c->device));
    return dtype == 64 ? cov_t<double>(c, k, d, n1, x1, n2, x2, out) : cov_t<float>(c, k, d, n1, x1, n2, x2, out);
}

static int need_fit(gpmi_gp* gp, const char* who, bool blocked_ok = true) {
    if (!gp) return GPMI_EARG;
    if (BlockedGP* b = blocked_of(gp)) {
        if (!blocked_ok) {
            gp->ctx->err = std::string(who) + ": not provided on a blocked handle (gpmi_gp_create_blocked)";
            return GPMI_EARG;
        }
        if (!b->fitted()) {
            gp->ctx->err = std::string(who) + ": no valid factorisation (call gpmi_fit first)";
            return GPMI_EARG;
        }
        return GPMI_OK;
    }
    if (!gp->fitted) {
        gp->ctx->err = std::string(who) + ": no valid factorisation (call gpmi_fit first)";
        return GPMI_EARG;
    }
    return GPMI_OK;
}

int gpmi_solve(gpmi_gp* gp, int64_t nrhs, void* b) {
    int rc = need_fit(gp, "gpmi_solve", true);
    if (rc) return rc;
    if (nrhs <= 0 || !b) return earg((gp ? gp->ctx : nullptr), "gpmi_solve: bad argument");
    GPMI_HIP(gp->ctx, hipSetDevice(gp->ctx->device));
    if (gp->group) return group_solve(gp, nrhs, b, true);
    if (BlockedGP* bl = blocked_of(gp)) {
        const int rcb = bl->solve(nrhs, b, true);
        if (rcb != GPMI_OK) gp->ctx->err = bl->error();
        return rcb;
    }
    return gp->dtype == 64 ? solve_t<double>(gp, nrhs, b, true) : solve_t<float>(gp, nrhs, b, true);
}

// update_mll!(gp; noise = false, kern = false), src/GPE.jl:203-211 with update_cK! skipped: the factor is kept
int gpmi_update_alpha(gpmi_gp* gp, const void* y_minus_mu, double* mll_out, void* alpha_out) {
    int rc = need_fit(gp, "gpmi_update_alpha", true);
    if (rc) return rc;
    if (!y_minus_mu) return earg(gp->ctx, "gpmi_update_alpha: bad argument");
    gpmi_ctx* c = gp->ctx;
    GPMI_HIP(c, hipSetDevice(c->device));
    if (gp->group) return group_update_alpha(gp, y_minus_mu, mll_out, alpha_out);
    if (BlockedGP* bl = blocked_of(gp)) {
        const int rcb = bl->update_alpha(y_minus_mu, mll_out, alpha_out);
        if (rcb != GPMI_OK) c->err = bl->error();
        return rcb;
    }
    const size_t es = gp->dtype == 64 ? 8 : 4, bytes = (size_t)gp->n * es;
    std::vector<char> b(bytes);
    memcpy(b.data(), y_minus_mu, bytes);
    rc = gp->dtype == 64 ? solve_t<double>(gp, 1, b.data(), true) : solve_t<float>(gp, 1, b.data(), true);
    if (rc) return rc;
    // the device copies gpmi_predict / gpmi_grad read (entries n .. npad stay zero: identity padding)
    GPMI_HIP(c, hipMemcpy(gp->ymu, y_minus_mu, bytes, hipMemcpyHostToDevice));
    GPMI_HIP(c, hipMemcpy(gp->alpha, b.data(), bytes, hipMemcpyHostToDevice));
    double dot = 0.0;
    for (int64_t i = 0; i < gp->n; ++i)
        dot += gp->dtype == 64 ? ((const double*)y_minus_mu)[i] * ((const double*)b.data())[i]
                               : (double)((const float*)y_minus_mu)[i] * (double)((const float*)b.data())[i];
    gp->mll = -(dot + gp->logdet + 1.8378770664093453 * (double)gp->n) / 2.0;  // GPE.jl:210
    if (mll_out) *mll_out = gp->mll;
    if (alpha_out) memcpy(alpha_out, b.data(), bytes);
    return GPMI_OK;
}

int gpmi_whiten(gpmi_gp* gp, int64_t nrhs, void* b) {
    int rc = need_fit(gp, "gpmi_whiten", true);
    if (rc) return rc;
    if (nrhs <= 0 || !b) return earg((gp ? gp->ctx : nullptr), "gpmi_whiten: bad argument");
    GPMI_HIP(gp->ctx, hipSetDevice(gp->ctx->device));
    if (gp->group) return group_solve(gp, nrhs, b, false);
    if (BlockedGP* bl = blocked_of(gp)) {
        const int rcb = bl->solve(nrhs, b, false);
        if (rcb != GPMI_OK) gp->ctx->err = bl->error();
        return rcb;
    }
    return gp->dtype == 64 ? solve_t<double>(gp, nrhs, b, false) : solve_t<float>(gp, nrhs, b, false);
}

int gpmi_inv_diag(gpmi_gp* gp, void* out) {
    if (!gp || !out) return earg((gp ? gp->ctx : nullptr), "gpmi_inv_diag: bad argument");
    int rc = need_fit(gp, "gpmi_inv_diag", true);
    if (rc != GPMI_OK) return rc;
    hipSetDevice(gp->ctx->device);
    if (gp->group) return group_inv_diag(gp, out);
    if (BlockedGP* bl = blocked_of(gp)) {
        const int rcb = bl->inv_diag(out);
        if (rcb != GPMI_OK) gp->ctx->err = bl->error();
        return rcb;
    }
    return gp->dtype == 64 ? gpmi::inv_diag_t<double>(gp, out) : gpmi::inv_diag_t<float>(gp, out);
}

int gpmi_logdet(gpmi_gp* gp, double* out) {
    int rc = need_fit(gp, "gpmi_logdet", true);
    if (rc) return rc;
    if (!out) return earg((gp ? gp->ctx : nullptr), "gpmi_logdet: bad argument");
    if (BlockedGP* b = blocked_of(gp)) {
        *out = b->logdet();
        return GPMI_OK;
    }
    *out = gp->logdet;
    return GPMI_OK;
}

int gpmi_factor_to_host(gpmi_gp* gp, void* U_out) {
    int rc = need_fit(gp, "gpmi_factor_to_host", true);
    if (rc) return rc;
    if (!U_out) return earg((gp ? gp->ctx : nullptr), "gpmi_factor_to_host: bad argument");
    gpmi_ctx* c = gp->ctx;
    GPMI_HIP(c, hipSetDevice(c->device));
    if (gp->group) return group_factor_to_host(gp, U_out);
    if (BlockedGP* bl = blocked_of(gp)) {
        const int rcb = bl->factor_to_host(U_out);
        if (rcb != GPMI_OK) gp->ctx->err = bl->error();
        return rcb;
    }
    const size_t es = gp->dtype == 64 ? 8 : 4;
    const int64_t n = gp->n;
    // row-major lower L  ==  column-major upper U: a straight 2-D copy, then clear the other triangle
    GPMI_HIP(c, hipMemcpy2D(U_out, (size_t)n * es, gp->A, (size_t)gp->ld * es, (size_t)n * es, (size_t)n, hipMemcpyDeviceToHost));
    char* o = (char*)U_out;
    for (int64_t i = 0; i + 1 < n; ++i) memset(o + ((size_t)i * n + (size_t)i + 1) * es, 0, (size_t)(n - 1 - i) * es);
    return GPMI_OK;
}

int gpmi_factor_diag(gpmi_gp* gp, void* diag_out) {
    int rc = need_fit(gp, "gpmi_factor_diag", true);
    if (rc) return rc;
    gpmi_ctx* c = gp->ctx;
    if (!diag_out) {
        c->err = "gpmi_factor_diag: null output";
        return GPMI_EARG;
    }
    GPMI_HIP(c, hipSetDevice(c->device));
    if (gp->group) return group_factor_diag(gp, diag_out);
    if (BlockedGP* b = blocked_of(gp)) {
        rc = b->factor_diag(diag_out);
        if (rc != GPMI_OK) c->err = b->error();
        return rc;
    }
    const size_t es = gp->dtype == 64 ? 8 : 4;
    // one element per row, source pitch ld + 1 elements
    GPMI_HIP(c, hipMemcpy2D(diag_out, es, gp->A, (size_t)(gp->ld + 1) * es, es, (size_t)gp->n, hipMemcpyDeviceToHost));
    return GPMI_OK;
}

int gpmi_profile_enable(gpmi_ctx* c, int on) {
    if (!c) return earg(c, "gpmi_profile_enable: bad argument");
    int rc = drain_profile(c);
    c->prof_on = on != 0;
    c->prof_only = (on >= 2 && on < 64) ? on - 2 : -1;
    c->prof_skip_chain = on == 64;
    c->prof_phases_only = on == 65;
    for (int i = 0; i < GPMI_PROF_NCLASS; ++i) {
        c->prof_n[i] = 0;
        c->prof_ms[i] = 0;
        c->prof_work[i] = 0;
        c->prof_bytes[i] = 0;
    }
    return rc;
}

int gpmi_profile_get(gpmi_ctx* c, int cls, int64_t* launches, double* total_ms, double* work) {
    if (!c || cls < 0 || cls >= GPMI_PROF_NCLASS) return earg(c, "gpmi_profile_get: bad argument");
    int rc = drain_profile(c);
    if (rc) return rc;
    if (launches) *launches = c->prof_n[cls];
    if (total_ms) *total_ms = c->prof_ms[cls];
    if (work) *work = c->prof_work[cls];
    c->prof_n[cls] = 0;
    c->prof_ms[cls] = 0;
    c->prof_work[cls] = 0;
    return GPMI_OK;
}

int gpmi_profile_get_bytes(gpmi_ctx* c, int cls, double* bytes) {
    if (!c || cls < 0 || cls >= GPMI_PROF_NCLASS || !bytes) {
        if (c) c->err = "gpmi_profile_get_bytes: bad class or null output";
        return GPMI_EARG;
    }
    int rc = drain_profile(c);
    if (rc) return rc;
    *bytes = c->prof_bytes[cls];
    c->prof_bytes[cls] = 0;
    return GPMI_OK;
}

int gpmi_bench_gemm(gpmi_ctx* c, int dtype, int64_t M, int64_t N, int64_t K, int lower, int variant, int iters,
                    double* ms_out) {
    if (!c || !ms_out || (dtype != 64 && dtype != 32) || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || N > M) return earg(c, "gpmi_bench_gemm: bad argument");
    if (K % 64 != 0) return earg(c, "gpmi_bench_gemm: bad argument");
#ifndef GPMI_TOOLS
    if (variant != 0 && variant != 256) return earg(c, "gpmi_bench_gemm: ablation variants exist in a GPMI_TOOLS build only (make TOOLS=1)");
#endif
    GPMI_HIP(c, hipSetDevice(c->device));
    return dtype == 64 ? gemm_bench<double>(c, M, N, K, lower, variant, iters, ms_out)
                       : gemm_bench<float>(c, M, N, K, lower, variant, iters, ms_out);
}

int gpmi_mfma_peak(gpmi_ctx* c, int dtype, double* tflops_out) {
    if (!c || !tflops_out || (dtype != 64 && dtype != 32)) return earg(c, "gpmi_mfma_peak: bad argument");
    GPMI_HIP(c, hipSetDevice(c->device));
    return dtype == 64 ? mfma_peak<double>(c, tflops_out) : mfma_peak<float>(c, tflops_out);
}

}  // extern "C"

// one context on one device (a member of a device group is one of these too: group_create, dev_hip.hip)
int gpmi::create_member_context(int dev, gpmi_ctx** out) { return create_one_context(dev, out); }
