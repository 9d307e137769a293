/* cov_oracle.c — plain-C restatement of the reference's cov! loops.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gp_oracle.py header): used by tests/ as
 * a second, scalar-loop checker of the NumPy oracle and by bench.py's
 * cpu_baseline leg as the single-threaded kernel-matrix build the reference
 * performs (its cov! loop is single-threaded).  Never linked into libgpmi.so.
 * Parity unpinned by reference outputs (the Julia reference cannot run here).
 *
 * Follows, line for line in structure (not in text):
 *   cov!(cK,k,X,data)        src/kernels/kernels.jl:39-50   j outer, i<j inner, mirror
 *   cov!(cK,k,X1,X2,data)    src/kernels/kernels.jl:56-71   i outer, j inner
 *   cov_ij(Stationary)       src/kernels/stationary.jl:25-27
 *   _SqEuclidean_ij          src/kernels/distance.jl:50-56  s=0.0; s += (x-y)^2
 *   _WeightedSqEuclidean_ij  src/kernels/distance.jl:82-88  s += (x-y)^2*w[k]
 *   distij(Euclidean)        src/kernels/distance.jl:65-71  exact 0 on the X1===X2 diagonal
 *   leaf cov(k,r)            se_iso.jl:39 se_ard.jl:43 mat12_iso.jl:41 mat12_ard.jl:43
 *                            mat32_iso.jl:41-42 mat32_ard.jl:43-44 mat52_iso.jl:40-41
 *                            mat52_ard.jl:43-44 rq_iso.jl:44 rq_ard.jl:47
 *   Noise cov_ij             src/kernels/noise.jl:31-37  (isapprox, rtol = sqrt(eps))
 *   Const                    src/kernels/const.jl:36
 *   Sum / Prod cov_ij        sum_kernel.jl:15  prod_kernel.jl:14
 *   Masked cov_ij            masked_kernel.jl:44-49 (active rows only)
 *   update_cK! nugget        src/GPE.jl:173,181-183 ; make_posdef! src/GP.jl:104-108
 *
 * The kernel arrives as the gpmi_kernel postfix descriptor of include/gpmi.h
 * (the boundary's data format, not product code).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include "../include/gpmi.h"

#define ORACLE_STACK 32

static int isapprox(double a, double b) {
    /* Base.isapprox(a,b; rtol=sqrt(eps), atol=0) */
    const double rtol = 1.4901161193847656e-08;
    if (a == b) return 1;
    double m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
    return isfinite(a) && isfinite(b) && fabs(a - b) <= rtol * m;
}

/* one covariance entry: x1, x2 point at the d contiguous features of the two observations */
static double cov_entry(const gpmi_kernel* k, int d, const double* x1, const double* x2, int same_point) {
    double stack[ORACLE_STACK];
    int sp = 0;
    int pp = 0; /* param cursor */
    for (int op = 0; op < k->n_ops; ++op) {
        int code = k->ops[op];
        if (code == GPMI_K_SUM) {
            double r = stack[--sp], l = stack[--sp];
            stack[sp++] = l + r;
            continue;
        }
        if (code == GPMI_K_PROD) {
            double r = stack[--sp], l = stack[--sp];
            stack[sp++] = l * r;
            continue;
        }
        int d0 = k->dims_off[op], d1 = k->dims_off[op + 1];
        int nd = (d1 > d0) ? (d1 - d0) : d;
        const int32_t* dims = (d1 > d0) ? (k->dims + d0) : NULL;
        const double* par = k->params + pp;
        double v;
        if (code == GPMI_K_NOISE) {
            int same = 1;
            for (int z = 0; z < nd; ++z) {
                int kk = dims ? dims[z] : z;
                if (!isapprox(x1[kk], x2[kk])) { same = 0; break; }
            }
            v = same ? par[0] : 0.0;
            pp += 1;
        } else if (code == GPMI_K_CONST) {
            v = par[0];
            pp += 1;
        } else {
            int ard = (code == GPMI_K_SE_ARD || code == GPMI_K_MAT12_ARD || code == GPMI_K_MAT32_ARD ||
                       code == GPMI_K_MAT52_ARD || code == GPMI_K_RQ_ARD);
            double s = 0.0;
            for (int z = 0; z < nd; ++z) {
                int kk = dims ? dims[z] : z;
                double df = x1[kk] - x2[kk];
                s += ard ? df * df * par[z] : df * df;
            }
            const double* q = par + (ard ? nd : 1); /* -> s2 (, alpha) */
            double s2 = q[0];
            int euclid = !(code == GPMI_K_SE_ISO || code == GPMI_K_SE_ARD || code == GPMI_K_RQ_ISO || code == GPMI_K_RQ_ARD);
            double r = s;
            if (euclid) r = same_point ? 0.0 : sqrt(s);
            switch (code) {
                case GPMI_K_SE_ISO: v = s2 * exp(-0.5 * r / par[0]); break;
                case GPMI_K_SE_ARD: v = s2 * exp(-r / 2.0); break;
                case GPMI_K_MAT12_ISO: v = s2 * exp(-r / par[0]); break;
                case GPMI_K_MAT12_ARD: v = s2 * exp(-r); break;
                case GPMI_K_MAT32_ISO: { double t = sqrt(3.0) * r / par[0]; v = s2 * (1.0 + t) * exp(-t); } break;
                case GPMI_K_MAT32_ARD: { double t = sqrt(3.0) * r; v = s2 * (1.0 + t) * exp(-t); } break;
                case GPMI_K_MAT52_ISO: { double t = sqrt(5.0) * r / par[0]; v = s2 * (1.0 + t + t * t / 3.0) * exp(-t); } break;
                case GPMI_K_MAT52_ARD: { double t = sqrt(5.0) * r; v = s2 * (1.0 + t + t * t / 3.0) * exp(-t); } break;
                case GPMI_K_RQ_ISO: v = s2 * pow(1.0 + r / (2.0 * q[1] * par[0]), -q[1]); break;
                case GPMI_K_RQ_ARD: v = s2 * pow(1.0 + 0.5 * r / q[1], -q[1]); break;
                default: v = NAN;
            }
            pp += (ard ? nd : 1) + 1 + ((code == GPMI_K_RQ_ISO || code == GPMI_K_RQ_ARD) ? 1 : 0);
        }
        stack[sp++] = v;
    }
    return stack[0];
}

/* Symmetric form, kernels.jl:39-50.  x: d x n col-major; out: n x n col-major. */
void oracle_cov_sym(const gpmi_kernel* k, int d, int64_t n, const double* x, double* out) {
    for (int64_t j = 0; j < n; ++j) {
        out[j + j * n] = cov_entry(k, d, x + j * d, x + j * d, 1);
        for (int64_t i = 0; i < j; ++i) {
            double v = cov_entry(k, d, x + i * d, x + j * d, 0);
            out[i + j * n] = v;
            out[j + i * n] = v;
        }
    }
}

/* Rectangular form, kernels.jl:56-71.  out: n1 x n2 col-major. */
void oracle_cov_rect(const gpmi_kernel* k, int d, int64_t n1, const double* x1, int64_t n2, const double* x2, double* out) {
    for (int64_t i = 0; i < n1; ++i)
        for (int64_t j = 0; j < n2; ++j)
            out[i + j * n1] = cov_entry(k, d, x1 + i * d, x2 + j * d, 0);
}

/* update_cK! up to (not including) the Cholesky: cov! + nugget (GPE.jl:169-186, GP.jl:104-108). */
void oracle_assemble(const gpmi_kernel* k, int d, int64_t n, const double* x, const double* log_noise, int64_t n_noise, double* out) {
    oracle_cov_sym(k, d, n, x, out);
    for (int64_t i = 0; i < n; ++i)
        out[i + i * n] += exp(2.0 * log_noise[n_noise == 1 ? 0 : i]);
}
