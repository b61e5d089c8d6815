/*
 * gpc_oracle.c -- CPU ORACLE (test infrastructure only; see gpc_oracle.h header).
 *
 * Plain-C restatement of the reference arithmetic, following
 *   src/rbf_kernel.cpp:15-18,61-71        src/gaussian_noise.cpp:9-18
 *   src/gaussian_noise_3d.cpp:11-20       src/probit_noise.cpp:11-31
 *   src/gaussian_process.cpp:15-64        src/sparse_gp.hpp:27-33,43-86,89-249,252-295,299-351,523-530,573-582
 *   src/sparse_gp_field.hpp:14-17,29-57,59-215,219-263,268-320
 *   src/gp_compressor.cpp:251-265,317-340,367-372
 * (paths relative to /root/reference).  Build with -ffp-contract=off so that
 * no FMA contraction changes the rounding sequence written here.
 */
#include "gpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ a1 / a2 */

/* src/rbf_kernel.cpp:15-18:  p(0)*exp(-0.5f / p(1) * (xi - xj).squaredNorm()) */
double orc_rbf_kernel(double p0, double p1, double xi0, double xi1, double xj0, double xj1)
{
    double d0 = xi0 - xj0, d1 = xi1 - xj1;
    double sq = d0 * d0 + d1 * d1;
    return p0 * exp((double)(-0.5f) / p1 * sq);
}

/* src/rbf_kernel.cpp:61-71: row i of K is p(0)*exp(-0.5f/p(1)*||X_j - BV_i||^2) */
void orc_rbf_construct_covariance_fast(double p0, double p1, int N, const double* x0, const double* x1,
                                       int b, const double* BV, double* K)
{
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < N; ++j) {
            double d0 = x0[j] - BV[2 * i], d1 = x1[j] - BV[2 * i + 1];
            double t = d0 * d0 + d1 * d1;
            K[i + (size_t)j * b] = p0 * exp((double)(-0.5f) / p1 * t);
        }
    }
}

/* ------------------------------------------------------------------ a3 / a4 / a5 */

/* src/gaussian_noise.cpp:9-12 */
double orc_gaussian_dx_ln(double s20, double y, double x, double sigma_x)
{
    return (y - x) / (s20 + sigma_x);
}

/* src/gaussian_noise.cpp:15-18 */
double orc_gaussian_dx2_ln(double s20, double y, double x, double sigma_x)
{
    (void)y; (void)x;
    return (double)(-1.0f) / (s20 + sigma_x);
}

/* src/gaussian_noise_3d.cpp:11-14 */
void orc_gaussian3d_dx_ln(double s20, int ny, const double* y, const double* x, double sigma_x, double* q)
{
    for (int c = 0; c < ny; ++c) q[c] = (y[c] - x[c]) / (s20 + sigma_x);
}

/* src/gaussian_noise_3d.cpp:17-20 */
double orc_gaussian3d_dx2_ln(double s20, double sigma_x)
{
    return (double)(-1.0f) / (s20 + sigma_x);
}

/* `2.0f*sqrt(2.0f)` in src/probit_noise.cpp:15,26 is a C++ expression: <math.h> of libstdc++ exposes the
 * sqrt(float) overload, so the product is evaluated in float and only then promoted (checked against the
 * compiled reference object, tests/golden/noise_ref.json). */
#define PROBIT_TWO_SQRT2 ((double)(2.0f * sqrtf(2.0f)))

/* src/probit_noise.cpp:11-18 */
double orc_probit_dx_ln(double s20, double y, double x, double sigma_x)
{
    double sigma = sqrt(s20 + sigma_x);
    double z = y * x / sigma;
    double ef = erf(z) / PROBIT_TWO_SQRT2;
    double efprim = exp(-z * z / 2) / sqrt((double)2.0f * M_PI);
    return y / sigma * efprim / ef;
}

/* src/probit_noise.cpp:21-31 */
double orc_probit_dx2_ln(double s20, double y, double x, double sigma_x)
{
    double sigma2 = s20 + sigma_x;
    double sigma = sqrt(sigma2);
    double z = y * x / sigma;
    double ef = erf(z) / PROBIT_TWO_SQRT2;
    double efprim = exp(-z * z / (double)2.0f) / sqrt((double)2.0f * M_PI);
    double efprimprim = -z * efprim;
    double first = efprim / ef;
    return (efprimprim / ef - first * first) / sigma2;
}

/* The fixed variant (noise_model 2): probit_noise with Phi(z) = (1 + erf(z / sqrt 2)) / 2 -- a CDF -- in place of
 * erf(z)/(2.0f*sqrt(2.0f)) (src/probit_noise.cpp:15,26); every other operation as upstream. */
double orc_probit_std_dx_ln(double s20, double y, double x, double sigma_x)
{
    double sigma = sqrt(s20 + sigma_x);
    double z = y * x / sigma;
    double ef = 0.5 * erfc(-z * 0.70710678118654752440);
    double efprim = exp(-z * z / 2) / sqrt((double)2.0f * M_PI);
    return y / sigma * efprim / ef;
}

double orc_probit_std_dx2_ln(double s20, double y, double x, double sigma_x)
{
    double sigma2 = s20 + sigma_x;
    double sigma = sqrt(sigma2);
    double z = y * x / sigma;
    double ef = 0.5 * erfc(-z * 0.70710678118654752440);
    double efprim = exp(-z * z / (double)2.0f) / sqrt((double)2.0f * M_PI);
    double efprimprim = -z * efprim;
    double first = efprim / ef;
    return (efprimprim / ef - first * first) / sigma2;
}

/* ------------------------------------------------------------------ a6 - a8: dense GP */

void orc_dense_default_params(orc_dense_params* p)
{
    /* src/gaussian_process.h:21 with the squaring of src/gaussian_process.cpp:8-9 */
    p->sigmaf_sq = 0.05 * 0.05;
    p->l_sq = 3.0 * 3.0;
    p->sigman_sq = 0.04 * 0.04;
    p->ref_double_noise = 1;
}

/* src/gaussian_process.cpp:47-50 */
static double squared_exp_distance(const orc_dense_params* p, double xi0, double xi1, double xj0, double xj1)
{
    double d0 = xi0 - xj0, d1 = xi1 - xj1;
    double sq = d0 * d0 + d1 * d1;
    return p->sigmaf_sq * exp((double)(-0.5f) / p->l_sq * sq);
}

int orc_dense_fit(const orc_dense_params* p, int n, const double* x0, const double* x1,
                  const double* y, int ny, double* L, double* alpha)
{
    if (n <= 0) return 0;
    size_t ld = (size_t)n;
    /* covariance_matrix(K, X, X, true): src/gaussian_process.cpp:52-64 -- all n*n entries, no symmetry used */
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            double c = squared_exp_distance(p, x0[i], x1[i], x0[j], x1[j]);
            if (i == j) c += p->sigman_sq;
            L[i + j * ld] = c;
        }
    }
    /* C = K; C.diagonal() += sigman_sq  (src/gaussian_process.cpp:20-21; the second addition, F5) */
    if (p->ref_double_noise)
        for (int i = 0; i < n; ++i) L[i + i * ld] += p->sigman_sq;
    /* chol = C.llt()  (src/gaussian_process.cpp:22): lower Cholesky, left-looking column form */
    int info = 0;
    for (int j = 0; j < n; ++j) {
        double d = L[j + j * ld];
        for (int k = 0; k < j; ++k) d -= L[j + k * ld] * L[j + k * ld];
        if (!(d > 0.0)) { info = 1 + j; break; }
        d = sqrt(d);
        L[j + j * ld] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = L[i + j * ld];
            for (int k = 0; k < j; ++k) s -= L[i + k * ld] * L[j + k * ld];
            L[i + j * ld] = s / d;
        }
    }
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) L[i + j * ld] = 0.0;
    if (info) {
        for (int c = 0; c < ny; ++c)
            for (int i = 0; i < n; ++i) alpha[i + (size_t)c * n] = NAN;
        return info;
    }
    /* alpha = chol.solve(y)  (src/gaussian_process.cpp:24): L z = y, L^T alpha = z */
    for (int c = 0; c < ny; ++c) {
        double* a = alpha + (size_t)c * n;
        const double* yc = y + (size_t)c * n;
        for (int i = 0; i < n; ++i) {
            double s = yc[i];
            for (int k = 0; k < i; ++k) s -= L[i + k * ld] * a[k];
            a[i] = s / L[i + i * ld];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = a[i];
            for (int k = i + 1; k < n; ++k) s -= L[k + i * ld] * a[k];
            a[i] = s / L[i + i * ld];
        }
    }
    return 0;
}

void orc_dense_predict(const orc_dense_params* p, int n, const double* x0, const double* x1,
                       const double* L, const double* alpha, int ny,
                       int m, const double* xs0, const double* xs1, double* f_star, double* v_star)
{
    size_t ld = (size_t)n;
    double* kcol = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < m; ++j) {
        /* K_star.col(j): covariance_matrix(K_star, X, X_star), src/gaussian_process.cpp:31,52-64 */
        for (int i = 0; i < n; ++i) kcol[i] = squared_exp_distance(p, x0[i], x1[i], xs0[j], xs1[j]);
        /* f_star = K_star^T alpha  (src/gaussian_process.cpp:32) */
        for (int c = 0; c < ny; ++c) {
            const double* a = alpha + (size_t)c * n;
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += kcol[i] * a[i];
            f_star[j + (size_t)c * m] = s;
        }
        if (v_star) {
            /* v = chol.matrixL().solve(K_star); V_star(m) = k** - v^T v  (src/gaussian_process.cpp:35-43) */
            double vv = 0.0;
            for (int i = 0; i < n; ++i) {
                double s = kcol[i];
                for (int k = 0; k < i; ++k) s -= L[i + k * ld] * kcol[k];
                kcol[i] = s / L[i + i * ld];
                vv += kcol[i] * kcol[i];
            }
            double kss = squared_exp_distance(p, xs0[j], xs1[j], xs0[j], xs1[j]);
            v_star[j] = kss - vv;
        }
    }
    free(kcol);
}

int orc_dense_fit_predict_batch(const orc_dense_params* p, int P, const int32_t* off,
                                const double* x0, const double* x1, const double* y, int ny,
                                int m, const double* xs0, const double* xs1,
                                double* f_star, double* v_star, int32_t* status, double* alpha_out)
{
    int nmax = 0;
    for (int i = 0; i < P; ++i) {
        int n = off[i + 1] - off[i];
        if (n > nmax) nmax = n;
    }
    size_t N = (size_t)off[P];
    double* L = (double*)malloc(sizeof(double) * (size_t)nmax * (size_t)nmax + 8);
    double* alpha = (double*)malloc(sizeof(double) * (size_t)nmax * (size_t)ny + 8);
    double* yb = (double*)malloc(sizeof(double) * (size_t)nmax * (size_t)ny + 8);
    if (!L || !alpha || !yb) { free(L); free(alpha); free(yb); return -12; }
    for (int i = 0; i < P; ++i) {
        int o = off[i], n = off[i + 1] - off[i];
        /* y is ny planes of N (plane c at y + c*N); gather the patch rows of each plane */
        for (int c = 0; c < ny; ++c) memcpy(yb + (size_t)c * n, y + (size_t)c * N + o, sizeof(double) * (size_t)n);
        int info = orc_dense_fit(p, n, x0 + o, x1 + o, yb, ny, L, alpha);
        if (status) status[i] = info ? 1 : 0;
        if (alpha_out)
            for (int c = 0; c < ny; ++c) memcpy(alpha_out + (size_t)c * N + o, alpha + (size_t)c * n, sizeof(double) * (size_t)n);
        double* fs = f_star + (size_t)i * ny * m;
        orc_dense_predict(p, n, x0 + o, x1 + o, L, alpha, ny, m, xs0, xs1, fs, v_star ? v_star + (size_t)i * m : NULL);
    }
    free(L); free(alpha); free(yb);
    return 0;
}

/* ------------------------------------------------------------------ C5: dense GP with the probit functor, IRLS loop
 *
 * BASELINE config 5 ("probit_noise occupancy-GP variant, IRLS inner loop").  The reference never instantiates
 * probit_noise and holds no such loop: what it fixes is the Noise contract -- q = dx_ln(y, x, sigma_x) =
 * d/dx ln P(y|x), r = dx2_ln(y, x, sigma_x) = d2/dx2 ln P(y|x) (src/probit_noise.cpp:11-31) -- and the kernel
 * (src/rbf_kernel.cpp:15-18).  The loop below is the textbook one those two plug into: Newton's method for the mode of
 * p(f | y) ~ prod_i P(y_i | f_i) N(f | 0, K), Rasmussen & Williams (2006) eq. 3.18, in its iteratively-reweighted
 * least-squares form.  With W = -diag(r), g = q:
 *        f_new = (K^-1 + W)^-1 (W f + g) = K (K + W^-1)^-1 (f + W^-1 g)
 * i.e. every Newton step is ONE dense GP regression fit (gaussian_process::add_measurements, src/gaussian_process.cpp:15-26)
 * with per-point noise d_i = 1 / W_i on the diagonal and working targets t_i = f_i + g_i d_i:
 *        a = (K + diag d)^-1 t,   f_new = K a = t - d o a.
 * The functor is called with sigma_x = 0 (Laplace: the latent value itself, no predictive variance), so sigma^2 = s20.
 * Start f_i = y_i * f_init (f_init = 0 is R&W's start; the reference's "Phi" is singular at 0 and needs f_init > 0).
 * Stop after max_iter solves or when max_i |f_new_i - f_i| <= tol.  A weight that is not finite and positive ends the
 * patch with status 2 (the same NaN the reference's recursion would print, src/sparse_gp.hpp:245).
 * Predictive latent mean: f* = K*^T a  (R&W eq. 3.21: k*^T grad log p(y | f^) = k*^T a at the mode).
 * Returns 0 ok, 1 + j non-SPD pivot j, -2 NaN weight, -5 the step cap ended the loop (outputs are the last iterate). */
int orc_dense_irls_fit(const orc_dense_params* p, int noise_model, int n, const double* x0, const double* x1, const double* y,
                       int max_iter, double tol, double f_init, double* alpha, double* fhat, int32_t* iters)
{
    *iters = 0;
    if (n <= 0) return 0;
    const size_t ld = (size_t)n;
    double* K = (double*)malloc(sizeof(double) * ld * ld);      /* row-major lower triangle of the kernel matrix */
    double* L = (double*)malloc(sizeof(double) * ld * ld);      /* row-major Cholesky factor */
    double* d = (double*)malloc(sizeof(double) * ld * 2);
    double* t = d + n;
    if (!K || !L || !d) { free(K); free(L); free(d); return -12; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) K[i * ld + j] = squared_exp_distance(p, x0[i], x1[i], x0[j], x1[j]);
    for (int i = 0; i < n; ++i) fhat[i] = y[i] * f_init;
    int rc = 0;
    for (int it = 0; it < max_iter; ++it) {
        for (int i = 0; i < n; ++i) {
            double q, r;
            if (noise_model == 2) {
                q = orc_probit_std_dx_ln(p->sigman_sq, y[i], fhat[i], 0.0);
                r = orc_probit_std_dx2_ln(p->sigman_sq, y[i], fhat[i], 0.0);
            } else {
                q = orc_probit_dx_ln(p->sigman_sq, y[i], fhat[i], 0.0);
                r = orc_probit_dx2_ln(p->sigman_sq, y[i], fhat[i], 0.0);
            }
            const double W = -r;
            if (!(W > 0.0) || !(W < INFINITY) || q != q) { rc = -2; break; }
            d[i] = 1.0 / W;
            t[i] = fhat[i] + q * d[i];
        }
        if (rc) break;
        /* (K + diag d) = L L^T, row by row (Cholesky-Crout; dot products over contiguous rows) */
        for (int i = 0; i < n && !rc; ++i) {
            for (int j = 0; j <= i; ++j) {
                double s = K[i * ld + j];
                if (i == j) s += d[i];
                const double* li = L + i * ld;
                const double* lj = L + j * ld;
                for (int k = 0; k < j; ++k) s -= li[k] * lj[k];
                if (i == j) {
                    if (!(s > 0.0)) { rc = 1 + i; break; }
                    L[i * ld + i] = sqrt(s);
                } else {
                    L[i * ld + j] = s / L[j * ld + j];
                }
            }
        }
        if (rc) break;
        for (int i = 0; i < n; ++i) {                              /* L z = t */
            double s = t[i];
            for (int k = 0; k < i; ++k) s -= L[i * ld + k] * alpha[k];
            alpha[i] = s / L[i * ld + i];
        }
        for (int i = n - 1; i >= 0; --i) {                         /* L^T a = z */
            double s = alpha[i];
            for (int k = i + 1; k < n; ++k) s -= L[k * ld + i] * alpha[k];
            alpha[i] = s / L[i * ld + i];
        }
        double delta = 0.0;
        for (int i = 0; i < n; ++i) {
            const double fn = t[i] - d[i] * alpha[i];
            const double df = fabs(fn - fhat[i]);
            if (df > delta || df != df) delta = df;
            fhat[i] = fn;
        }
        *iters = it + 1;
        if (delta != delta) { rc = -2; break; }
        if (delta <= tol) break;
        if (it + 1 == max_iter) rc = -5;
    }
    if (rc && rc != -5)
        for (int i = 0; i < n; ++i) alpha[i] = fhat[i] = NAN;
    free(K); free(L); free(d);
    return rc;
}

/* batch driver, same shape as gpc_dense_irls_fit_predict (include/gpc.h); status: 0 ok, 1 non-SPD, 2 NaN, 5 not converged */
int orc_dense_irls_fit_predict_batch(const orc_dense_params* p, int noise_model, int max_iter, double tol, double f_init,
                                     int P, const int32_t* off, const double* x0, const double* x1, const double* y,
                                     int m, const double* xs0, const double* xs1, double* f_star, double* alpha_out,
                                     double* fhat_out, int32_t* iters, int32_t* status)
{
    int nmax = 0;
    for (int i = 0; i < P; ++i)
        if (off[i + 1] - off[i] > nmax) nmax = off[i + 1] - off[i];
    double* a = (double*)malloc(sizeof(double) * (size_t)(2 * nmax + 2));
    if (!a) return -12;
    double* fh = a + nmax + 1;
    for (int i = 0; i < P; ++i) {
        const int o = off[i], n = off[i + 1] - o;
        int32_t it = 0;
        const int rc = orc_dense_irls_fit(p, noise_model, n, x0 + o, x1 + o, y + o, max_iter, tol, f_init, a, fh, &it);
        if (rc == -12) { free(a); return -12; }
        if (status) status[i] = rc == 0 ? 0 : (rc > 0 ? 1 : (rc == -5 ? 5 : 2));
        if (iters) iters[i] = it;
        if (alpha_out) memcpy(alpha_out + o, a, sizeof(double) * (size_t)n);
        if (fhat_out) memcpy(fhat_out + o, fh, sizeof(double) * (size_t)n);
        if (m > 0) orc_dense_predict(p, n, x0 + o, x1 + o, NULL, a, 1, m, xs0, xs1, f_star + (size_t)i * m, NULL);
    }
    free(a);
    return 0;
}

/* ------------------------------------------------------------------ a9 - a13: sparse online GP */

struct orc_sparse {
    orc_sparse_params p;
    int ld;            /* leading dimension / max basis vectors */
    int b;             /* current_size */
    int total_count;
    double* alpha;     /* ny planes of ld */
    double* C;         /* ld x ld col-major */
    double* Q;
    double* BV;        /* 2 x ld interleaved */
    double* k;         /* scratch, ld each */
    double* e_hat;
    double* s;
    double* t;
    int32_t n_full, n_sparse, n_deleted;
    uint8_t last_dec;  /* decision byte of the last add (layout: gpc_oracle_hp.c) */
};

void orc_sparse_default_params(orc_sparse_params* p, int ny)
{
    p->p0 = (double)100e-0f;              /* src/rbf_kernel.h:24 */
    p->p1 = 1 * 1;
    p->capacity = 100;                    /* src/sparse_gp.h:48, src/sparse_gp_field.h:43 */
    p->ny = ny;
    p->noise_model = 0;
    p->field_delete_bug = 1;
    if (ny == 1) {
        p->s20 = (double)1e-1f;           /* src/sparse_gp.h:48 */
        p->eps_tol = (double)1e-6f;       /* src/sparse_gp.hpp:30 */
    } else {
        p->s20 = (double)1e2f;            /* src/sparse_gp_field.h:43 */
        p->eps_tol = (double)1e-4f;       /* src/sparse_gp_field.hpp:16 */
    }
}

orc_sparse* orc_sparse_create(const orc_sparse_params* p, int max_bv)
{
    orc_sparse* g = (orc_sparse*)calloc(1, sizeof(orc_sparse));
    if (!g) return NULL;
    g->p = *p;
    g->ld = max_bv < 2 ? 2 : max_bv;
    size_t ld = (size_t)g->ld;
    g->alpha = (double*)calloc(ld * (size_t)p->ny, sizeof(double));
    g->C = (double*)calloc(ld * ld, sizeof(double));
    g->Q = (double*)calloc(ld * ld, sizeof(double));
    g->BV = (double*)calloc(2 * ld, sizeof(double));
    g->k = (double*)calloc(ld, sizeof(double));
    g->e_hat = (double*)calloc(ld, sizeof(double));
    g->s = (double*)calloc(ld, sizeof(double));
    g->t = (double*)calloc(ld, sizeof(double));
    return g;
}

void orc_sparse_destroy(orc_sparse* g)
{
    if (!g) return;
    free(g->alpha); free(g->C); free(g->Q); free(g->BV);
    free(g->k); free(g->e_hat); free(g->s); free(g->t);
    free(g);
}

/* src/sparse_gp.hpp:573-582 */
void orc_sparse_reset(orc_sparse* g)
{
    g->total_count = 0;
    g->b = 0;
    g->n_full = g->n_sparse = g->n_deleted = 0;
}

int orc_sparse_size(const orc_sparse* g) { return g->b; }
int orc_sparse_total_count(const orc_sparse* g) { return g->total_count; }

#define Cm(i, j) g->C[(size_t)(i) + (size_t)(j) * ld]
#define Qm(i, j) g->Q[(size_t)(i) + (size_t)(j) * ld]
#define Al(c, i) g->alpha[(size_t)(c) * ld + (size_t)(i)]

/* src/sparse_gp.hpp:252-295 (ny==1) and src/sparse_gp_field.hpp:219-263 (ny>1) */
void orc_sparse_delete_bv(orc_sparse* g, int loc)
{
    const size_t ld = (size_t)g->ld;
    const int b = g->b, last = b - 1, ny = g->p.ny;
    double alphastar[8];
    double* Cstar = g->s;   /* scratch vectors (length b-1 after the swap/shrink) */
    double* Qstar = g->t;

    /* First swap loc to the last spot (:256-258) */
    for (int c = 0; c < ny; ++c) {
        alphastar[c] = Al(c, loc);
        Al(c, loc) = Al(c, last);
    }
    /* Now C (:261-270) */
    double cstar = Cm(loc, loc);
    for (int i = 0; i < b; ++i) Cstar[i] = Cm(i, loc);
    Cstar[loc] = Cstar[last];
    {
        double* Crep = g->k;
        for (int i = 0; i < b; ++i) Crep[i] = Cm(i, last);
        Crep[loc] = Crep[last];
        for (int i = 0; i < b; ++i) Cm(loc, i) = Crep[i];
        for (int i = 0; i < b; ++i) Cm(i, loc) = Crep[i];
    }
    /* and Q (:273-281) */
    double qstar = Qm(loc, loc);
    for (int i = 0; i < b; ++i) Qstar[i] = Qm(i, loc);
    Qstar[loc] = Qstar[last];
    {
        double* Qrep = g->k;
        for (int i = 0; i < b; ++i) Qrep[i] = Qm(i, last);
        Qrep[loc] = Qrep[last];
        for (int i = 0; i < b; ++i) Qm(loc, i) = Qrep[i];
        for (int i = 0; i < b; ++i) Qm(i, loc) = Qrep[i];
    }
    const int nb = b - 1;
    /* the actual removal, Appendix G section g (:284-288) */
    if (ny == 1) {
        double f = alphastar[0] / (qstar + cstar);
        for (int i = 0; i < nb; ++i) Al(0, i) -= f * (Qstar[i] + Cstar[i]);
    } else {
        /* src/sparse_gp_field.hpp:250-253: qc = (qstar + cstar)*(Qstar + Cstar)  -- multiplies (F8) */
        for (int i = 0; i < nb; ++i) {
            double qc = g->p.field_delete_bug ? (qstar + cstar) * (Qstar[i] + Cstar[i])
                                              : (Qstar[i] + Cstar[i]) / (qstar + cstar);
            for (int c = 0; c < ny; ++c) Al(c, i) -= alphastar[c] * qc;
        }
    }
    for (int j = 0; j < nb; ++j) {
        for (int i = 0; i < nb; ++i) {
            double qq = (Qstar[i] * Qstar[j]) / qstar;
            double qc = ((Qstar[i] + Cstar[i]) * (Qstar[j] + Cstar[j])) / (qstar + cstar);
            Cm(i, j) += qq - qc;
            Qm(i, j) -= qq;
        }
    }
    /* And the BV (:291-292) */
    g->BV[2 * loc] = g->BV[2 * last];
    g->BV[2 * loc + 1] = g->BV[2 * last + 1];
    g->b = nb;
    g->n_deleted++;
}

/* src/sparse_gp.hpp:89-249 and src/sparse_gp_field.hpp:59-215 */
void orc_sparse_add(orc_sparse* g, double x0, double x1, const double* y)
{
    const size_t ld = (size_t)g->ld;
    const orc_sparse_params* P = &g->p;
    const int ny = P->ny;
    g->total_count++;
    double kstar = orc_rbf_kernel(P->p0, P->p1, x0, x1, x0, x1);

    if (g->b == 0) {
        /* First point (:100-114): Equations 2.46 with q, r, s collapsed */
        for (int c = 0; c < ny; ++c) Al(c, 0) = y[c] / (kstar + P->s20);
        Cm(0, 0) = (double)(-1.0f) / (kstar + P->s20);
        Qm(0, 0) = (double)(1.0f) / kstar;
        g->b = 1;
        g->BV[0] = x0;
        g->BV[1] = x1;
        g->n_full++;
        g->last_dec = 0x81;
        return;   /* the reference only checks isnan(C(0,0)) after this (:245) */
    }

    int b = g->b;
    double* k = g->k;
    double* e_hat = g->e_hat;
    /* construct_covariance(k, X, BV) (:119, :523-530) */
    for (int i = 0; i < b; ++i) k[i] = orc_rbf_kernel(P->p0, P->p1, x0, x1, g->BV[2 * i], g->BV[2 * i + 1]);

    /* m = alpha^T k ; s2 = kstar + k^T C k (:121-122) */
    double m[8];
    for (int c = 0; c < ny; ++c) {
        double s = 0.0;
        for (int i = 0; i < b; ++i) s += Al(c, i) * k[i];
        m[c] = s;
    }
    double kCk = 0.0;
    for (int j = 0; j < b; ++j) {
        double tj = 0.0;
        for (int i = 0; i < b; ++i) tj += k[i] * Cm(i, j);
        kCk += tj * k[j];
    }
    double s2 = kstar + kCk;

    /* r = noise.dx2_ln(y, m, s2); q = noise.dx_ln(y, m, s2) (:134-137) */
    double r, q[8];
    if (ny == 1 && P->noise_model == 1) {
        r = orc_probit_dx2_ln(P->s20, y[0], m[0], s2);
        q[0] = orc_probit_dx_ln(P->s20, y[0], m[0], s2);
    } else if (ny == 1 && P->noise_model == 2) {
        r = orc_probit_std_dx2_ln(P->s20, y[0], m[0], s2);
        q[0] = orc_probit_std_dx_ln(P->s20, y[0], m[0], s2);
    } else if (ny == 1) {
        r = orc_gaussian_dx2_ln(P->s20, y[0], m[0], s2);
        q[0] = orc_gaussian_dx_ln(P->s20, y[0], m[0], s2);
    } else {
        r = orc_gaussian3d_dx2_ln(P->s20, s2);
        orc_gaussian3d_dx_ln(P->s20, ny, y, m, s2, q);
    }

    /* e_hat = Q*k (:140); gamma = kstar - k^T e_hat (:144) */
    for (int i = 0; i < b; ++i) e_hat[i] = 0.0;
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < b; ++i) e_hat[i] += Qm(i, j) * k[j];
    double ke = 0.0;
    for (int i = 0; i < b; ++i) ke += k[i] * e_hat[i];
    double gamma = kstar - ke;
    if (gamma < (double)1e-12f) gamma = 0;   /* :146-151 */

    double* s = g->s;
    /* s.head = C*k (:160 / :171) */
    for (int i = 0; i < b; ++i) s[i] = 0.0;
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < b; ++i) s[i] += Cm(i, j) * k[j];

    if (gamma < P->eps_tol && P->capacity != -1) {
        /* sparse update (:155-163) */
        double eta = 1 / (1 + gamma * r);
        for (int i = 0; i < b; ++i) s[i] = s[i] + e_hat[i];      /* s_hat = C*k + e_hat */
        for (int c = 0; c < ny; ++c) {
            double qe = q[c] * eta;
            for (int i = 0; i < b; ++i) Al(c, i) += s[i] * qe;
        }
        double re = r * eta;
        for (int j = 0; j < b; ++j)
            for (int i = 0; i < b; ++i) Cm(i, j) += (re * s[i]) * s[j];
        g->n_sparse++;
        g->last_dec = 0;
    } else {
        /* full update (:164-203) */
        g->last_dec = 1;
        s[b] = (double)1.0f;
        for (int c = 0; c < ny; ++c) {
            Al(c, b) = 0;
            for (int i = 0; i <= b; ++i) Al(c, i) += q[c] * s[i];
        }
        for (int i = 0; i <= b; ++i) { Cm(b, i) = 0; Cm(i, b) = 0; }
        for (int j = 0; j <= b; ++j)
            for (int i = 0; i <= b; ++i) Cm(i, j) += (r * s[i]) * s[j];
        g->BV[2 * b] = x0;
        g->BV[2 * b + 1] = x1;
        for (int i = 0; i <= b; ++i) { Qm(b, i) = 0; Qm(i, b) = 0; }
        e_hat[b] = (double)(-1.0f);
        double ig = (double)1.0f / gamma;
        for (int j = 0; j <= b; ++j)
            for (int i = 0; i <= b; ++i) Qm(i, j) += (ig * e_hat[i]) * e_hat[j];
        g->b = b + 1;
        g->n_full++;
    }

    /* Delete BVs if necessary (:206-223) */
    while (g->b > P->capacity && P->capacity > 0) {
        double minscore = 0, score;
        int minloc = -1;
        for (int i = 0; i < g->b; ++i) {
            double a2 = 0.0;
            for (int c = 0; c < ny; ++c) a2 += Al(c, i) * Al(c, i);   /* alpha(i)^2 / row squaredNorm (field :178) */
            score = a2 / (Qm(i, i) + Cm(i, i));
            if (i == 0 || score < minscore) { minscore = score; minloc = i; }
        }
        orc_sparse_delete_bv(g, minloc);
        if (((g->last_dec >> 1) & 7) < 7) g->last_dec += 2;
    }
    /* Delete for geometric reasons (:226-242) */
    {
        double minscore = 0, score;
        int minloc = -1;
        while (minscore < (double)1e-9f && g->b > 1) {
            for (int i = 0; i < g->b; ++i) {
                score = (double)1.0f / Qm(i, i);
                if (i == 0 || score < minscore) { minscore = score; minloc = i; }
            }
            if (minscore < (double)1e-9f) {
                orc_sparse_delete_bv(g, minloc);
                if (((g->last_dec >> 4) & 7) < 7) g->last_dec += 16;
            }
        }
    }
}

/* src/sparse_gp.hpp:59-86 with the permutation made an explicit input (F7) */
void orc_sparse_add_measurements(orc_sparse* g, int n, const double* x0, const double* x1,
                                 const double* y, const int32_t* perm)
{
    double yy[8];
    for (int i = 0; i < n; ++i) {
        int r = perm ? perm[i] : i;
        for (int c = 0; c < g->p.ny; ++c) yy[c] = y[(size_t)c * n + r];
        orc_sparse_add(g, x0[r], x1[r], yy);
    }
}

/* the same with the decision trace (one byte per point in insertion order; layout in gpc_oracle_hp.c) */
void orc_sparse_add_measurements_trace(orc_sparse* g, int n, const double* x0, const double* x1,
                                       const double* y, const int32_t* perm, uint8_t* trace)
{
    double yy[8];
    for (int i = 0; i < n; ++i) {
        int r = perm ? perm[i] : i;
        for (int c = 0; c < g->p.ny; ++c) yy[c] = y[(size_t)c * n + r];
        orc_sparse_add(g, x0[r], x1[r], yy);
        if (trace) trace[i] = g->last_dec;
    }
}

/* src/sparse_gp.hpp:299-351, src/sparse_gp_field.hpp:268-320 */
void orc_sparse_predict(const orc_sparse* g, int m, const double* xs0, const double* xs1,
                        double* f_star, double* sigconf, int conf)
{
    const size_t ld = (size_t)g->ld;
    const orc_sparse_params* P = &g->p;
    const int b = g->b, ny = P->ny;
    double* k = (double*)malloc(sizeof(double) * (size_t)(b > 0 ? b : 1));
    for (int p = 0; p < m; ++p) {
        double kstar = orc_rbf_kernel(P->p0, P->p1, xs0[p], xs1[p], xs0[p], xs1[p]);
        for (int i = 0; i < b; ++i) k[i] = orc_rbf_kernel(P->p0, P->p1, xs0[p], xs1[p], g->BV[2 * i], g->BV[2 * i + 1]);
        double sigma;
        if (b == 0) {
            for (int c = 0; c < ny; ++c) f_star[(size_t)c * m + p] = 0;
            sigma = kstar + P->s20;
        } else {
            for (int c = 0; c < ny; ++c) {
                double s = 0.0;
                for (int i = 0; i < b; ++i) s += Al(c, i) * k[i];
                f_star[(size_t)c * m + p] = s;
            }
            double kCk = 0.0;
            for (int j = 0; j < b; ++j) {
                double tj = 0.0;
                for (int i = 0; i < b; ++i) tj += k[i] * Cm(i, j);
                kCk += tj * k[j];
            }
            sigma = P->s20 + kstar + kCk;
        }
        if (sigma < 0) sigma = 0;                         /* :334-337 (reference also prints) */
        if (conf) {
            sigma /= kstar + P->s20;
            sigma = (double)100.0f * ((double)1.0f - sigma);
        } else {
            sigma = sqrt(sigma);
        }
        if (sigconf) sigconf[p] = sigma;
    }
    free(k);
}

/* ---- registration inner loop ("next" row f1 of SURVEY section 8) --------------------------------------------------
 * sparse_gp::compute_likelihoods -> likelihood (sparse_gp.hpp:387-427), compute_derivatives -> likelihood_dx
 * (:463-508), kernel derivative rbf_kernel::kernel_dx (rbf_kernel.cpp:33-41); field variants
 * sparse_gp_field.hpp:322-349 and :353-392.  y holds ny planes of n.  l[n]; dX[n][3] row = point, columns as the
 * reference fills them (d/dy, d/dx0, d/dx1; the field variant writes 0 into column 0).  Float literals as written. */
void orc_sparse_likelihood(const orc_sparse* g, int n, const double* x0, const double* x1, const double* y,
                           double* dX, double* l)
{
    const size_t ld = (size_t)g->ld;
    const orc_sparse_params* P = &g->p;
    const int b = g->b, ny = P->ny;
    double* k = (double*)malloc(sizeof(double) * (size_t)(b > 0 ? b : 1));
    double* v = (double*)malloc(sizeof(double) * (size_t)(b > 0 ? b : 1));
    for (int p = 0; p < n; ++p) {
        const double kstar = orc_rbf_kernel(P->p0, P->p1, x0[p], x1[p], x0[p], x1[p]);
        for (int i = 0; i < b; ++i) k[i] = orc_rbf_kernel(P->p0, P->p1, x0[p], x1[p], g->BV[2 * i], g->BV[2 * i + 1]);
        /* v = C k;  kCk = k^T C k */
        double kCk = 0.0;
        for (int i = 0; i < b; ++i) {
            double s = 0.0;
            for (int j = 0; j < b; ++j) s += Cm(i, j) * k[j];
            v[i] = s;
            kCk += k[i] * s;
        }
        double mu[3] = {0, 0, 0}, off[3], sq = 0.0;
        for (int c = 0; c < ny; ++c) {
            for (int i = 0; i < b; ++i) mu[c] += Al(c, i) * k[i];
            off[c] = y[(size_t)c * n + p] - mu[c];
            sq += off[c] * off[c];
        }
        /* likelihood: b == 0 -> prior around 0 (the same expression, empty sums) */
        const double sigma = P->s20 + kstar + kCk;
        if (l) {
            double norm = (ny == 1) ? (double)2.0f * M_PI * sigma : pow((double)2.0f * M_PI, (double)ny) * sigma;
            l[p] = (double)1.0f / sqrt(norm) * exp((double)(-0.5f) / sigma * sq);
        }
        if (dX) {
            /* kernel_dx row i: -p0/p1 * (x - BV_i) * exp(-0.5f/p1 |x - BV_i|^2)  (rbf_kernel.cpp:38-40); k_star_dx = 0 */
            double sdx[2] = {0, 0}, kda[2][3] = {{0, 0, 0}, {0, 0, 0}};
            for (int i = 0; i < b; ++i) {
                const double d0 = x0[p] - g->BV[2 * i], d1 = x1[p] - g->BV[2 * i + 1];
                const double e = exp((double)(-0.5f) / P->p1 * (d0 * d0 + d1 * d1));
                const double g0 = -P->p0 / P->p1 * d0 * e, g1 = -P->p0 / P->p1 * d1 * e;
                sdx[0] += g0 * v[i];                   /* k_dx^T (C k) */
                sdx[1] += g1 * v[i];
                for (int c = 0; c < ny; ++c) {         /* k_dx^T alpha  (2 x ny) */
                    kda[0][c] += g0 * Al(c, i);
                    kda[1][c] += g1 * Al(c, i);
                }
            }
            const double sigma_d = P->s20 + kCk + kstar;      /* likelihood_dx associates differently from likelihood (:485) */
            const double sqrtsigma = sqrt(sigma_d);
            const double exppart = (double)0.5f / (sigma_d * sqrtsigma) * exp((double)(-0.5f) / sigma_d * sq);
            for (int d = 0; d < 2; ++d) {
                const double sigma_dx = (double)2.0f * sdx[d];
                double ko = 0.0;
                for (int c = 0; c < ny; ++c) ko += kda[d][c] * off[c];
                const double first = -sigma_dx, second = (double)2.0f * ko, third = sigma_dx / sigma_d * sq;
                dX[(size_t)p * 3 + 1 + d] = exppart * (first + second + third);
            }
            dX[(size_t)p * 3] = (ny == 1) ? (double)(-1.0f) / (sigma_d * sqrtsigma) * off[0] * exppart : 0.0;
        }
    }
    free(k);
    free(v);
}

/* ---- f4: the live part of sparse_gp::train_parameters (sparse_gp.hpp:586-640, up to the exit(0) at :640) on a trained
 * state: gradient ascent on kernel.param()(0) = sigma_f^2 with the state (alpha, C, BV) held fixed, exactly as the inner
 * do-loop does --
 *     delta = sum_i likelihood_dtheta(x_i, y_i)      (:510-519; kernel_dtheta rbf_kernel.cpp:49-58)
 *     p(0) += step * delta(0)                        (:624)
 *     ls.push_back( sum_i log_likelihood(x_i, y_i) ) (:625-627, log_likelihood :356-385, evaluated with the NEW p(0))
 *     if (counter > max_counter) break; ++counter;   (:630-633, max_counter = 100 upstream)
 * while (delta.norm() > 1e-2f)                       (:636)
 * and the DEBUG early return for fewer than 20 basis vectors (:609-611) -> iters = 0.  ny == 1 only (sparse_gp_field has no
 * such method).  The object's own kernel parameter is left alone; the trained value comes back in *p0_out.  ls must hold
 * max_counter + 2 entries.  log_likelihood's long double temporaries (:380-382) are evaluated in double. */
void orc_sparse_train_sigmaf(const orc_sparse* g, int n, const double* x0, const double* x1, const double* y, double step,
                             int max_counter, double* p0_out, int32_t* iters, double* ls, double* delta_out)
{
    const size_t ld = (size_t)g->ld;
    const int b = g->b;
    const double p1 = g->p.p1, s20 = g->p.s20;
    double p0 = g->p.p0;
    *p0_out = p0;
    *iters = 0;
    delta_out[0] = delta_out[1] = 0.0;
    if (b < 20) return;
    const double logsqrt2pi = (double)0.5f * log((double)2.0f * M_PI);
    double* k = (double*)malloc(sizeof(double) * (size_t)b);
    int counter = 0;
    double delta[2];
    do {
        delta[0] = delta[1] = 0.0;
        for (int i = 0; i < n; ++i) {                         /* likelihood_dtheta */
            double ak = 0.0, kd0 = 0.0, kd1 = 0.0;
            for (int j = 0; j < b; ++j) {
                const double d0 = x0[i] - g->BV[2 * j], d1 = x1[i] - g->BV[2 * j + 1];
                const double offset = d0 * d0 + d1 * d1;
                const double e = exp((double)(-0.5f) / p1 * offset);          /* k_dtheta(j, 0) */
                const double t1 = p0 * (double)0.5f / (p1 * p1) * offset * e; /* k_dtheta(j, 1) */
                ak += Al(0, j) * (p0 * e);                                    /* alpha^T k */
                kd0 += e * Al(0, j);
                kd1 += t1 * Al(0, j);
            }
            delta[0] += (ak - y[i]) * kd0;
            delta[1] += (ak - y[i]) * kd1;
        }
        p0 += step * delta[0];
        double lsum = 0.0;
        for (int i = 0; i < n; ++i) {                         /* log_likelihood with the updated parameter */
            for (int j = 0; j < b; ++j) k[j] = orc_rbf_kernel(p0, p1, x0[i], x1[i], g->BV[2 * j], g->BV[2 * j + 1]);
            const double kstar = orc_rbf_kernel(p0, p1, x0[i], x1[i], x0[i], x1[i]);
            double mu = 0.0, kCk = 0.0;
            for (int r = 0; r < b; ++r) {
                double sum = 0.0;
                for (int c = 0; c < b; ++c) sum += Cm(r, c) * k[c];
                kCk += k[r] * sum;
                mu += k[r] * Al(0, r);
            }
            const double sigma = s20 + kstar + kCk;
            const double cent2 = (y[i] - mu) * (y[i] - mu);
            lsum += -logsqrt2pi - (double)0.5f * log(sigma) - (double)0.5f * cent2 / sigma;
        }
        ls[counter] = lsum;
        *iters = counter + 1;
        if (counter > max_counter) break;
        ++counter;
    } while (sqrt(delta[0] * delta[0] + delta[1] * delta[1]) > (double)1e-2f);
    free(k);
    *p0_out = p0;
    delta_out[0] = delta[0];
    delta_out[1] = delta[1];
}

void orc_sparse_get_state(const orc_sparse* g, double* alpha, double* C, double* Q, double* BV)
{
    const size_t ld = (size_t)g->ld;
    const int b = g->b;
    if (alpha)
        for (int c = 0; c < g->p.ny; ++c)
            for (int i = 0; i < b; ++i) alpha[(size_t)c * b + i] = Al(c, i);
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < b; ++i) {
            if (C) C[i + (size_t)j * b] = Cm(i, j);
            if (Q) Q[i + (size_t)j * b] = Qm(i, j);
        }
    if (BV) memcpy(BV, g->BV, sizeof(double) * 2 * (size_t)b);
}

void orc_sparse_get_counters(const orc_sparse* g, int32_t* n_full, int32_t* n_sparse, int32_t* n_deleted)
{
    if (n_full) *n_full = g->n_full;
    if (n_sparse) *n_sparse = g->n_sparse;
    if (n_deleted) *n_deleted = g->n_deleted;
}

/* src/sparse_gp.hpp:43-56 */
void orc_shuffle_libc(int n, int32_t* ind)
{
    for (int i = 0; i < n; ++i) ind[i] = i;
    for (int i = n - 1; i > 0; --i) {
        int r = rand() % i;
        int32_t temp = ind[i];
        ind[i] = ind[r];
        ind[r] = temp;
    }
}

void orc_shuffle_stream(int n, const uint32_t* rs, int32_t* ind)
{
    for (int i = 0; i < n; ++i) ind[i] = i;
    int t = 0;
    for (int i = n - 1; i > 0; --i) {
        int r = (int)(rs[t++] % (uint32_t)i);
        int32_t temp = ind[i];
        ind[i] = ind[r];
        ind[r] = temp;
    }
}

/* ------------------------------------------------------------------ a15 */

/* src/gp_compressor.cpp:317-332: y outer, x inner; X*(p,0) from x, X*(p,1) from y */
void orc_grid(double res, int sz, double* xs0, double* xs1)
{
    int points = 0;
    for (int y = 0; y < sz; ++y) {
        for (int x = 0; x < sz; ++x) {
            xs0[points] = res * (((double)x + (double)0.5f) / (double)sz - (double)0.5f);
            xs1[points] = res * (((double)y + (double)0.5f) / (double)sz - (double)0.5f);
            ++points;
        }
    }
}

/* src/gp_compressor.cpp:335-343: pt = R*(f, X*(m,0), X*(m,1)) + mean, stored as float */
void orc_reproject(const double* R, const double* mean, double f, double a, double b, float* xyz)
{
    for (int i = 0; i < 3; ++i) {
        double v = R[i + 0] * f + R[i + 3] * a + R[i + 6] * b;
        xyz[i] = (float)(v + mean[i]);
    }
}

/* src/gp_compressor.cpp:251-265.  x.cast<short>() is evaluated as on x86-64 (cvttsd2si then
 * truncation to 16 bits), so e.g. 40000.0 wraps negative and flattens to 0, as the reference binary would. */
void orc_flatten_colors(const double* c3, uint8_t* rgb)
{
    for (int i = 0; i < 3; ++i) {
        double x = c3[i];
        int v;
        if (isnan(x) || isinf(x)) {
            v = 255;
        } else {
            int32_t w = (x >= 2147483648.0 || x < -2147483648.0) ? INT32_MIN : (int32_t)x;
            int16_t sh = (int16_t)(uint16_t)(uint32_t)w;
            v = sh;
            if (v < 0) v = 0;
            else if (v > 255) v = 255;
        }
        rgb[i] = (uint8_t)v;
    }
}

/* ---- whole-batch driver of the sparse path: for every patch of a ragged batch, gp_compressor::train_processes' per-patch body
 * (src/gp_compressor.cpp:146-163: add_measurements on the patch's rows, in the explicit insertion order, F7) followed by the
 * per-patch body of load_compressed (src/gp_compressor.cpp:333: predict_measurements on the shared grid).  One call covers a
 * patch range, so bench.py's cpu_baseline threads and the parity statistics of the tests spend their time in C, not in a
 * Python loop.  perm (may be NULL = identity) holds patch-local row indices, laid out like the rows (perm + off[i]).
 * f_star: P x ny x m; sigma (P x m), bv_count (P) and f_train (ny planes of N: the prediction at the patch's own points, the
 * reference's training-set RMS block src/gp_compressor.cpp:303-315) may be NULL.  Returns 0, or -1 when an allocation fails. */
int orc_sparse_fit_predict_batch(const orc_sparse_params* p, int max_bv, int P, const int32_t* off,
                                 const double* x0, const double* x1, const double* y, const int32_t* perm,
                                 int m, const double* xs0, const double* xs1,
                                 double* f_star, double* sigma, int32_t* bv_count, double* f_train)
{
    const int ny = p->ny;
    const size_t N = (size_t)off[P];
    orc_sparse* g = orc_sparse_create(p, max_bv);
    if (!g) return -1;
    double yy[8];
    for (int i = 0; i < P; ++i) {
        orc_sparse_reset(g);
        const int lo = off[i], n = off[i + 1] - off[i];
        for (int t = 0; t < n; ++t) {
            int r = lo + (perm ? perm[lo + t] : t);
            for (int c = 0; c < ny; ++c) yy[c] = y[(size_t)c * N + (size_t)r];
            orc_sparse_add(g, x0[r], x1[r], yy);
        }
        orc_sparse_predict(g, m, xs0, xs1, f_star + (size_t)i * (size_t)ny * (size_t)m, sigma ? sigma + (size_t)i * (size_t)m : NULL, 0);
        if (bv_count) bv_count[i] = g->b;
        if (f_train && n > 0) {
            /* the reference's training-set RMS block (src/gp_compressor.cpp:303-315): predict_measurements on the patch's own X */
            double* ft = (double*)malloc(sizeof(double) * (size_t)ny * (size_t)n);
            if (!ft) { orc_sparse_destroy(g); return -1; }
            orc_sparse_predict(g, n, x0 + lo, x1 + lo, ft, NULL, 0);
            for (int c = 0; c < ny; ++c)
                for (int t = 0; t < n; ++t) f_train[(size_t)c * N + (size_t)(lo + t)] = ft[(size_t)c * n + t];
            free(ft);
        }
    }
    orc_sparse_destroy(g);
    return 0;
}
