/*
 * gpc_oracle.h -- CPU ORACLE for the per-patch GP regression hot path of
 * nilsbore/gp_compressor.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * shipped library (gp_compressor_amd/csrc -> libgpc_hip.so) never links,
 * loads or calls anything in oracle/.
 *
 * It is a plain-C restatement (no Eigen) of the reference arithmetic; every
 * function cites the reference file:line (relative to /root/reference) it
 * follows.  All arithmetic is IEEE double like the reference; `float`
 * literals of the reference (1e-6f, 1e-1f, -0.5f ...) are reproduced as
 * (double)(float) values.
 *
 * PARITY PINNING.  The reference holds no tests, fixtures or golden vectors
 * for this path and its Eigen/PCL sources cannot be compiled
 * here (F11; no Eigen headers, no network).  What *is* pinned:
 *   - gaussian_noise / probit_noise: against the reference's own objects
 *     compiled from /root/reference/src (oracle/_ref, see oracle/Makefile),
 *     golden values committed in tests/golden/noise_ref.json;
 *   - everything else: by the known-answer identities of SURVEY.md section 8(c)
 *     (closed-form 1/2-point updates, capacity=-1 == exact GP, Q*K_BV = I,
 *     dense path vs an independent LAPACK solve of K + 2 sigma_n^2 I), and by
 *     an independent NumPy restatement (tests/np_restatement.py).
 * At the Eigen boundary (summation order inside products / LLT) parity is
 * UNPINNED: results are determined up to rounding only.
 *
 * Layouts: X is "SoA" = Eigen column-major n x 2, i.e. x0[n], x1[n].
 * Dense matrices are column-major with an explicit leading dimension.
 */
#ifndef GPC_ORACLE_H
#define GPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1/a2: rbf_kernel (src/rbf_kernel.cpp:15-18, 61-71; defaults rbf_kernel.h:24) */
double orc_rbf_kernel(double p0, double p1, double xi0, double xi1, double xj0, double xj1);
/* K is b x N column-major (ld = b); X is 2 x N (x0,x1 SoA), BV is 2 x b interleaved (AoS) */
void orc_rbf_construct_covariance_fast(double p0, double p1, int N, const double* x0, const double* x1,
                                       int b, const double* BV, double* K);

/* ---- a3: gaussian_noise (src/gaussian_noise.cpp:9-18) */
double orc_gaussian_dx_ln(double s20, double y, double x, double sigma_x);
double orc_gaussian_dx2_ln(double s20, double y, double x, double sigma_x);
/* ---- a4: gaussian_noise_3d (src/gaussian_noise_3d.cpp:11-20); q has ny entries */
void orc_gaussian3d_dx_ln(double s20, int ny, const double* y, const double* x, double sigma_x, double* q);
double orc_gaussian3d_dx2_ln(double s20, double sigma_x);
/* ---- a5: probit_noise (src/probit_noise.cpp:11-31) */
double orc_probit_dx_ln(double s20, double y, double x, double sigma_x);
double orc_probit_dx2_ln(double s20, double y, double x, double sigma_x);

/* ---- a6-a8: gaussian_process (dense exact GP), src/gaussian_process.cpp:15-64 */
typedef struct {
    double sigmaf_sq, l_sq, sigman_sq;  /* ctor squares its args: gaussian_process.cpp:8-9 */
    int ref_double_noise;               /* 1 = reference behaviour (noise added twice, F5) */
} orc_dense_params;

/* defaults of gaussian_process.h:21 (sigmaf=0.05, l=3, sigman=0.04), squared */
void orc_dense_default_params(orc_dense_params* p);

/* Fit: builds K (+noise), LLT, alpha = chol.solve(y).  y has ny planes of n (ny>=1).
 * L (n x n col-major, ld=n, lower; upper left untouched/zero) and alpha (ny planes of n) are outputs.
 * Returns 0, or 1+j if pivot j is not positive (Eigen would report NumericalIssue). */
int orc_dense_fit(const orc_dense_params* p, int n, const double* x0, const double* x1,
                  const double* y, int ny, double* L, double* alpha);
/* Predict: f_star (ny planes of m); v_star (m) may be NULL (variance skipped). */
void orc_dense_predict(const orc_dense_params* p, int n, const double* x0, const double* x1,
                       const double* L, const double* alpha, int ny,
                       int m, const double* xs0, const double* xs1, double* f_star, double* v_star);

/* ---- C5: dense GP + probit functor, Newton / IRLS loop (extension; definition in gpc_oracle.c).
 * p->sigman_sq is s20 of the functor; noise_model 1 = probit_noise as written, 2 = with a proper CDF. */
double orc_probit_std_dx_ln(double s20, double y, double x, double sigma_x);
double orc_probit_std_dx2_ln(double s20, double y, double x, double sigma_x);
int orc_dense_irls_fit(const orc_dense_params* p, int noise_model, int n, const double* x0, const double* x1, const double* y,
                       int max_iter, double tol, double f_init, double* alpha, double* fhat, int32_t* iters);
int orc_dense_irls_fit_predict_batch(const orc_dense_params* p, int noise_model, int max_iter, double tol, double f_init,
                                     int P, const int32_t* off, const double* x0, const double* x1, const double* y,
                                     int m, const double* xs0, const double* xs1, double* f_star, double* alpha_out,
                                     double* fhat_out, int32_t* iters, int32_t* status);

/* ---- a9-a13: sparse_gp / sparse_gp_field (src/sparse_gp.hpp, src/sparse_gp_field.hpp) */
typedef struct {
    double p0, p1;        /* rbf_kernel params: sigmaf_sq, l_sq (rbf_kernel.h:24: 100, 1) */
    double s20;           /* sparse_gp.h:48 (1e-1f) / sparse_gp_field.h:43 (1e2f) */
    double eps_tol;       /* sparse_gp.hpp:30 (1e-6f) / sparse_gp_field.hpp:16 (1e-4f) */
    int capacity;         /* 100 default; -1 = never sparse-update, never delete (exact GP) */
    int ny;               /* 1 = sparse_gp, 3 = sparse_gp_field */
    int noise_model;      /* 0 gaussian, 1 probit as written, 2 probit with a proper CDF (ny==1 only; F6 extension) */
    int field_delete_bug; /* 1 = reproduce sparse_gp_field.hpp:250-253 (F8); only used when ny>1 */
} orc_sparse_params;

void orc_sparse_default_params(orc_sparse_params* p, int ny);

typedef struct orc_sparse orc_sparse;
/* max_bv: upper bound for the number of basis vectors ever held (capacity+1, or n_total for capacity=-1) */
orc_sparse* orc_sparse_create(const orc_sparse_params* p, int max_bv);
void orc_sparse_destroy(orc_sparse* g);
void orc_sparse_reset(orc_sparse* g);                       /* sparse_gp.hpp:573-582 */
int orc_sparse_size(const orc_sparse* g);                   /* sparse_gp.hpp:35-39 */
int orc_sparse_total_count(const orc_sparse* g);
/* one point: sparse_gp.hpp:89-249 / sparse_gp_field.hpp:59-215. y has ny entries. */
void orc_sparse_add(orc_sparse* g, double x0, double x1, const double* y);
/* add_measurements with an EXPLICIT insertion order (F7): perm==NULL -> identity.
 * y is ny planes of n (plane c at y + c*n).  sparse_gp.hpp:59-86 */
void orc_sparse_add_measurements(orc_sparse* g, int n, const double* x0, const double* x1,
                                 const double* y, const int32_t* perm);
void orc_sparse_add_measurements_trace(orc_sparse* g, int n, const double* x0, const double* x1,
                                       const double* y, const int32_t* perm, uint8_t* trace);
void orc_sparse_delete_bv(orc_sparse* g, int loc);          /* sparse_gp.hpp:252-295 */
/* predict_measurements: sparse_gp.hpp:299-351; f_star is ny planes of m; sigconf (m) may be NULL */
void orc_sparse_predict(const orc_sparse* g, int m, const double* xs0, const double* xs1,
                        double* f_star, double* sigconf, int conf);
/* state access for tests: alpha (ny planes of b), C, Q (b x b col-major ld=b), BV (2 x b interleaved) */
/* compute_likelihoods / compute_derivatives (sparse_gp.hpp:387-427, 463-508; field: sparse_gp_field.hpp:322-392) */
void orc_sparse_likelihood(const orc_sparse* g, int n, const double* x0, const double* x1, const double* y,
                           double* dX, double* l);
/* the live part of train_parameters (sparse_gp.hpp:586-640): gradient ascent on sigma_f^2 on the trained state; ls holds
 * max_counter + 2 entries, delta_out 2 */
void orc_sparse_train_sigmaf(const orc_sparse* g, int n, const double* x0, const double* x1, const double* y, double step,
                             int max_counter, double* p0_out, int32_t* iters, double* ls, double* delta_out);
void orc_sparse_get_state(const orc_sparse* g, double* alpha, double* C, double* Q, double* BV);
/* statistics: how many full / sparse updates and deletions happened (for BV-count agreement reports) */
void orc_sparse_get_counters(const orc_sparse* g, int32_t* n_full, int32_t* n_sparse, int32_t* n_deleted);

/* ---- binary128 arbiter of the same recursion (gpc_oracle_hp.c, liboracle_hp.so; Gaussian noise only) */
typedef struct hp_sparse hp_sparse;
hp_sparse* hp_sparse_create(const orc_sparse_params* p, int max_bv);
void hp_sparse_destroy(hp_sparse* g);
int hp_sparse_size(const hp_sparse* g);
void hp_sparse_add_measurements(hp_sparse* g, int n, const double* x0, const double* x1, const double* y, const int32_t* perm,
                                uint8_t* trace);
void hp_sparse_predict(const hp_sparse* g, int m, const double* xs0, const double* xs1, double* f_star, double* sigma_out);
void hp_sparse_get_state(const hp_sparse* g, double* alpha, double* C, double* Q, double* BV);

/* sparse_gp::shuffle (sparse_gp.hpp:43-56) with libc rand(), exactly as written (rand() % i) */
void orc_shuffle_libc(int n, int32_t* ind);
/* the same permutation scheme driven by a caller-supplied stream of non-negative ints r[n-1]
 * (r[t] is the t-th rand() value): lets tests fix the order without libc state */
void orc_shuffle_stream(int n, const uint32_t* r, int32_t* ind);

/* ---- a15: decompression grid and reprojection (src/gp_compressor.cpp:317-340, 367-372, 251-265) */
void orc_grid(double res, int sz, double* xs0, double* xs1);           /* m = sz*sz, p = y*sz + x */
/* R is 3x3 column-major, mean 3; out xyz is float[3] like pcl::PointXYZRGB */
void orc_reproject(const double* R, const double* mean, double f, double xs0, double xs1, float* xyz);
void orc_flatten_colors(const double* c3, uint8_t* rgb);               /* gp_compressor.cpp:251-265 */

/* ---- f2: the patch producer, gp_compressor::project_cloud + compute_rotation + project_points
 *      (src/gp_compressor.cpp:177-249, 29-64, 66-118); see gpc_oracle_producer.c for what is pinned down */
typedef struct {
    int P, n_total;
    int32_t* off;                 /* P + 1 */
    double *x0, *x1, *y;          /* n_total: pt(1), pt(2), mean-removed pt(0) */
    double* rgb;                  /* 3 planes of n_total, mean-removed */
    double *R, *mean, *rgb_mean;  /* P x 9 (column-major: normal, u, v), P x 3, P x 3 */
    uint8_t* W;                   /* P x sz*sz occupancy */
    int32_t* src;                 /* n_total: index of the cloud point each patch point came from */
} orc_patches;
int orc_project_cloud(const float* xyz, const uint8_t* rgb, int n, double res, int sz, orc_patches* out);
void orc_patches_free(orc_patches* o);
void orc_smallest_eigvec4(double A[4][4], double v[4]);
void orc_compute_rotation(double M[4][4], int k, double R[9]);

/* ---- whole-batch drivers used by tests and by bench.py's cpu_baseline (kind "port") ---- */
/* dense path for a ragged batch (CSR offsets), same signature shape as gpc_dense_fit_predict in include/gpc.h.
 * scratch is allocated internally.  status[i]: 0 ok, 1 non-SPD pivot. */
int orc_dense_fit_predict_batch(const orc_dense_params* p, int P, const int32_t* off,
                                const double* x0, const double* x1, const double* y, int ny,
                                int m, const double* xs0, const double* xs1,
                                double* f_star, double* v_star, int32_t* status, double* alpha_out);
/* sparse path for a ragged batch: per patch add_measurements (explicit patch-local insertion order, NULL = identity) then
 * predict_measurements on the shared grid (src/gp_compressor.cpp:146-163, 333).  f_star P x ny x m; sigma (P x m) and
 * bv_count (P) may be NULL.  hp_ = the binary128 arbiter running the same traversal. */
int orc_sparse_fit_predict_batch(const orc_sparse_params* p, int max_bv, int P, const int32_t* off,
                                 const double* x0, const double* x1, const double* y, const int32_t* perm,
                                 int m, const double* xs0, const double* xs1,
                                 double* f_star, double* sigma, int32_t* bv_count, double* f_train);
int hp_sparse_fit_predict_batch(const orc_sparse_params* p, int max_bv, int P, const int32_t* off,
                                const double* x0, const double* x1, const double* y, const int32_t* perm,
                                int m, const double* xs0, const double* xs1,
                                double* f_star, double* sigma, int32_t* bv_count, double* f_train);

#ifdef __cplusplus
}
#endif
#endif
