/*
 * gpc.h -- C-ABI of the MI355X-native per-patch Gaussian-process hot path of gp_compressor.
 *
 * The reference (nilsbore/gp_compressor) has no FFI: its seam is the duck-typed contract that
 * gp_compressor uses on every element of `gps` / `RGB_gps` (src/gp_compressor.h:55-56):
 *
 *     add_measurements(X, y)                      src/sparse_gp.h:40, src/sparse_gp_field.h:37, src/gaussian_process.h:20
 *     predict_measurements(f_star, X_star, sig)   src/sparse_gp.h:41-42, src/sparse_gp_field.h:38-39, src/gaussian_process.h:19
 *     size(), reset()                             src/sparse_gp.h:36,39
 *
 * called once per octree-leaf patch from gp_compressor::train_processes (src/gp_compressor.cpp:121-175) and
 * gp_compressor::load_compressed (src/gp_compressor.cpp:298-380).  A per-object call per patch would serialise
 * the GPU, so the entry points below are the same calls BATCHED over patches: plain pointers and sizes, a ragged
 * CSR batch, no C++ / torch types.  All paths are relative to /root/reference.
 *
 * Layout conventions
 *   - patch i owns rows off[i] .. off[i+1]-1 of x0, x1, y   (off has P+1 entries, off[0] == 0)
 *   - X is SoA: x0[N], x1[N]  == Eigen column-major n x 2 (src/gp_compressor.cpp:146-155)
 *   - y is `ny` planes of N doubles (plane c at y + c*N): ny = 1 depth (VectorXd y), ny = 3 RGB (MatrixXd C n x 3,
 *     column-major), which is what sparse_gp_field::add_measurements takes (src/sparse_gp_field.hpp:46-57)
 *   - f_star is [P][ny][m]; v_star / sigma is [P][m]
 *   - every function returns 0 or a negative errno-style code and NEVER aborts (the reference exit(0)s,
 *     src/gp_compressor.cpp:138,215); per-patch conditions are reported in status[P]
 *   - *_dev entry points take DEVICE pointers and enqueue on the context's HIP stream without synchronising;
 *     the plain entry points take HOST pointers and are synchronous (H2D, launch, D2H).
 *
 * There is no CPU fallback: without a HIP device gpc_ctx_create() fails with GPC_ENODEV.
 */
#ifndef GPC_H
#define GPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPC_VERSION 100 /* 0.1.0 */

/* return codes */
#define GPC_OK 0
#define GPC_EINVAL (-22)  /* bad argument (NULL pointer, negative size, ny not in {1,3}, ...) */
#define GPC_ENOMEM (-12)  /* device or host allocation failed */
#define GPC_ENODEV (-19)  /* no HIP device / device index out of range */
#define GPC_EHIP (-5)     /* a HIP runtime call failed; see gpc_last_error() */
#define GPC_ERANGE (-34)  /* a size exceeds what the kernels support (n > GPC_MAX_POINTS, capacity > GPC_MAX_BV) */

/* per-patch status values */
#define GPC_STATUS_OK 0
#define GPC_STATUS_NOT_SPD 1      /* Cholesky pivot <= 1e-14 (sigmaf_sq + noise): numerically singular (Eigen::LLT would
                                     report NumericalIssue on a pivot <= 0); outputs are NaN */
#define GPC_STATUS_NAN 2          /* state became NaN ("sparse_gp::C has become Nan", src/sparse_gp.hpp:245) */
#define GPC_STATUS_SIGMA_CLAMPED 3 /* predictive sigma^2 < 0 was clamped to 0 (src/sparse_gp.hpp:334-337) */
#define GPC_STATUS_OVERFLOW 4     /* sparse, capacity == -1 only: basis set would exceed GPC_MAX_BV; point skipped */
#define GPC_STATUS_NOT_CONVERGED 5 /* gpc_dense_irls_fit_predict: max_iter Newton steps without max|df| <= tol; outputs are the last iterate */

#define GPC_MAX_POINTS 1024 /* largest n per patch of the dense path (BASELINE config 5) */
#define GPC_MAX_BV 256      /* largest sparse capacity (BASELINE config 4 uses 200) */

/* Hyper-parameters: exactly the constants the reference hard-codes per object. */
typedef struct gpc_params {
    double sigmaf_sq;             /* kernel amplitude: rbf_kernel p(0) (src/rbf_kernel.h:24) / gaussian_process sigmaf^2 */
    double l_sq;                  /* squared length scale: rbf_kernel p(1) / gaussian_process l^2 */
    double noise;                 /* dense: sigman_sq (src/gaussian_process.h:21);  sparse: s20 (src/sparse_gp.h:48) */
    double eps_tol;               /* sparse only: src/sparse_gp.hpp:30 (1e-6f), src/sparse_gp_field.hpp:16 (1e-4f) */
    int32_t capacity;             /* sparse only: max basis vectors; -1 = exact GP (src/sparse_gp.hpp:155,206) */
    int32_t noise_model;          /* 0 gaussian_noise(_3d); 1 probit_noise as written (src/probit_noise.cpp:11-31; its "Phi" is
                                     erf(z)/(2.0f*sqrt(2.0f)), not a CDF); 2 probit_noise with Phi(z) = (1 + erf(z/sqrt 2))/2,
                                     everything else as upstream.  1 and 2: ny == 1 only; never instantiated upstream */
    int32_t ref_double_noise;     /* dense: 1 = add sigman_sq twice like src/gaussian_process.cpp:19-22,59-61 */
    int32_t ref_field_delete_bug; /* sparse ny==3: 1 = multiply like src/sparse_gp_field.hpp:250-253, 0 = divide */
    int32_t want_variance;        /* dense: also compute V_star (src/gaussian_process.cpp:35-43) */
    int32_t reserved;
} gpc_params;

/* gaussian_process(double sigmaf = 0.05, double l = 3, double sigman = 0.04), squared (src/gaussian_process.h:21, .cpp:8-9) */
void gpc_default_params_dense(gpc_params* p);
/* sparse_gp(capacity=100, s0=1e-1f), eps_tol 1e-6f (ny==1)  |  sparse_gp_field(capacity=100, s0=1e2f), eps_tol 1e-4f (ny==3);
 * rbf_kernel(sigmaf_sq = 100e-0f, l_sq = 1) */
void gpc_default_params_sparse(gpc_params* p, int ny);

int gpc_version(void);

/* ---- context: one per process per GPU (owns the device workspace; thread-safe per context) ---------------- */
/* Ownership: objects created from a context (gpc_sparse, gpc_patches) hold a reference on it and may be destroyed before
 * OR after gpc_ctx_destroy, in any order -- neither order aborts or touches freed memory.  gpc_ctx_destroy synchronises
 * the stream and releases the device workspace at once; from then on every call that takes the context, or one of its
 * surviving children, returns GPC_EINVAL, except the children's own destroy functions, which release their device
 * buffers as usual (the last one releases the context's host struct).  The context pointer itself must not be used
 * again once it AND all its children have been destroyed. */
typedef struct gpc_ctx gpc_ctx;
int gpc_ctx_create(gpc_ctx** out, int device);
/* hip_stream: a hipStream_t passed as void*, used as is (NULL is HIP's default stream); GPC_STREAM_OWN selects the
 * non-blocking stream the context created for itself (the initial setting).  Not owned. */
#define GPC_STREAM_OWN ((void*)(intptr_t)-1)
int gpc_ctx_set_stream(gpc_ctx* ctx, void* hip_stream);
int gpc_ctx_synchronize(gpc_ctx* ctx);
/* Device buffers for callers built without the HIP toolchain (the reference is a g++ / CMake project): what the `_dev`
 * entry points need to be chained from plain C++.  gpc_dev_memcpy is synchronous and ordered after the work already
 * enqueued on the context's stream; kind: */
#define GPC_COPY_H2D 1
#define GPC_COPY_D2H 2
#define GPC_COPY_D2D 3
int gpc_dev_malloc(gpc_ctx* ctx, size_t bytes, void** out);
int gpc_dev_free(gpc_ctx* ctx, void* p);
int gpc_dev_memcpy(gpc_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);
/* Page-locked host memory for the buffers handed to the host-pointer entries (gpc_dense_fit_predict[_grid] ...).  Those entries
 * cut the batch into chunks and overlap upload, kernel and download on separate streams; from pinned memory the copies run in
 * place on the SDMA engines, from ordinary (pageable) memory they are first staged through pinned buffers of the context by a
 * threaded memcpy.  A reference-side binding assembles its X, y, C batch anyway (src/gp_compressor.cpp:146-155 copies the
 * lists into matrices): assembling it in gpc_host_alloc memory saves the staging copy. */
int gpc_host_alloc(gpc_ctx* ctx, size_t bytes, void** out);
int gpc_host_free(gpc_ctx* ctx, void* p);
void gpc_ctx_destroy(gpc_ctx* ctx);
/* text of the last failure on this context ("" if none); valid until the next call on the context */
const char* gpc_last_error(const gpc_ctx* ctx);
/* name of the kernel variant the last dense call dispatched to (for tests and profiles) */
const char* gpc_last_dense_kernel(const gpc_ctx* ctx);

/* ---- dense exact GP: gaussian_process::add_measurements + predict_measurements, batched ------------------- */
/* Replaces, per patch:  gp.add_measurements(X, y); gp.predict_measurements(f_star, X_star, V_star);
 * (src/gaussian_process.cpp:15-45).  Computes K (+noise), its Cholesky factor, alpha, f_star = K*^T alpha and,
 * when params->want_variance and v_star != NULL, V_star.  alpha_out (ny planes of N) may be NULL. */
int gpc_dense_fit_predict(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off,
                          const double* x0, const double* x1, const double* y, int ny,
                          int m, const double* xs0, const double* xs1,
                          double* f_star, double* v_star, double* alpha_out, int32_t* status);
/* Same with device pointers.  n_max >= max_i(off[i+1]-off[i]) and n_total == off[P] are passed by value because
 * `off` lives on the device. */
int gpc_dense_fit_predict_dev(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off, int n_max, int n_total,
                              const double* x0, const double* x1, const double* y, int ny,
                              int m, const double* xs0, const double* xs1,
                              double* f_star, double* v_star, double* alpha_out, int32_t* status);
/* The decompression grid of gp_compressor::load_compressed is the same for every patch and separable:
 * X*(p,0) = res*((x+.5)/sz-.5), X*(p,1) = res*((y+.5)/sz-.5), p = y*sz + x, m = sz*sz
 * (src/gp_compressor.cpp:317-332).  These entry points build it internally and evaluate K* as an outer product
 * of two sz x n factor tables (rounding differs from the point-wise kernel by O(1 ulp) per entry). */
int gpc_dense_fit_predict_grid(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off,
                               const double* x0, const double* x1, const double* y, int ny,
                               double res, int sz, double* f_star, double* alpha_out, int32_t* status);
int gpc_dense_fit_predict_grid_dev(gpc_ctx* ctx, const gpc_params* params, int P, const int32_t* off, int n_max,
                                   int n_total, const double* x0, const double* x1, const double* y, int ny,
                                   double res, int sz, double* f_star, double* alpha_out, int32_t* status);

/* ---- dense GP with the probit functor: Newton / IRLS loop on the GPU (BASELINE config 5) ---------------------------- */
/* The reference never instantiates probit_noise and holds no IRLS loop (its occupancy map is a ray-cast boolean mask,
 * src/gp_mapping.cpp:154-211): what it fixes is the Noise contract, q = dx_ln(y, x, sigma_x), r = dx2_ln(y, x, sigma_x)
 * (src/probit_noise.cpp:11-31), and the kernel.  This entry runs the textbook loop those plug into -- Newton's method for
 * the mode of p(f | y) with labels y = +-1 (Rasmussen & Williams 2006, Alg. 3.1) in IRLS form: with W = -r, g = q at
 * sigma_x = 0, every step is one gaussian_process::add_measurements-style fit (src/gaussian_process.cpp:15-26) with
 * per-point noise 1 / W_i and working targets t_i = f_i + g_i / W_i:
 *     a = (K + W^-1)^-1 t,   f <- K a = t - W^-1 a;    start f = y * f_init;   stop: max_iter solves or max|df| <= tol.
 * params: sigmaf_sq, l_sq (rbf_kernel, src/rbf_kernel.cpp:15-18), noise = s20 of the functor, noise_model = 1 (as written;
 * singular at f = 0, needs f_init > 0) or 2 (proper CDF; f_init = 0 is the textbook start).  y: N labels.
 * Prediction: latent mean f* = K*^T a on X* -- point-wise (xs0, xs1, m) or, when xs0 == NULL, the sz x sz grid of
 * gp_compressor::load_compressed (res, sz; m is ignored).  f_star [P][m]; alpha_out [N] (= a), fhat_out [N] (the mode at the
 * training points), iters [P] (solves performed) and status [P] may be NULL.  Status GPC_STATUS_NOT_CONVERGED: the step cap ended
 * the loop (outputs are the last iterate).  Status GPC_STATUS_NAN: a weight W_i was not
 * finite and positive (with noise_model 1 this is the normal outcome when a step crosses f = 0); outputs of the patch are NaN.
 * Definition and CPU restatement: oracle/gpc_oracle.c (orc_dense_irls_fit). */
typedef struct gpc_irls_params {
    int32_t max_iter;   /* >= 1 */
    int32_t reserved;
    double tol;         /* on max_i |f_new_i - f_i| */
    double f_init;      /* f_i = y_i * f_init before the first step */
} gpc_irls_params;
/* max_iter 20, tol 1e-9, f_init 0 */
void gpc_default_params_irls(gpc_irls_params* p);
int gpc_dense_irls_fit_predict(gpc_ctx* ctx, const gpc_params* params, const gpc_irls_params* irls, int P, const int32_t* off,
                               const double* x0, const double* x1, const double* y, int m, const double* xs0, const double* xs1,
                               double res, int sz, double* f_star, double* alpha_out, double* fhat_out, int32_t* iters,
                               int32_t* status);
int gpc_dense_irls_fit_predict_dev(gpc_ctx* ctx, const gpc_params* params, const gpc_irls_params* irls, int P, const int32_t* off,
                                   int n_max, int n_total, const double* x0, const double* x1, const double* y, int m,
                                   const double* xs0, const double* xs1, double res, int sz, double* f_star, double* alpha_out,
                                   double* fhat_out, int32_t* iters, int32_t* status);
/* The Noise contract itself, evaluated on the DEVICE by the very functions the kernels inline (csrc/gpc_device.h):
 * q[i] = dx_ln(y[i], x[i], sigma_x[i]), r[i] = dx2_ln(...) for noise_model 0 (src/gaussian_noise.cpp:9-18), 1 or 2
 * (src/probit_noise.cpp:11-31).  Host pointers, n triples.  This is what pins the device functors to the compiled
 * reference objects (tests/golden/noise_ref.json). */
int gpc_noise_eval(gpc_ctx* ctx, int noise_model, double s20, int n, const double* y, const double* x, const double* sigma_x,
                   double* q, double* r);

/* ---- sparse online GP: sparse_gp<rbf_kernel, gaussian_noise> / sparse_gp_field<rbf_kernel, gaussian_noise_3d> */
/* One handle holds the persistent state (alpha, C, Q, BV, current_size) of P independent patch GPs on the
 * device: the batched equivalent of `std::vector<sparse_gp<...>> gps` (src/gp_compressor.h:55-56). */
typedef struct gpc_sparse gpc_sparse;
int gpc_sparse_create(gpc_ctx* ctx, const gpc_params* params, int P, int ny, gpc_sparse** out);
void gpc_sparse_destroy(gpc_sparse* g);
/* reset(): src/sparse_gp.hpp:573-582, for all patches */
int gpc_sparse_reset(gpc_sparse* g);
/* add_measurements(X, y) for every patch (src/sparse_gp.hpp:59-86); may be called repeatedly (online growth,
 * src/gp_mapping.cpp:338-339).  perm holds, per patch, the insertion order as patch-local row indices
 * (the reference draws it from libc rand(), src/sparse_gp.hpp:43-56; here it is an explicit input, NULL = identity). */
int gpc_sparse_add(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, const double* y,
                   const int32_t* perm, int32_t* status);
int gpc_sparse_add_dev(gpc_sparse* g, const int32_t* off, int n_max, int n_total, const double* x0, const double* x1,
                       const double* y, const int32_t* perm, int32_t* status);
/* Diagnostic: record the branch decisions of the following add calls.  trace_dev is a DEVICE buffer of n_total bytes (the
 * n_total of those calls), NULL switches it off.  Byte off[i] + t belongs to the t-th point patch i inserted in the call:
 * bit 0: 1 = full update (basis grew, src/sparse_gp.hpp:164-203), 0 = sparse update (:155-163); bits 1-3: capacity
 * deletions that followed (:206-223); bits 4-6: geometric deletions (:226-242); 0x81: first point of an empty GP (:100-114).
 * The CPU oracle and its binary128 arbiter emit the same bytes (oracle/gpc_oracle_hp.c), which is how the tests count the
 * decisions an fp64 implementation takes differently from the exact recursion. */
int gpc_sparse_set_trace(gpc_sparse* g, uint8_t* trace_dev);
/* predict_measurements(f_star, X_star, sigconf, conf) for every patch on one shared X_star (src/sparse_gp.hpp:299-351).
 * sigma may be NULL (the caller in src/gp_compressor.cpp:333-334 discards it); conf selects the 0-100 confidence form. */
int gpc_sparse_predict(gpc_sparse* g, int m, const double* xs0, const double* xs1, double* f_star, double* sigma,
                       int conf, int32_t* status);
int gpc_sparse_predict_dev(gpc_sparse* g, int m, const double* xs0, const double* xs1, double* f_star, double* sigma,
                           int conf, int32_t* status);
/* predict_measurements(f, X_i, sigconf, conf) with every patch on ITS OWN point set (ragged like the add call's batch): what the
 * reference's per-patch training-set RMS block does (src/gp_compressor.cpp:303-315, printed at :381).  Patch i reads rows
 * off[i]..off[i+1]-1 of x0, x1 and writes the same rows of f (ny planes of n_total) and sigma (n_total, may be NULL). */
int gpc_sparse_predict_points(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, double* f, double* sigma,
                              int conf, int32_t* status);
int gpc_sparse_predict_points_dev(gpc_sparse* g, const int32_t* off, int n_total, const double* x0, const double* x1,
                                  double* f, double* sigma, int conf, int32_t* status);
/* Registration inner loop: sparse_gp::compute_derivatives + compute_likelihoods
 * (src/sparse_gp.h:44-45 -> src/sparse_gp.hpp:387-427, 463-508; field: src/sparse_gp_field.h:40-41 -> .hpp:322-392; call site
 * src/gp_registration.cpp:175-195), batched over patches: patch i evaluates its own rows off[i]..off[i+1]-1 of x0, x1 and
 * the ny planes of y against its current state.  dX is [N][3], row = point, columns as the reference fills them
 * (d/dy -- 0 in the field variant --, d/dx0, d/dx1); l is [N].  Either output may be NULL.  Gaussian noise model only. */
int gpc_sparse_likelihood(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, const double* y,
                          double* dX, double* l);
int gpc_sparse_likelihood_dev(gpc_sparse* g, const int32_t* off, int n_total, const double* x0, const double* x1,
                              const double* y, double* dX, double* l);
/* size() of every patch GP (src/sparse_gp.hpp:35-39) -- host pointer */
int gpc_sparse_sizes(gpc_sparse* g, int32_t* bv_count);
/* state read-back for tests (host pointers; each may be NULL): alpha [P][ny][cap1], C,Q [P][cap1][cap1] column-major,
 * BV [P][cap1][2], with cap1 = gpc_sparse_ld(g).  C and Q are symmetric matrices stored in full; the add kernels may work on
 * one triangle and mirror it (a state that went through them at capacity > 64 comes back exactly symmetric, a smaller one
 * carries the rounding of its rank-one updates in both triangles): a state handed to gpc_sparse_set_state must be symmetric
 * to rounding, as every state of the recursion is. */
int gpc_sparse_get_state(gpc_sparse* g, double* alpha, double* C, double* Q, double* BV);
int gpc_sparse_ld(const gpc_sparse* g);
/* inverse of gpc_sparse_get_state, for loading a stored model (row f3; the reference's save_compressed writes nothing,
 * src/gp_compressor.cpp:21-27): same layouts, host pointers; C and Q may be NULL (zeroed -- enough for the mean prediction) */
int gpc_sparse_set_state(gpc_sparse* g, const int32_t* bv_count, const double* alpha, const double* C, const double* Q,
                         const double* BV);

/* ---- hyper-parameter training: the live part of sparse_gp::train_parameters ---------------- */
/* src/sparse_gp.hpp:586-640 up to the exit(0) at :640 (the call site is commented out upstream, src/gp_compressor.cpp:161):
 * gradient ascent on kernel.param()(0) = sigma_f^2 with the trained state held fixed, per patch and entirely on the device,
 *     do { delta = sum_i likelihood_dtheta(x_i, y_i);          (:510-519, kernel_dtheta src/rbf_kernel.cpp:49-58)
 *          p(0) += step * delta(0);                            (:624, step = 1e-4f upstream)
 *          ls.push_back(sum_i log_likelihood(x_i, y_i));       (:625-627, :356-385)
 *          if (counter > max_counter) break; ++counter;        (:630-633, max_counter = 100 upstream)
 *     } while (delta.norm() > 1e-2f);                          (:636)
 * A patch with fewer than 20 basis vectors is left alone like upstream (:609-611): iters = 0, p0 = the object's value.
 * ny == 1 only.  Outputs per patch: p0[P] the trained sigma_f^2, iters[P], ls[P][max_counter + 2] the likelihood trace the
 * reference plots, delta[P][2] the last gradient.  The object itself is not modified (upstream re-trains in the outer loop
 * the exit(0) cuts off): create a new gpc_sparse with the trained parameter to use it. */
int gpc_sparse_train_sigmaf(gpc_sparse* g, const int32_t* off, const double* x0, const double* x1, const double* y, double step,
                            int max_counter, double* p0, int32_t* iters, double* ls, double* delta);
int gpc_sparse_train_sigmaf_dev(gpc_sparse* g, const int32_t* off, int n_total, const double* x0, const double* x1, const double* y,
                                double step, int max_counter, double* p0, int32_t* iters, double* ls, double* delta);

/* ---- the step after the path: reprojection + colour clamp, fused ------------------------ */
/* The tail of the patch loop of gp_compressor::load_compressed (src/gp_compressor.cpp:335-373, flatten_colors :251-265):
 * pt = R_i (f*, x*_0, x*_1) + mean_i as float, rgb = clamp(short(C* + RGB_mean_i)), written as pcl::PointXYZRGB records.
 * Patches with bv_count[i] == 0 are skipped and the output is compacted in patch order (the reference's `counter`,
 * :299-301); bv_count == NULL means every patch is trained.  f_star [P][m]; c_star [P][3][m] or NULL (colours 0);
 * rotations [P][9] column-major (columns = normal, u, v); means, rgb_means [P][3].  cloud must hold P*m records;
 * n_points receives the number written (device pointer in the _dev variant).  Bit-identical to the CPU oracle. */
typedef struct gpc_point_xyzrgb {   /* memory layout of pcl::PointXYZRGB: 32 bytes */
    float x, y, z, w;               /* w = 1.0f (PCL_ADD_POINT4D) */
    uint8_t b, g, r, a;             /* PCL_ADD_RGB; a = 255 */
    float pad[3];
} gpc_point_xyzrgb;
int gpc_reproject(gpc_ctx* ctx, int P, int m, const int32_t* bv_count, const double* xs0, const double* xs1, const double* f_star,
                  const double* c_star, const double* rotations, const double* means, const double* rgb_means,
                  gpc_point_xyzrgb* cloud, int32_t* n_points);
int gpc_reproject_dev(gpc_ctx* ctx, int P, int m, const int32_t* bv_count, const double* xs0, const double* xs1,
                      const double* f_star, const double* c_star, const double* rotations, const double* means,
                      const double* rgb_means, gpc_point_xyzrgb* cloud, int32_t* n_points);

/* ---- the step before the path: the patch producer on the GPU ---------------------------- */
/* gp_compressor::project_cloud + compute_rotation + project_points (src/gp_compressor.cpp:177-249, 29-64, 66-118): a
 * pcl::PointXYZRGB cloud in, the ragged patch batch of the entry points above out, resident in HBM -- voxel leaves of
 * side `res` (anchored at the cloud's minimum corner, visited in ascending (z, y, x) order), radiusSearch(center,
 * sqrt(3)/2 res) over the 27 neighbouring voxels, plane frame R_i from the smallest singular vector of the homogeneous
 * points (:35-61), exclusive point ownership in leaf order (occupied_indices, :81-89), the +-res/2 window (:85-87), depth
 * and colour mean removal (:101-107), centre shift (:116) and the sz x sz occupancy mask W (:90-92, :117).
 * The arithmetic is the oracle's operation for operation (no FMA contraction): every output is bit-identical to it.
 * gpc_project_cloud takes a HOST cloud, gpc_project_cloud_dev a DEVICE cloud; both synchronise (the sizes of the
 * result depend on the data) and return an object owning the device buffers.  Errors: GPC_EINVAL (res <= 0, sz < 1,
 * a non-finite coordinate), GPC_ERANGE (more than 2^21 voxels along an axis, or 2^62 in total). */
typedef struct gpc_patches gpc_patches;
typedef struct gpc_patches_view {
    int32_t P, n_total, n_max, m;     /* patches (= leaves), points owned in total, largest patch, sz*sz */
    const int32_t* off;               /* P + 1 */
    const double *x0, *x1, *y;        /* n_total: pt(1), pt(2), mean-removed pt(0)   (X and y of :146-155) */
    const double* rgb;                /* 3 planes of n_total: mean-removed colours   (C of :146-155) */
    const double* rotations;          /* P x 9 column-major (columns = normal, u, v) */
    const double *means, *rgb_means;  /* P x 3 */
    const uint8_t* W;                 /* P x m occupancy mask */
    const int32_t* src;               /* n_total: index of the cloud point behind each patch point */
} gpc_patches_view;
int gpc_project_cloud(gpc_ctx* ctx, const gpc_point_xyzrgb* cloud, int n, double res, int sz, gpc_patches** out);
int gpc_project_cloud_dev(gpc_ctx* ctx, const gpc_point_xyzrgb* cloud, int n, double res, int sz, gpc_patches** out);
/* sizes + DEVICE pointers (valid until gpc_patches_destroy): feed them to gpc_dense_fit_predict_grid_dev /
 * gpc_sparse_add_dev / gpc_reproject_dev without a host round trip.  The buffers are READ-ONLY for the caller: the dense entry
 * points recognise the context's most recent batch by its `off` buffer and P, and size their per-size-class launches from the
 * counts the producer took while cutting it. */
int gpc_patches_view_dev(const gpc_patches* p, gpc_patches_view* view);
/* copy to HOST buffers sized by the view's counts; NULL pointers are skipped */
int gpc_patches_fetch(const gpc_patches* p, int32_t* off, double* x0, double* x1, double* y, double* rgb, double* rotations,
                      double* means, double* rgb_means, uint8_t* W, int32_t* src);
void gpc_patches_destroy(gpc_patches* p);

/* ---- patch -> rank partition for one process per GPU (src/gp_compressor.cpp:146-163: patches are independent) ----- */
/* Longest-processing-time assignment of P patches with per-patch cost n_i^3 (dense) or n_i*cap^2 (sparse) onto
 * `world` ranks, every rank padded to ceil(P/world) slots so that the single all-gather of f_star is fixed-size.
 * slot_patch has world*ceil(P/world) entries (patch id or -1 for padding), rank r owns slots [r*S, (r+1)*S). */
int gpc_partition_patches(int P, const int32_t* off, int world, int sparse_capacity, int32_t* slot_patch);

/* ---- multi-GPU: the one exchange step ------------------------------------------------------- */
/* Every rank fits + predicts the S = ceil(P / world) slots gpc_partition_patches gave it; ONE ncclAllGather of the slot buffers
 * over RCCL / xGMI, then a device gather to patch order, reassembles f_star [P][row] on every rank.  RCCL is bound at run time
 * (no link-time dependency; a copy already loaded in the process, e.g. PyTorch's, is shared).
 *   one process per GPU:      rank 0 calls gpc_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel
 *                             the host has (file, socket, MPI); every rank: gpc_comm_create(ctx, world, rank, id, &c).
 *                             A communicator the host created itself goes through gpc_comm_adopt (ncclComm_t as void*; not owned).
 *   one process, N GPUs (how the single-process reference would use a node): one gpc_ctx per device, gpc_comm_create_all, and
 *                             the per-device calls bracketed by gpc_group_start / gpc_group_end.  Inside a bracket the
 *                             collectives are only enqueued at gpc_group_end, so pass f_star = NULL to gpc_allgather_fstar_dev
 *                             there and run gpc_unpermute_fstar_dev per device after the bracket.
 * gpc_comm_set_partition(c, P, slot_patch) takes the table gpc_partition_patches filled (host pointer, world * S entries) and
 * keeps its inverse on the device.  gpc_allgather_fstar_dev: local_f [S][row_doubles] (this rank's slots, padding slots
 * included), gathered [world * S][row_doubles] scratch, f_star [P][row_doubles] or NULL; all DEVICE pointers, enqueued on the
 * context's stream, no synchronisation.  A gpc_comm holds a reference on its context like the other children. */
typedef struct gpc_comm gpc_comm;
#define GPC_UNIQUE_ID_BYTES 128
int gpc_comm_unique_id(void* id128);
int gpc_comm_create(gpc_ctx* ctx, int world, int rank, const void* id128, gpc_comm** out);
int gpc_comm_create_all(int ndev, gpc_ctx* const* ctxs, gpc_comm** out /* ndev */);
int gpc_comm_adopt(gpc_ctx* ctx, void* nccl_comm, int world, int rank, gpc_comm** out);
void gpc_comm_destroy(gpc_comm* c);
int gpc_comm_world(const gpc_comm* c);
int gpc_comm_rank(const gpc_comm* c);
/* path of the RCCL library in use ("" before the first communicator) */
const char* gpc_comm_library(void);
int gpc_comm_set_partition(gpc_comm* c, int P, const int32_t* slot_patch);
int gpc_group_start(void);
int gpc_group_end(void);
int gpc_allgather_fstar_dev(gpc_comm* c, int row_doubles, const double* local_f, double* gathered, double* f_star);
int gpc_unpermute_fstar_dev(gpc_comm* c, int row_doubles, const double* gathered, double* f_star);

/* ---- diagnostics ------------------------------------------------------------------------------------------------ */
/* Host-side evaluation of the table-driven exp() the kernels use for the RBF kernel (same source, csrc/gpc_device.h),
 * so that its error against libm -- which the reference calls, src/rbf_kernel.cpp:17 -- can be bounded without a GPU. */
void gpc_test_exp_host(const double* x, double* out, int n);
/* Same for the small-argument polynomial (-2^-5 <= x <= 0) the register-tile kernel switches to when the patch extent
 * proves the range. */
void gpc_test_exp_small_host(const double* x, double* out, int n);
/* the multi-threaded staging copy of the host-pointer entries (pageable caller buffers), callable without a GPU: concurrency test */
void gpc_test_par_memcpy(void* dst, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* GPC_H */
