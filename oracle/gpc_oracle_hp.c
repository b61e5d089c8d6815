/* gpc_oracle_hp.c -- EXTENDED-PRECISION ARBITER for the sparse online GP (test infrastructure, like the rest of oracle/).
 *
 * The same recursion as gpc_oracle.c's orc_sparse_add / orc_sparse_delete_bv / orc_sparse_predict -- i.e.
 * /root/reference/src/sparse_gp.hpp:89-249, 252-295, 299-351 and the field variant src/sparse_gp_field.hpp:59-215, 219-263,
 * 268-320 -- with every variable, product and sum carried in IEEE binary128 (__float128, 113-bit significand: 60 more bits
 * than the reference's doubles), Gaussian noise only.  Inputs are the same doubles, thresholds are the same float literals
 * promoted, the branch structure is identical.
 *
 * What it is for: the recursion takes data-dependent branches (`gamma < eps_tol`, argmin scores, the geometric deletion
 * threshold) on quantities that, for ill-conditioned kernels, are dominated by fp64 rounding noise.  Two correct fp64
 * implementations (this repo's C oracle, its NumPy restatement, the GPU kernels, the reference's own Eigen build) then
 * disagree with EACH OTHER; the arbiter says what the exact recursion does, so that tests can state a tolerance as
 * "the GPU is no further from the exact recursion than the fp64 CPU oracle is" and count the branch decisions each fp64
 * implementation gets wrong (tests/test_oracle.py, tests/test_sparse_gpu.py).
 *
 * A decision trace (one byte per added point) can be recorded by all three implementations:
 *     bit 0: 1 = full update (basis grew), 0 = sparse (projected) update;   bits 1-3: capacity deletions after the point;
 *     bits 4-6: geometric deletions after the point;   bit 7: first point of an empty GP.
 */
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gpc_oracle.h"

typedef __float128 R;

struct hp_sparse {
    orc_sparse_params p;
    int ld, b;
    R *alpha, *C, *Q, *BV, *k, *e_hat, *s, *t;
};
typedef struct hp_sparse hp_sparse;

#define F(lit) ((R)(double)(lit))   /* a float literal of the reference, promoted like C++ promotes it to double */

static R hp_rbf(const orc_sparse_params* P, R a0, R a1, R b0, R b1)
{
    /* src/rbf_kernel.cpp:15-18: p(0)*exp(-0.5f/p(1)*(xi-xj).squaredNorm()) */
    R d0 = a0 - b0, d1 = a1 - b1;
    return (R)P->p0 * expq(F(-0.5f) / (R)P->p1 * (d0 * d0 + d1 * d1));
}

hp_sparse* hp_sparse_create(const orc_sparse_params* p, int max_bv)
{
    hp_sparse* g = (hp_sparse*)calloc(1, sizeof(hp_sparse));
    if (!g) return NULL;
    g->p = *p;
    g->ld = max_bv < 2 ? 2 : max_bv;
    size_t ld = (size_t)g->ld;
    g->alpha = (R*)calloc(ld * (size_t)p->ny, sizeof(R));
    g->C = (R*)calloc(ld * ld, sizeof(R));
    g->Q = (R*)calloc(ld * ld, sizeof(R));
    g->BV = (R*)calloc(2 * ld, sizeof(R));
    g->k = (R*)calloc(ld, sizeof(R));
    g->e_hat = (R*)calloc(ld, sizeof(R));
    g->s = (R*)calloc(ld, sizeof(R));
    g->t = (R*)calloc(ld, sizeof(R));
    return g;
}

void hp_sparse_destroy(hp_sparse* g)
{
    if (!g) return;
    free(g->alpha); free(g->C); free(g->Q); free(g->BV); free(g->k); free(g->e_hat); free(g->s); free(g->t);
    free(g);
}

int hp_sparse_size(const hp_sparse* g) { return g->b; }

#define Cm(i, j) g->C[(size_t)(i) + (size_t)(j) * ld]
#define Qm(i, j) g->Q[(size_t)(i) + (size_t)(j) * ld]
#define Al(c, i) g->alpha[(size_t)(c) * ld + (size_t)(i)]

/* src/sparse_gp.hpp:252-295, src/sparse_gp_field.hpp:219-263 */
static void hp_delete_bv(hp_sparse* g, int loc)
{
    const size_t ld = (size_t)g->ld;
    const int b = g->b, last = b - 1, ny = g->p.ny;
    R alphastar[8];
    R* Cstar = g->s;
    R* Qstar = g->t;
    for (int c = 0; c < ny; ++c) { alphastar[c] = Al(c, loc); Al(c, loc) = Al(c, last); }
    R cstar = Cm(loc, loc);
    for (int i = 0; i < b; ++i) Cstar[i] = Cm(i, loc);
    Cstar[loc] = Cstar[last];
    {
        R* rep = g->k;
        for (int i = 0; i < b; ++i) rep[i] = Cm(i, last);
        rep[loc] = rep[last];
        for (int i = 0; i < b; ++i) Cm(loc, i) = rep[i];
        for (int i = 0; i < b; ++i) Cm(i, loc) = rep[i];
    }
    R qstar = Qm(loc, loc);
    for (int i = 0; i < b; ++i) Qstar[i] = Qm(i, loc);
    Qstar[loc] = Qstar[last];
    {
        R* rep = g->k;
        for (int i = 0; i < b; ++i) rep[i] = Qm(i, last);
        rep[loc] = rep[last];
        for (int i = 0; i < b; ++i) Qm(loc, i) = rep[i];
        for (int i = 0; i < b; ++i) Qm(i, loc) = rep[i];
    }
    const int nb = b - 1;
    if (ny == 1) {
        R f = alphastar[0] / (qstar + cstar);
        for (int i = 0; i < nb; ++i) Al(0, i) -= f * (Qstar[i] + Cstar[i]);
    } else {
        for (int i = 0; i < nb; ++i) {
            R qc = g->p.field_delete_bug ? (qstar + cstar) * (Qstar[i] + Cstar[i]) : (Qstar[i] + Cstar[i]) / (qstar + cstar);
            for (int c = 0; c < ny; ++c) Al(c, i) -= alphastar[c] * qc;
        }
    }
    for (int j = 0; j < nb; ++j)
        for (int i = 0; i < nb; ++i) {
            R qq = (Qstar[i] * Qstar[j]) / qstar;
            R qc = ((Qstar[i] + Cstar[i]) * (Qstar[j] + Cstar[j])) / (qstar + cstar);
            Cm(i, j) += qq - qc;
            Qm(i, j) -= qq;
        }
    g->BV[2 * loc] = g->BV[2 * last];
    g->BV[2 * loc + 1] = g->BV[2 * last + 1];
    g->b = nb;
}

/* src/sparse_gp.hpp:89-249, src/sparse_gp_field.hpp:59-215; returns the decision byte */
static uint8_t hp_add(hp_sparse* g, double x0d, double x1d, const double* yd)
{
    const size_t ld = (size_t)g->ld;
    const orc_sparse_params* P = &g->p;
    const int ny = P->ny;
    const R x0 = x0d, x1 = x1d, s20 = P->s20;
    R kstar = hp_rbf(P, x0, x1, x0, x1);
    if (g->b == 0) {
        for (int c = 0; c < ny; ++c) Al(c, 0) = (R)yd[c] / (kstar + s20);
        Cm(0, 0) = F(-1.0f) / (kstar + s20);
        Qm(0, 0) = F(1.0f) / kstar;
        g->b = 1;
        g->BV[0] = x0;
        g->BV[1] = x1;
        return 0x81;
    }
    int b = g->b;
    R *k = g->k, *e_hat = g->e_hat, *s = g->s;
    for (int i = 0; i < b; ++i) k[i] = hp_rbf(P, x0, x1, g->BV[2 * i], g->BV[2 * i + 1]);
    R m[8];
    for (int c = 0; c < ny; ++c) {
        R a = 0;
        for (int i = 0; i < b; ++i) a += Al(c, i) * k[i];
        m[c] = a;
    }
    R kCk = 0;
    for (int j = 0; j < b; ++j) {
        R tj = 0;
        for (int i = 0; i < b; ++i) tj += k[i] * Cm(i, j);
        kCk += tj * k[j];
    }
    R s2 = kstar + kCk;
    /* gaussian_noise(_3d)::dx_ln / dx2_ln (src/gaussian_noise.cpp:9-18, src/gaussian_noise_3d.cpp:11-20) */
    R r = F(-1.0f) / (s20 + s2), q[8];
    for (int c = 0; c < ny; ++c) q[c] = ((R)yd[c] - m[c]) / (s20 + s2);
    for (int i = 0; i < b; ++i) e_hat[i] = 0;
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < b; ++i) e_hat[i] += Qm(i, j) * k[j];
    R ke = 0;
    for (int i = 0; i < b; ++i) ke += k[i] * e_hat[i];
    R gamma = kstar - ke;
    if (gamma < F(1e-12f)) gamma = 0;
    for (int i = 0; i < b; ++i) s[i] = 0;
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < b; ++i) s[i] += Cm(i, j) * k[j];
    uint8_t dec = 0;
    if (gamma < (R)P->eps_tol && P->capacity != -1) {
        R eta = 1 / (1 + gamma * r);
        for (int i = 0; i < b; ++i) s[i] = s[i] + e_hat[i];
        for (int c = 0; c < ny; ++c) {
            R qe = q[c] * eta;
            for (int i = 0; i < b; ++i) Al(c, i) += s[i] * qe;
        }
        R re = r * eta;
        for (int j = 0; j < b; ++j)
            for (int i = 0; i < b; ++i) Cm(i, j) += (re * s[i]) * s[j];
    } else {
        dec = 1;
        s[b] = F(1.0f);
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
        e_hat[b] = F(-1.0f);
        R ig = F(1.0f) / gamma;
        for (int j = 0; j <= b; ++j)
            for (int i = 0; i <= b; ++i) Qm(i, j) += (ig * e_hat[i]) * e_hat[j];
        g->b = b + 1;
    }
    int ncap = 0, ngeo = 0;
    while (g->b > P->capacity && P->capacity > 0) {
        R minscore = 0, score;
        int minloc = -1;
        for (int i = 0; i < g->b; ++i) {
            R a2 = 0;
            for (int c = 0; c < ny; ++c) a2 += Al(c, i) * Al(c, i);
            score = a2 / (Qm(i, i) + Cm(i, i));
            if (i == 0 || score < minscore) { minscore = score; minloc = i; }
        }
        hp_delete_bv(g, minloc);
        ++ncap;
    }
    {
        R minscore = 0, score;
        int minloc = -1;
        while (minscore < F(1e-9f) && g->b > 1) {
            for (int i = 0; i < g->b; ++i) {
                score = F(1.0f) / Qm(i, i);
                if (i == 0 || score < minscore) { minscore = score; minloc = i; }
            }
            if (minscore < F(1e-9f)) { hp_delete_bv(g, minloc); ++ngeo; }
        }
    }
    return (uint8_t)(dec | ((ncap > 7 ? 7 : ncap) << 1) | ((ngeo > 7 ? 7 : ngeo) << 4));
}

/* add_measurements with an explicit insertion order; trace (n bytes, in insertion order) may be NULL */
void hp_sparse_add_measurements(hp_sparse* g, int n, const double* x0, const double* x1, const double* y, const int32_t* perm,
                                uint8_t* trace)
{
    double yy[8];
    for (int i = 0; i < n; ++i) {
        int r = perm ? perm[i] : i;
        for (int c = 0; c < g->p.ny; ++c) yy[c] = y[(size_t)c * n + r];
        uint8_t d = hp_add(g, x0[r], x1[r], yy);
        if (trace) trace[i] = d;
    }
}

/* predict_measurements, outputs rounded to double once at the end */
void hp_sparse_predict(const hp_sparse* g, int m, const double* xs0, const double* xs1, double* f_star, double* sigma_out)
{
    const size_t ld = (size_t)g->ld;
    const orc_sparse_params* P = &g->p;
    const int b = g->b, ny = P->ny;
    R* k = (R*)malloc(sizeof(R) * (size_t)(b > 0 ? b : 1));
    for (int p = 0; p < m; ++p) {
        R q0 = xs0[p], q1 = xs1[p];
        R kstar = hp_rbf(P, q0, q1, q0, q1);
        for (int i = 0; i < b; ++i) k[i] = hp_rbf(P, q0, q1, g->BV[2 * i], g->BV[2 * i + 1]);
        R sigma = kstar + (R)P->s20;
        for (int c = 0; c < ny; ++c) {
            R a = 0;
            for (int i = 0; i < b; ++i) a += Al(c, i) * k[i];
            f_star[(size_t)c * m + p] = (double)a;
        }
        R kCk = 0;
        for (int j = 0; j < b; ++j) {
            R tj = 0;
            for (int i = 0; i < b; ++i) tj += k[i] * Cm(i, j);
            kCk += tj * k[j];
        }
        sigma += kCk;
        if (sigma < 0) sigma = 0;
        if (sigma_out) sigma_out[p] = (double)sqrtq(sigma);
    }
    free(k);
}

/* state rounded to double: alpha (ny planes of b), C, Q (b x b column-major, ld = b), BV (2 x b interleaved) */
void hp_sparse_get_state(const hp_sparse* g, double* alpha, double* C, double* Q, double* BV)
{
    const size_t ld = (size_t)g->ld;
    const int b = g->b;
    for (int c = 0; c < g->p.ny; ++c)
        for (int i = 0; i < b; ++i) alpha[(size_t)c * b + i] = (double)Al(c, i);
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < b; ++i) {
            C[(size_t)i + (size_t)j * b] = (double)Cm(i, j);
            Q[(size_t)i + (size_t)j * b] = (double)Qm(i, j);
        }
    for (int i = 0; i < 2 * b; ++i) BV[i] = (double)g->BV[i];
}

/* the arbiter's twin of orc_sparse_fit_predict_batch (gpc_oracle.c): same arguments, same traversal, binary128 arithmetic */
int hp_sparse_fit_predict_batch(const orc_sparse_params* p, int max_bv, int P, const int32_t* off,
                                const double* x0, const double* x1, const double* y, const int32_t* perm,
                                int m, const double* xs0, const double* xs1,
                                double* f_star, double* sigma, int32_t* bv_count, double* f_train)
{
    const int ny = p->ny;
    const size_t N = (size_t)off[P];
    double yy[8];
    for (int i = 0; i < P; ++i) {
        hp_sparse* g = hp_sparse_create(p, max_bv);
        if (!g) return -1;
        const int lo = off[i], n = off[i + 1] - off[i];
        for (int t = 0; t < n; ++t) {
            int r = lo + (perm ? perm[lo + t] : t);
            for (int c = 0; c < ny; ++c) yy[c] = y[(size_t)c * N + (size_t)r];
            hp_add(g, x0[r], x1[r], yy);
        }
        hp_sparse_predict(g, m, xs0, xs1, f_star + (size_t)i * (size_t)ny * (size_t)m, sigma ? sigma + (size_t)i * (size_t)m : NULL);
        if (bv_count) bv_count[i] = g->b;
        if (f_train && n > 0) {
            double* ft = (double*)malloc(sizeof(double) * (size_t)ny * (size_t)n);
            if (!ft) { hp_sparse_destroy(g); return -1; }
            hp_sparse_predict(g, n, x0 + lo, x1 + lo, ft, NULL);
            for (int c = 0; c < ny; ++c)
                for (int t = 0; t < n; ++t) f_train[(size_t)c * N + (size_t)(lo + t)] = ft[(size_t)c * n + t];
            free(ft);
        }
        hp_sparse_destroy(g);
    }
    return 0;
}
