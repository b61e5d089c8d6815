/* gpc_oracle_producer.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see gpc_oracle.h) for the step before the hot path,
 * gp_compressor::project_cloud + compute_rotation + project_points
 * (/root/reference/src/gp_compressor.cpp:177-249, 29-64, 66-118) -- cloud in, per-leaf patch buffers out.
 *
 * What is restated and what is pinned down where the reference leans on PCL / Eigen internals:
 *   - leaves: PCL's octree at resolution `res` is a voxel grid of side `res`; its bounding box and depth-first leaf
 *     order are PCL-internal.  Here the grid is anchored at the cloud's minimum corner and leaves are visited in
 *     ascending (z, y, x) voxel order.
 *   - radiusSearch(center, sqrt(3)/2 res) (:194, :220): exact sphere test over the 27 neighbouring voxels (radius < res);
 *     PCL returns the hits in octree traversal order, here: neighbour voxels in (dz, dy, dx) order, ascending point
 *     index inside a voxel.  This order is the order of the points inside a patch.
 *   - JacobiSVD(points^T).matrixV().col(3) (:35-36) is the eigenvector of the smallest eigenvalue of the 4x4 moment
 *     matrix sum p p^T of the homogeneous points; solved with a cyclic Jacobi sweep (sign fixed by :40-61 as upstream).
 *   - project_points: exclusive ownership through occupied_indices (:81-83, :89), the +-res/2 window (:85-87), mean
 *     removal (:101-107), center shift (:116), occupancy mask W (:90-92, :117).  A leaf that ends up owning no point
 *     keeps its centre (upstream divides 0/0 there).
 * Every floating-point expression is written in the association the GPU kernel (csrc/producer.hip) and the C++ host
 * producer (host/gp_compressor.cpp) use; built with -ffp-contract=off the three agree bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gpc_oracle.h"

typedef struct { int32_t z, y, x, idx; } pkey;

static int pkey_cmp(const void* a, const void* b)
{
    const pkey* p = (const pkey*)a;
    const pkey* q = (const pkey*)b;
    if (p->z != q->z) return p->z < q->z ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

/* leaf id of voxel (x, y, z) in the sorted leaf table, or -1 */
static int find_leaf(const pkey* leaves, int P, int x, int y, int z)
{
    int lo = 0, hi = P - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        const pkey* k = &leaves[mid];
        int c = (k->z != z) ? (k->z < z ? -1 : 1) : (k->y != y) ? (k->y < y ? -1 : 1) : (k->x != x) ? (k->x < x ? -1 : 1) : 0;
        if (c == 0) return mid;
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* eigenvector of the smallest eigenvalue of the symmetric 4x4 matrix A (destroyed): cyclic Jacobi with exact
 * annihilation, run until the off-diagonal part is exactly zero (6-8 sweeps) */
void orc_smallest_eigvec4(double A[4][4], double v[4])
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double offd = 0;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) offd += A[p][q] * A[p][q];
        if (offd == 0.0) break;
        for (int p = 0; p < 4; ++p) {
            for (int q = p + 1; q < 4; ++q) {
                if (A[p][q] == 0.0) continue;
                const double g = 100.0 * fabs(A[p][q]);     /* negligible against both diagonal entries: drop it */
                if (fabs(A[p][p]) + g == fabs(A[p][p]) && fabs(A[q][q]) + g == fabs(A[q][q])) {
                    A[p][q] = A[q][p] = 0.0;
                    continue;
                }
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                A[p][q] = A[q][p] = 0.0;                      /* the rotation annihilates this pair: make it exact */
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
        }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (A[i][i] < A[best][best]) best = i;
    for (int k = 0; k < 4; ++k) v[k] = V[k][best];
}

static void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

static void normalize3(double a[3])
{
    const double n = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n > 0) { a[0] /= n; a[1] /= n; a[2] /= n; }
}

/* src/gp_compressor.cpp:29-64; M = sum of p p^T over the k homogeneous points; R column-major (normal, u, v) */
void orc_compute_rotation(double M[4][4], int k, double R[9])
{
    if (k < 4) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double v[4];
    orc_smallest_eigvec4(M, v);
    double normal[3] = {v[0], v[1], v[2]};
    normalize3(normal);
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, z[3] = {0, 0, 1};
    double c1[3], c2[3];
    const double ax = fabs(normal[0]), ay = fabs(normal[1]), az = fabs(normal[2]);
    if (ax > ay && ax > az) {
        if (normal[0] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
        cross3(z, normal, c1);
    } else if (ay > ax && ay > az) {
        if (normal[1] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
        cross3(x, normal, c1);
    } else {
        if (normal[2] < 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
        cross3(y, normal, c1);
    }
    normalize3(c1);
    cross3(normal, c1, c2);
    for (int a = 0; a < 3; ++a) { R[a] = normal[a]; R[3 + a] = c1[a]; R[6 + a] = c2[a]; }
}

void orc_patches_free(orc_patches* o)
{
    free(o->off); free(o->x0); free(o->x1); free(o->y); free(o->rgb); free(o->R); free(o->mean); free(o->rgb_mean);
    free(o->W); free(o->src);
    memset(o, 0, sizeof(*o));
}

int orc_project_cloud(const float* xyz, const uint8_t* rgb, int n, double res, int sz, orc_patches* out)
{
    memset(out, 0, sizeof(*out));
    out->off = (int32_t*)calloc(1, sizeof(int32_t));
    if (n <= 0) return 0;
    const int m = sz * sz;
    double mn[3] = {xyz[0], xyz[1], xyz[2]};
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a)
            if ((double)xyz[3 * i + a] < mn[a]) mn[a] = xyz[3 * i + a];
    pkey* pts = (pkey*)malloc(sizeof(pkey) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        pts[i].x = (int32_t)floor(((double)xyz[3 * i] - mn[0]) / res);
        pts[i].y = (int32_t)floor(((double)xyz[3 * i + 1] - mn[1]) / res);
        pts[i].z = (int32_t)floor(((double)xyz[3 * i + 2] - mn[2]) / res);
        pts[i].idx = i;
    }
    qsort(pts, (size_t)n, sizeof(pkey), pkey_cmp);
    int P = 0;
    for (int i = 0; i < n; ++i)
        if (i == 0 || pts[i].x != pts[i - 1].x || pts[i].y != pts[i - 1].y || pts[i].z != pts[i - 1].z) ++P;
    pkey* leaves = (pkey*)malloc(sizeof(pkey) * (size_t)P);
    int32_t* lstart = (int32_t*)malloc(sizeof(int32_t) * ((size_t)P + 1));
    for (int i = 0, l = 0; i < n; ++i)
        if (i == 0 || pts[i].x != pts[i - 1].x || pts[i].y != pts[i - 1].y || pts[i].z != pts[i - 1].z) {
            leaves[l] = pts[i];
            lstart[l++] = i;
        }
    lstart[P] = n;

    out->P = P;
    out->off = (int32_t*)realloc(out->off, sizeof(int32_t) * ((size_t)P + 1));
    out->x0 = (double*)malloc(sizeof(double) * (size_t)n);
    out->x1 = (double*)malloc(sizeof(double) * (size_t)n);
    out->y = (double*)malloc(sizeof(double) * (size_t)n);
    out->src = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    double* col = (double*)malloc(sizeof(double) * 3 * (size_t)n);      /* AoS while the total is unknown */
    out->R = (double*)malloc(sizeof(double) * 9 * (size_t)P);
    out->mean = (double*)malloc(sizeof(double) * 3 * (size_t)P);
    out->rgb_mean = (double*)malloc(sizeof(double) * 3 * (size_t)P);
    out->W = (uint8_t*)calloc((size_t)P * (size_t)m + 1, 1);
    uint8_t* occupied = (uint8_t*)calloc((size_t)n, 1);
    int32_t* search = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);

    const double radius = sqrt(3.0f) / 2.0f * res;                       /* :194 */
    const double half = res / 2.0f;
    int total = 0;
    out->off[0] = 0;
    for (int l = 0; l < P; ++l) {
        const pkey key = leaves[l];
        const double center[3] = {mn[0] + (key.x + 0.5) * res, mn[1] + (key.y + 0.5) * res, mn[2] + (key.z + 0.5) * res};
        int k = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int nb = find_leaf(leaves, P, key.x + dx, key.y + dy, key.z + dz);
                    if (nb < 0) continue;
                    for (int s = lstart[nb]; s < lstart[nb + 1]; ++s) {
                        const int gi = pts[s].idx;
                        const double ex = (double)xyz[3 * gi] - center[0], ey = (double)xyz[3 * gi + 1] - center[1],
                                     ez = (double)xyz[3 * gi + 2] - center[2];
                        if (ex * ex + ey * ey + ez * ez <= radius * radius) search[k++] = gi;
                    }
                }
        double M[4][4] = {{0}};
        for (int q = 0; q < k; ++q) {
            const double p4[4] = {xyz[3 * search[q]], xyz[3 * search[q] + 1], xyz[3 * search[q] + 2], 1.0};
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) M[a][b] += p4[a] * p4[b];
        }
        double* R = out->R + 9 * (size_t)l;
        orc_compute_rotation(M, k, R);
        /* project_points (:66-118) */
        const int first = total;
        double mnd = 0, cmean[3] = {0, 0, 0};
        uint8_t* W = out->W + (size_t)l * (size_t)m;
        for (int q = 0; q < k; ++q) {
            const int gi = search[q];
            if (occupied[gi]) continue;
            const double d[3] = {(double)xyz[3 * gi] - center[0], (double)xyz[3 * gi + 1] - center[1], (double)xyz[3 * gi + 2] - center[2]};
            double pt[3];
            for (int a = 0; a < 3; ++a) pt[a] = R[3 * a] * d[0] + R[3 * a + 1] * d[1] + R[3 * a + 2] * d[2];
            if (pt[1] > half || pt[1] < -half || pt[2] > half || pt[2] < -half) continue;
            mnd += pt[0];
            occupied[gi] = 1;
            int gx = (int)((double)sz * (pt[1] / res + 0.5f)), gy = (int)((double)sz * (pt[2] / res + 0.5f));
            gx = gx < 0 ? 0 : (gx > sz - 1 ? sz - 1 : gx);
            gy = gy < 0 ? 0 : (gy > sz - 1 ? sz - 1 : gy);
            W[sz * gx + gy] = 1;
            out->y[total] = pt[0];
            out->x0[total] = pt[1];
            out->x1[total] = pt[2];
            out->src[total] = gi;
            for (int a = 0; a < 3; ++a) {
                col[3 * (size_t)total + a] = rgb[3 * gi + a];
                cmean[a] += rgb[3 * gi + a];
            }
            ++total;
        }
        const int cnt = total - first;
        double* mid = out->mean + 3 * (size_t)l;
        for (int a = 0; a < 3; ++a) mid[a] = center[a];
        if (cnt > 0) {
            mnd /= (double)cnt;
            for (int a = 0; a < 3; ++a) cmean[a] /= (double)cnt;
            for (int q = first; q < total; ++q) out->y[q] -= mnd;
            for (int a = 0; a < 3; ++a) mid[a] += mnd * R[a];
        }
        for (int q = first; q < total; ++q)
            for (int a = 0; a < 3; ++a) col[3 * (size_t)q + a] -= cmean[a];
        for (int a = 0; a < 3; ++a) out->rgb_mean[3 * (size_t)l + a] = cmean[a];
        out->off[l + 1] = total;
    }
    out->n_total = total;
    out->rgb = (double*)malloc(sizeof(double) * 3 * (size_t)(total > 0 ? total : 1));
    for (int q = 0; q < total; ++q)
        for (int a = 0; a < 3; ++a) out->rgb[(size_t)a * (size_t)total + q] = col[3 * (size_t)q + a];
    free(col); free(occupied); free(search); free(pts); free(leaves); free(lstart);
    return 0;
}
