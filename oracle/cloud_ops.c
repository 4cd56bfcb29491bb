/*
 * cloud_ops.c -- CPU ORACLE (test infrastructure).
 * Restates Open3D geometry::PointCloud::{VoxelDownSample, RemoveStatisticalOutliers,
 * EstimateCovariances, EstimateNormals} as reached from the reference call sites
 * ALL_FUNCTIONS.py:293-302 and 2_MGICP_refinement_in_NCLT_dataset.py:146-153
 * (SURVEY.md Appendix A.1-A.4).
 */
#include "oracle_internal.h"
#include <omp.h>

int orc_set_num_threads(int n) { int p = omp_get_max_threads(); if (n > 0) omp_set_num_threads(n); return p; }
int orc_get_num_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------ algebra */
void orc_sym3_eig(const double A[9], double w[3], double V[9]) {
    double a[9]; memcpy(a, A, sizeof a);
    double v[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = a[p * 3 + q];
                if (apq == 0.0) continue;
                double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { /* A <- A J */
                    double akp = a[k * 3 + p], akq = a[k * 3 + q];
                    a[k * 3 + p] = c * akp - s * akq; a[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) { /* A <- J^T A */
                    double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                    a[p * 3 + k] = c * apk - s * aqk; a[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
                    v[k * 3 + p] = c * vkp - s * vkq; v[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = a[0]; w[1] = a[4]; w[2] = a[8];
    memcpy(V, v, sizeof v);
}

void orc_sym3_inv_sqrt(const double M[9], double W[9]) {
    double w[3], V[9];
    orc_sym3_eig(M, w, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += V[i * 3 + k] * V[j * 3 + k] / sqrt(w[k]);
            W[i * 3 + j] = s;
        }
}

int orc_ldlt6_solve(const double *A36, const double *b6, double *x6) {
    /* LDL^T with symmetric (diagonal) pivoting, as Eigen::LDLT */
    double A[36]; memcpy(A, A36, sizeof A);
    int perm[6]; for (int i = 0; i < 6; i++) perm[i] = i;
    double L[36]; memset(L, 0, sizeof L); double D[6];
    for (int k = 0; k < 6; k++) {
        int piv = k; double best = fabs(A[k * 6 + k]);
        for (int i = k + 1; i < 6; i++) if (fabs(A[i * 6 + i]) > best) { best = fabs(A[i * 6 + i]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < 6; j++) { double t = A[k * 6 + j]; A[k * 6 + j] = A[piv * 6 + j]; A[piv * 6 + j] = t; }
            for (int j = 0; j < 6; j++) { double t = A[j * 6 + k]; A[j * 6 + k] = A[j * 6 + piv]; A[j * 6 + piv] = t; }
            for (int j = 0; j < k; j++) { double t = L[k * 6 + j]; L[k * 6 + j] = L[piv * 6 + j]; L[piv * 6 + j] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        double d = A[k * 6 + k];
        D[k] = d; L[k * 6 + k] = 1.0;
        if (d == 0.0 || !isfinite(d)) return ORC_ENUMERIC;
        for (int i = k + 1; i < 6; i++) L[i * 6 + k] = A[i * 6 + k] / d;
        for (int i = k + 1; i < 6; i++)
            for (int j = k + 1; j < 6; j++) A[i * 6 + j] -= L[i * 6 + k] * d * L[j * 6 + k];
    }
    double y[6], z[6];
    for (int i = 0; i < 6; i++) { double s = b6[perm[i]]; for (int j = 0; j < i; j++) s -= L[i * 6 + j] * y[j]; y[i] = s; }
    for (int i = 0; i < 6; i++) y[i] /= D[i];
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < 6; j++) s -= L[j * 6 + i] * z[j]; z[i] = s; }
    for (int i = 0; i < 6; i++) x6[perm[i]] = z[i];
    for (int i = 0; i < 6; i++) if (!isfinite(x6[i])) return ORC_ENUMERIC;
    return ORC_OK;
}

void orc_vec6_to_T(const double *x, double *T) {
    double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
    /* Rz(g) * Ry(b) * Rx(a) */
    T[0] = cg * cb; T[1] = cg * sb * sa - sg * ca; T[2] = cg * sb * ca + sg * sa; T[3] = x[3];
    T[4] = sg * cb; T[5] = sg * sb * sa + cg * ca; T[6] = sg * sb * ca - cg * sa; T[7] = x[4];
    T[8] = -sb;     T[9] = cb * sa;                T[10] = cb * ca;               T[11] = x[5];
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

void orc_rigid_inverse(const double *T, double *Ti) {
    double R[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
    double t[3] = { T[3], T[7], T[11] };
    double out[16];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) out[i * 4 + j] = R[j * 3 + i];
        out[i * 4 + 3] = -(R[0 * 3 + i] * t[0] + R[1 * 3 + i] * t[1] + R[2 * 3 + i] * t[2]);
    }
    out[12] = out[13] = out[14] = 0; out[15] = 1;
    memcpy(Ti, out, sizeof out);
}

/* ------------------------------------------------------- neighbourhood modes */
int64_t orc_neighbourhood(const orc_kdtree *t, const double *q, int mode, int knn, double radius,
                          int32_t **idx, double **d2, int64_t *cap, void *scratch) {
    if (mode == ORC_SEARCH_RADIUS) return orc_kdtree_radius(t, q, radius * radius, idx, d2, cap);
    if (*cap < knn) {
        *cap = knn;
        *idx = (int32_t *)realloc(*idx, sizeof(int32_t) * (size_t)knn);
        *d2 = (double *)realloc(*d2, sizeof(double) * (size_t)knn);
    }
    double r2 = mode == ORC_SEARCH_HYBRID ? radius * radius : INFINITY;
    return orc_kdtree_knn(t, q, knn, r2, *idx, *d2, scratch);
}

/* ------------------------------------------------------------- voxel (A.1) */
typedef struct { int32_t k[3]; int32_t idx; } vox_key;
static int cmp_vox(const void *a, const void *b) {
    const vox_key *x = (const vox_key *)a, *y = (const vox_key *)b;
    for (int d = 0; d < 3; d++) { if (x->k[d] < y->k[d]) return -1; if (x->k[d] > y->k[d]) return 1; }
    return (x->idx > y->idx) - (x->idx < y->idx);   /* keep input order inside a voxel */
}

int orc_voxel_down_sample(const double *xyz, int64_t n, double voxel, double *out_xyz, int64_t *out_n,
                          const double *normals_in, double *normals_out) {
    if (voxel <= 0.0 || n < 0) return ORC_EINVAL;
    *out_n = 0;
    if (n == 0) return ORC_OK;
    double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int64_t i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) { double v = xyz[i * 3 + d]; if (v < mn[d]) mn[d] = v; if (v > mx[d]) mx[d] = v; }
    double org[3];
    for (int d = 0; d < 3; d++) {
        org[d] = mn[d] - voxel * 0.5;
        if (voxel * 2147483647.0 < (mx[d] + voxel * 0.5) - org[d]) return ORC_EINVAL; /* "voxel_size is too small" */
    }
    vox_key *keys = (vox_key *)malloc(sizeof(vox_key) * (size_t)n);
    if (!keys) return ORC_ENOMEM;
    for (int64_t i = 0; i < n; i++) {
        for (int d = 0; d < 3; d++) keys[i].k[d] = (int32_t)floor((xyz[i * 3 + d] - org[d]) / voxel);
        keys[i].idx = (int32_t)i;
    }
    qsort(keys, (size_t)n, sizeof(vox_key), cmp_vox);
    int64_t m = 0, i = 0;
    while (i < n) {
        int64_t j = i; double s[3] = { 0, 0, 0 }, sn[3] = { 0, 0, 0 };
        while (j < n && keys[j].k[0] == keys[i].k[0] && keys[j].k[1] == keys[i].k[1] && keys[j].k[2] == keys[i].k[2]) {
            const double *p = xyz + (int64_t)keys[j].idx * 3;
            s[0] += p[0]; s[1] += p[1]; s[2] += p[2];
            if (normals_in) { const double *q = normals_in + (int64_t)keys[j].idx * 3; sn[0] += q[0]; sn[1] += q[1]; sn[2] += q[2]; }
            j++;
        }
        double c = (double)(j - i);
        out_xyz[m * 3 + 0] = s[0] / c; out_xyz[m * 3 + 1] = s[1] / c; out_xyz[m * 3 + 2] = s[2] / c;
        if (normals_in && normals_out) { normals_out[m * 3 + 0] = sn[0] / c; normals_out[m * 3 + 1] = sn[1] / c; normals_out[m * 3 + 2] = sn[2] / c; }
        m++; i = j;
    }
    free(keys);
    *out_n = m;
    return ORC_OK;
}

/* --------------------------------------------------------------- SOR (A.2) */
int orc_remove_statistical_outlier(const double *xyz, int64_t n, int nb_neighbors, double std_ratio, uint8_t *keep,
                                   double *avg_dist, double *mean_out, double *std_out) {
    if (nb_neighbors < 1 || std_ratio <= 0.0) return ORC_EINVAL;
    if (n == 0) { if (mean_out) *mean_out = 0; if (std_out) *std_out = 0; return ORC_OK; }
    double *avg = avg_dist ? avg_dist : (double *)malloc(sizeof(double) * (size_t)n);
    orc_kdtree *t = orc_kdtree_build(xyz, n, 3);
    size_t sb = orc_kdtree_scratch_bytes(t, nb_neighbors);
    int64_t valid = 0;
#pragma omp parallel reduction(+ : valid)
    {
        void *scratch = malloc(sb);
        int32_t *ti = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb_neighbors);
        double *td = (double *)malloc(sizeof(double) * (size_t)nb_neighbors);
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; i++) {
            int c = orc_kdtree_knn(t, xyz + i * 3, nb_neighbors, INFINITY, ti, td, scratch);
            double mean = -1.0;
            if (c > 0) { valid++; double s = 0; for (int j = 0; j < c; j++) s += sqrt(td[j]); mean = s / c; }
            avg[i] = mean;
        }
        free(scratch); free(ti); free(td);
    }
    orc_kdtree_free(t);
    if (valid == 0) { memset(keep, 0, (size_t)n); if (!avg_dist) free(avg); return ORC_OK; }
    double cm = 0; for (int64_t i = 0; i < n; i++) if (avg[i] > 0) cm += avg[i];
    cm /= (double)valid;
    double sq = 0; for (int64_t i = 0; i < n; i++) if (avg[i] > 0) sq += (avg[i] - cm) * (avg[i] - cm);
    double sd = sqrt(sq / (double)(valid - 1));
    double thr = cm + std_ratio * sd;
    for (int64_t i = 0; i < n; i++) keep[i] = (avg[i] > 0 && avg[i] < thr) ? 1 : 0;
    if (mean_out) *mean_out = cm; if (std_out) *std_out = sd;
    if (!avg_dist) free(avg);
    return ORC_OK;
}

/* ------------------------------------------------------ covariances (A.3) */
static void covariance_from_indices(const double *xyz, const int32_t *idx, int64_t c, double *C) {
    double cu[9] = { 0 };
    for (int64_t j = 0; j < c; j++) {
        const double *p = xyz + (int64_t)idx[j] * 3;
        cu[0] += p[0]; cu[1] += p[1]; cu[2] += p[2];
        cu[3] += p[0] * p[0]; cu[4] += p[0] * p[1]; cu[5] += p[0] * p[2];
        cu[6] += p[1] * p[1]; cu[7] += p[1] * p[2]; cu[8] += p[2] * p[2];
    }
    for (int k = 0; k < 9; k++) cu[k] /= (double)c;
    C[0] = cu[3] - cu[0] * cu[0]; C[4] = cu[6] - cu[1] * cu[1]; C[8] = cu[8] - cu[2] * cu[2];
    C[1] = C[3] = cu[4] - cu[0] * cu[1];
    C[2] = C[6] = cu[5] - cu[0] * cu[2];
    C[5] = C[7] = cu[7] - cu[1] * cu[2];
}

int orc_estimate_covariances(const double *xyz, int64_t n, int mode, int knn, double radius, double *cov9) {
    if (n < 0) return ORC_EINVAL;
    if ((mode == ORC_SEARCH_KNN || mode == ORC_SEARCH_HYBRID) && knn < 1) return ORC_EINVAL;
    if ((mode == ORC_SEARCH_RADIUS || mode == ORC_SEARCH_HYBRID) && radius <= 0) return ORC_EINVAL;
    if (n == 0) return ORC_OK;
    orc_kdtree *t = orc_kdtree_build(xyz, n, 3);
    size_t sb = orc_kdtree_scratch_bytes(t, knn > 0 ? knn : 1);
#pragma omp parallel
    {
        void *scratch = malloc(sb);
        int32_t *ti = NULL; double *td = NULL; int64_t cap = 0;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; i++) {
            int64_t c = orc_neighbourhood(t, xyz + i * 3, mode, knn, radius, &ti, &td, &cap, scratch);
            double *C = cov9 + i * 9;
            if (c >= 3) covariance_from_indices(xyz, ti, c, C);
            else { memset(C, 0, 9 * sizeof(double)); C[0] = C[4] = C[8] = 1.0; }
        }
        free(scratch); free(ti); free(td);
    }
    orc_kdtree_free(t);
    return ORC_OK;
}

/* -------------------------------------- analytic 3x3 eigenvector (A.4) ----
 * Published algorithm: D. Eberly, "A Robust Eigensolver for 3x3 Symmetric
 * Matrices" (Geometric Tools), non-iterative variant: eigenvalues by the
 * trigonometric formula on the scaled matrix, eigenvector 0 from the largest
 * cross product of rows of A - lambda I, eigenvector 1 from the reduced 2x2
 * system in the orthogonal complement, the third by a cross product.          */
static void eigenvector0(const double *A, double ev, double *out) {
    double r0[3] = { A[0] - ev, A[1], A[2] }, r1[3] = { A[1], A[4] - ev, A[5] }, r2[3] = { A[2], A[5], A[8] - ev };
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01); cross3(r0, r2, c02); cross3(r1, r2, c12);
    double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2 = dot3(c12, c12);
    double dmax = d0; int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) { imax = 2; }
    const double *c = imax == 0 ? c01 : (imax == 1 ? c02 : c12);
    double d = imax == 0 ? d0 : (imax == 1 ? d1 : d2);
    double s = sqrt(d);
    out[0] = c[0] / s; out[1] = c[1] / s; out[2] = c[2] / s;
}

static void eigenvector1(const double *A, const double *e0, double ev1, double *out) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv; U[1] = 0; U[2] = e0[0] * inv;
    } else {
        double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0; U[1] = e0[2] * inv; U[2] = -e0[1] * inv;
    }
    cross3(e0, U, V);
    double AU[3] = { A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[4] * U[1] + A[5] * U[2], A[2] * U[0] + A[5] * U[1] + A[8] * U[2] };
    double AV[3] = { A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[4] * V[1] + A[5] * V[2], A[2] * V[0] + A[5] * V[1] + A[8] * V[2] };
    double m00 = dot3(U, AU) - ev1, m01 = dot3(U, AV), m11 = dot3(V, AV) - ev1;
    double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        double mx = a00 > a01 ? a00 : a01;
        if (mx > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
            for (int k = 0; k < 3; k++) out[k] = m01 * U[k] - m00 * V[k];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    } else {
        double mx = a11 > a01 ? a11 : a01;
        if (mx > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
            for (int k = 0; k < 3; k++) out[k] = m11 * U[k] - m01 * V[k];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    }
}

void orc_fast_eigen3x3(const double cov[9], double normal[3]) {
    double A[9]; memcpy(A, cov, sizeof A);
    double mc = A[0];
    for (int k = 1; k < 9; k++) if (A[k] > mc) mc = A[k];
    if (mc == 0.0) { normal[0] = normal[1] = normal[2] = 0; return; }
    for (int k = 0; k < 9; k++) A[k] /= mc;
    double norm = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (norm > 0) {
        double q = (A[0] + A[4] + A[8]) / 3.0;
        double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
        double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
        double c00 = b11 * b22 - A[5] * A[5];
        double c01 = A[1] * b22 - A[5] * A[2];
        double c02 = A[1] * A[5] - b11 * A[2];
        double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double hd = det * 0.5; if (hd < -1) hd = -1; if (hd > 1) hd = 1;
        double angle = acos(hd) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        double beta2 = cos(angle) * 2.0, beta0 = cos(angle + two_thirds_pi) * 2.0, beta1 = -(beta0 + beta2);
        double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
        double e0[3], e1[3], e2[3];
        if (hd >= 0) {
            eigenvector0(A, ev2, e2);
            if (ev2 < ev0 && ev2 < ev1) { memcpy(normal, e2, sizeof e2); return; }
            eigenvector1(A, e2, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) { memcpy(normal, e1, sizeof e1); return; }
            cross3(e1, e2, e0);
            memcpy(normal, e0, sizeof e0); return;
        } else {
            eigenvector0(A, ev0, e0);
            if (ev0 < ev1 && ev0 < ev2) { memcpy(normal, e0, sizeof e0); return; }
            eigenvector1(A, e0, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) { memcpy(normal, e1, sizeof e1); return; }
            cross3(e0, e1, e2);
            memcpy(normal, e2, sizeof e2); return;
        }
    } else {
        if (A[0] < A[4] && A[0] < A[8]) { normal[0] = 1; normal[1] = 0; normal[2] = 0; }
        else if (A[4] < A[0] && A[4] < A[8]) { normal[0] = 0; normal[1] = 1; normal[2] = 0; }
        else { normal[0] = 0; normal[1] = 0; normal[2] = 1; }
    }
}

/* ------------------------------------------------------------ normals (A.4) */
int orc_estimate_normals(const double *xyz, int64_t n, int mode, int knn, double radius, const double *prior,
                         const double *cov9_in, double *normals) {
    if (n < 0) return ORC_EINVAL;
    double *cov = NULL;
    if (!cov9_in) {
        cov = (double *)malloc(sizeof(double) * 9 * (size_t)(n > 0 ? n : 1));
        int rc = orc_estimate_covariances(xyz, n, mode, knn, radius, cov);
        if (rc != ORC_OK) { free(cov); return rc; }
        cov9_in = cov;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double nv[3];
        orc_fast_eigen3x3(cov9_in + i * 9, nv);
        double nn = sqrt(dot3(nv, nv));
        if (nn == 0.0 || !(nn == nn)) {
            if (prior) { nv[0] = prior[i * 3]; nv[1] = prior[i * 3 + 1]; nv[2] = prior[i * 3 + 2]; }
            else { nv[0] = 0; nv[1] = 0; nv[2] = 1; }
        }
        if (prior && dot3(nv, prior + i * 3) < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
        normals[i * 3] = nv[0]; normals[i * 3 + 1] = nv[1]; normals[i * 3 + 2] = nv[2];
    }
    free(cov);
    return ORC_OK;
}
