/*
 * fpfh.c -- CPU ORACLE (test infrastructure).
 * Restates Open3D pipelines::registration::ComputeFPFHFeature (Feature.cpp: ComputePairFeatures,
 * ComputeSPFHFeature, ComputeFPFHFeature) as called at ALL_FUNCTIONS.py:186-187 and
 * 1_FGR_pairwise_registration_in_NCLT_dataset.py:49-50.  Output: n x 33, row per point.
 */
#include "oracle_internal.h"
#include <omp.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static void pair_features(const double *p1, const double *n1, const double *p2, const double *n2, double *f) {
    double dp[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] };
    f[0] = f[1] = f[2] = 0; f[3] = sqrt(dot3(dp, dp));
    if (f[3] == 0.0) { f[3] = 0; return; }
    double a[3] = { n1[0], n1[1], n1[2] }, b[3] = { n2[0], n2[1], n2[2] };
    double angle1 = dot3(a, dp) / f[3], angle2 = dot3(b, dp) / f[3];
    double f2;
    if (acos(fabs(angle1)) > acos(fabs(angle2))) {
        a[0] = n2[0]; a[1] = n2[1]; a[2] = n2[2]; b[0] = n1[0]; b[1] = n1[1]; b[2] = n1[2];
        dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2];
        f2 = -angle2;
    } else f2 = angle1;
    double v[3]; cross3(dp, a, v);
    double vn = sqrt(dot3(v, v));
    if (vn == 0.0) { f[0] = f[1] = f[2] = f[3] = 0; return; }
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    double w[3]; cross3(a, v, w);
    f[2] = f2;
    f[1] = dot3(v, b);
    f[0] = atan2(dot3(w, b), dot3(a, b));
}

static inline int bin11(double x) {
    int h = (int)floor(x);
    if (h < 0) h = 0; if (h >= 11) h = 10;
    return h;
}

int orc_compute_fpfh(const double *xyz, const double *nrm, int64_t n, int mode, int knn, double radius, double *feat33) {
    if (n < 0) return ORC_EINVAL;
    if (n == 0) return ORC_OK;
    double *spfh = (double *)calloc((size_t)n * 33, sizeof(double));
    memset(feat33, 0, sizeof(double) * 33 * (size_t)n);
    orc_kdtree *t = orc_kdtree_build(xyz, n, 3);
    size_t sb = orc_kdtree_scratch_bytes(t, knn > 0 ? knn : 1);
#pragma omp parallel
    {
        void *scratch = malloc(sb);
        int32_t *ti = NULL; double *td = NULL; int64_t cap = 0;
#pragma omp for schedule(dynamic, 128)
        for (int64_t i = 0; i < n; i++) {
            int64_t c = orc_neighbourhood(t, xyz + i * 3, mode, knn, radius, &ti, &td, &cap, scratch);
            if (c > 1) {
                double inc = 100.0 / (double)(c - 1);
                double *h = spfh + i * 33;
                for (int64_t k = 1; k < c; k++) {
                    double pf[4];
                    pair_features(xyz + i * 3, nrm + i * 3, xyz + (int64_t)ti[k] * 3, nrm + (int64_t)ti[k] * 3, pf);
                    h[bin11(11 * (pf[0] + M_PI) / (2.0 * M_PI))] += inc;
                    h[bin11(11 * (pf[1] + 1.0) * 0.5) + 11] += inc;
                    h[bin11(11 * (pf[2] + 1.0) * 0.5) + 22] += inc;
                }
            }
        }
#pragma omp for schedule(dynamic, 128)
        for (int64_t i = 0; i < n; i++) {
            int64_t c = orc_neighbourhood(t, xyz + i * 3, mode, knn, radius, &ti, &td, &cap, scratch);
            if (c > 1) {
                double sum[3] = { 0, 0, 0 };
                double *F = feat33 + i * 33;
                for (int64_t k = 1; k < c; k++) {
                    double dist = td[k];
                    if (dist == 0.0) continue;
                    const double *s = spfh + (int64_t)ti[k] * 33;
                    for (int j = 0; j < 33; j++) { double val = s[j] / dist; sum[j / 11] += val; F[j] += val; }
                }
                for (int j = 0; j < 3; j++) if (sum[j] != 0.0) sum[j] = 100.0 / sum[j];
                for (int j = 0; j < 33; j++) { F[j] *= sum[j / 11]; F[j] += spfh[i * 33 + j]; }
            }
        }
        free(scratch); free(ti); free(td);
    }
    orc_kdtree_free(t);
    free(spfh);
    return ORC_OK;
}
