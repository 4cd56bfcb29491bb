/*
 * gicp.c -- CPU ORACLE (test infrastructure).
 * Restates Open3D pipelines::registration::{RegistrationGeneralizedICP,
 * RegistrationICP, TransformationEstimationForGeneralizedICP, RobustKernel,
 * EvaluateRegistration, GetInformationMatrixFromPointClouds} as reached from
 * ALL_FUNCTIONS.py:220-226, 304-311, 327-331 and
 * 2_MGICP_refinement_in_NCLT_dataset.py:155-162.
 */
#include "oracle_internal.h"
#include <omp.h>

/* A.5.1: C = R diag(eps,1,1) R^T, R = rotation taking e1 onto the normal,
 * identity when e1.n < -0.99                                                   */
int orc_covariances_from_normals(const double *normals, int64_t n, double eps, double *cov9) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const double *x = normals + i * 3;
        double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        double c = x[0];                       /* e1 . x */
        if (!(c < -0.99)) {
            double v[3] = { 0, -x[2], x[1] };  /* e1 x x */
            double sv[9] = { 0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0 };
            double sv2[9]; m3_mul(sv, sv, sv2);
            double f = 1.0 / (1.0 + c);
            for (int k = 0; k < 9; k++) R[k] += sv[k] + sv2[k] * f;
        }
        double Cd[9] = { eps, 0, 0, 0, 1, 0, 0, 0, 1 }, RC[9];
        m3_mul(R, Cd, RC);
        m3_mul_bt(RC, R, cov9 + i * 9);
    }
    return ORC_OK;
}

static int64_t g_sum_chunk = 256;
int orc_set_sum_chunk(int chunk) { int old = (int)g_sum_chunk; if (chunk >= 1) g_sum_chunk = chunk; return old; }

static inline double loss_weight(int loss, double k, double r) {
    switch (loss) {
        case ORC_LOSS_L1: return 1.0 / fabs(r);
        case ORC_LOSS_GM: { double d = k + r * r; return k / (d * d); }
        default: return 1.0;
    }
}

/* A.6 on a fixed correspondence set; deterministic (serial order) accumulation in
 * fixed chunks so that the sums are identical for every thread count.           */
int orc_gicp_linearize(const double *src_xyz, const double *src_cov9, const double *tgt_xyz, const double *tgt_cov9,
                       const int32_t *corr, int64_t n_corr, int loss, double loss_k, double *JTJ, double *JTr, double *r2) {
    /* fixed-size chunks summed in chunk order: the result does not depend on the thread count.  The chunk size is a knob
     * (orc_set_sum_chunk, default 256) ONLY so that tests can measure how far the L1-IRLS end pose moves when nothing but the
     * float64 summation tree changes: that spread is the noise floor every device-vs-oracle L1 bound is derived from. */
    const int64_t CHUNK = g_sum_chunk;
    int64_t n_chunks = (n_corr + CHUNK - 1) / CHUNK;
    double *part = (double *)calloc((size_t)(n_chunks > 0 ? n_chunks : 1) * 43, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t ch = 0; ch < n_chunks; ch++) {
        double *Al = part + ch * 43, *bl = Al + 36, *rl = Al + 42;
        int64_t c1 = (ch + 1) * CHUNK < n_corr ? (ch + 1) * CHUNK : n_corr;
        for (int64_t c = ch * CHUNK; c < c1; c++) {
            int64_t is = corr[c * 2], it = corr[c * 2 + 1];
            const double *vs = src_xyz + is * 3, *vt = tgt_xyz + it * 3;
            const double *Cs = src_cov9 + is * 9, *Ct = tgt_cov9 + it * 9;
            double d[3] = { vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2] };
            double M[9], W[9];
            for (int k = 0; k < 9; k++) M[k] = Ct[k] + Cs[k];
            orc_sym3_inv_sqrt(M, W);
            /* J = W * [ -skew(vs) | I ] */
            double S[9] = { 0, vs[2], -vs[1], -vs[2], 0, vs[0], vs[1], -vs[0], 0 }; /* -skew(vs) */
            double WS[9]; m3_mul(W, S, WS);
            for (int row = 0; row < 3; row++) {
                double J[6] = { WS[row * 3], WS[row * 3 + 1], WS[row * 3 + 2], W[row * 3], W[row * 3 + 1], W[row * 3 + 2] };
                double r = W[row * 3] * d[0] + W[row * 3 + 1] * d[1] + W[row * 3 + 2] * d[2];
                double w = loss_weight(loss, loss_k, r);
                for (int a = 0; a < 6; a++) {
                    for (int bb = 0; bb < 6; bb++) Al[a * 6 + bb] += J[a] * w * J[bb];
                    bl[a] += J[a] * w * r;
                }
                *rl += r * r;
            }
        }
    }
    double A[36] = { 0 }, b[6] = { 0 }, rr = 0;
    for (int64_t ch = 0; ch < n_chunks; ch++) {
        const double *Al = part + ch * 43;
        for (int k = 0; k < 36; k++) A[k] += Al[k];
        for (int k = 0; k < 6; k++) b[k] += Al[36 + k];
        rr += Al[42];
    }
    free(part);
    memcpy(JTJ, A, sizeof A); memcpy(JTr, b, sizeof b); if (r2) *r2 = rr;
    return ORC_OK;
}

int orc_solve_update(const double *JTJ, const double *JTr, double *T16) {
    double nb[6], x[6];
    for (int k = 0; k < 6; k++) nb[k] = -JTr[k];
    int rc = orc_ldlt6_solve(JTJ, nb, x);
    if (rc != ORC_OK) { m4_identity(T16); return rc; }
    orc_vec6_to_T(x, T16);
    return ORC_OK;
}

static void search_corr(const orc_kdtree *tree, const double *src_xyz, int64_t ns, double max_dist, int32_t *corr,
                        int64_t *n_corr, double *fitness, double *rmse) {
    double r2 = max_dist * max_dist;
    int32_t *match = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ns > 0 ? ns : 1));
    double *md2 = (double *)malloc(sizeof(double) * (size_t)(ns > 0 ? ns : 1));
    size_t sb = orc_kdtree_scratch_bytes(tree, 1);
#pragma omp parallel
    {
        void *scratch = malloc(sb);
#pragma omp for schedule(dynamic, 512)
        for (int64_t i = 0; i < ns; i++) {
            int32_t id; double d2;
            int c = orc_kdtree_knn(tree, src_xyz + i * 3, 1, r2, &id, &d2, scratch);
            match[i] = c > 0 ? id : -1; md2[i] = c > 0 ? d2 : 0.0;
        }
        free(scratch);
    }
    int64_t m = 0; double e2 = 0;
    for (int64_t i = 0; i < ns; i++)
        if (match[i] >= 0) { if (corr) { corr[m * 2] = (int32_t)i; corr[m * 2 + 1] = match[i]; } e2 += md2[i]; m++; }
    *n_corr = m;
    if (m == 0) { *fitness = 0; *rmse = 0; }
    else { *fitness = (double)m / (double)ns; *rmse = sqrt(e2 / (double)m); }
    free(match); free(md2);
}

int orc_find_correspondences(const double *src_xyz, int64_t ns, const double *tgt_xyz, int64_t nt, double max_dist,
                             int32_t *corr, int64_t *n_corr, double *fitness, double *rmse) {
    if (max_dist <= 0) return ORC_EINVAL;
    orc_kdtree *t = orc_kdtree_build(tgt_xyz, nt, 3);
    search_corr(t, src_xyz, ns, max_dist, corr, n_corr, fitness, rmse);
    orc_kdtree_free(t);
    return ORC_OK;
}

static void transform_cloud(const double *T, double *xyz, double *cov9, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double *p = xyz + i * 3;
        double x = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
        double y = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
        double z = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
        p[0] = x; p[1] = y; p[2] = z;
        if (cov9) {
            double R[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] }, RC[9];
            m3_mul(R, cov9 + i * 9, RC);
            m3_mul_bt(RC, R, cov9 + i * 9);
        }
    }
}

static int is_identity4(const double *T) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (T[i * 4 + j] != (i == j ? 1.0 : 0.0)) return 0;
    return 1;
}

int orc_registration_gicp(const double *src_xyz, const double *src_nrm, const double *src_cov9, int64_t ns,
                          const double *tgt_xyz, const double *tgt_nrm, const double *tgt_cov9, int64_t nt,
                          double max_dist, const double *T0, int loss, double loss_k, double eps, double rel_fitness,
                          double rel_rmse, int max_it, orc_result *out, int32_t *corr_out, double *trace) {
    if (max_dist <= 0.0) return ORC_EINVAL;
    if ((!src_nrm && !src_cov9) || (!tgt_nrm && !tgt_cov9)) return ORC_EINVAL;
    size_t sn = (size_t)(ns > 0 ? ns : 1), tn = (size_t)(nt > 0 ? nt : 1);
    double *P = (double *)malloc(sizeof(double) * 3 * sn);
    double *Cs = (double *)malloc(sizeof(double) * 9 * sn);
    double *Ct = (double *)malloc(sizeof(double) * 9 * tn);
    int32_t *corr = (int32_t *)malloc(sizeof(int32_t) * 2 * sn);
    memcpy(P, src_xyz, sizeof(double) * 3 * (size_t)ns);
    if (src_cov9) memcpy(Cs, src_cov9, sizeof(double) * 9 * (size_t)ns); else orc_covariances_from_normals(src_nrm, ns, eps, Cs);
    if (tgt_cov9) memcpy(Ct, tgt_cov9, sizeof(double) * 9 * (size_t)nt); else orc_covariances_from_normals(tgt_nrm, nt, eps, Ct);

    double T[16]; memcpy(T, T0, sizeof T);
    orc_kdtree *tree = orc_kdtree_build(tgt_xyz, nt, 3);
    if (!is_identity4(T)) transform_cloud(T, P, Cs, ns);
    int64_t nc; double fit, rmse;
    search_corr(tree, P, ns, max_dist, corr, &nc, &fit, &rmse);
    if (trace) { trace[0] = fit; trace[1] = rmse; }
    int it = 0, converged = 0, rc_all = ORC_OK;
    for (it = 0; it < max_it; it++) {
        double U[16];
        if (nc == 0) m4_identity(U);
        else {
            double JTJ[36], JTr[6], r2;
            orc_gicp_linearize(P, Cs, tgt_xyz, Ct, corr, nc, loss, loss_k, JTJ, JTr, &r2);
            orc_solve_update(JTJ, JTr, U);   /* identity on failure, as Open3D */
        }
        m4_mul(U, T, T);
        transform_cloud(U, P, Cs, ns);
        double bfit = fit, brmse = rmse;
        search_corr(tree, P, ns, max_dist, corr, &nc, &fit, &rmse);
        if (trace) { trace[(it + 1) * 2] = fit; trace[(it + 1) * 2 + 1] = rmse; }
        if (fabs(bfit - fit) < rel_fitness && fabs(brmse - rmse) < rel_rmse) { converged = 1; it++; break; }
    }
    for (int k = 0; k < 16; k++) if (!isfinite(T[k])) rc_all = ORC_ENUMERIC;
    memcpy(out->T, T, sizeof T);
    out->fitness = fit; out->inlier_rmse = rmse; out->n_corr = nc; out->iterations = it; out->converged = converged;
    if (corr_out) memcpy(corr_out, corr, sizeof(int32_t) * 2 * (size_t)nc);
    orc_kdtree_free(tree);
    free(P); free(Cs); free(Ct); free(corr);
    return rc_all;
}

/* one cloud of one scale: voxel -> SOR -> normals. Returns malloc'ed xyz/normals */
static int prep_scale(const double *xyz, const double *nrm_in, int64_t n, double voxel, int sor_k, double sor_std,
                      int normal_k, double **oxyz, double **onrm, int64_t *n_vox, int64_t *n_clean) {
    size_t nn = (size_t)(n > 0 ? n : 1);
    double *vx = (double *)malloc(sizeof(double) * 3 * nn), *vn = nrm_in ? (double *)malloc(sizeof(double) * 3 * nn) : NULL;
    int64_t m = 0;
    int rc = orc_voxel_down_sample(xyz, n, voxel, vx, &m, nrm_in, vn);
    if (rc != ORC_OK) { free(vx); free(vn); return rc; }
    uint8_t *keep = (uint8_t *)malloc((size_t)(m > 0 ? m : 1));
    rc = orc_remove_statistical_outlier(vx, m, sor_k, sor_std, keep, NULL, NULL, NULL);
    if (rc != ORC_OK) { free(vx); free(vn); free(keep); return rc; }
    int64_t c = 0;
    for (int64_t i = 0; i < m; i++) if (keep[i]) {
        memmove(vx + c * 3, vx + i * 3, 3 * sizeof(double));
        if (vn) memmove(vn + c * 3, vn + i * 3, 3 * sizeof(double));
        c++;
    }
    free(keep);
    double *nr = (double *)malloc(sizeof(double) * 3 * (size_t)(c > 0 ? c : 1));
    rc = orc_estimate_normals(vx, c, ORC_SEARCH_KNN, normal_k, 0.0, vn, NULL, nr);
    free(vn);
    if (rc != ORC_OK) { free(vx); free(nr); return rc; }
    *oxyz = vx; *onrm = nr; *n_vox = m; *n_clean = c;
    return ORC_OK;
}

int orc_multiscale_gicp(const double *src_xyz, const double *src_nrm, int64_t ns, const double *tgt_xyz,
                        const double *tgt_nrm, int64_t nt, const double *voxels, const double *dists, int n_scales,
                        int sor_k, double sor_std, int normal_k, const double *T0, int loss, double loss_k, double eps,
                        double rel_fitness, double rel_rmse, int max_it, orc_scale_stats *stats, int32_t *corr_last) {
    double T[16]; memcpy(T, T0, sizeof T);
    for (int s = 0; s < n_scales; s++) {
        double *sx = NULL, *sn = NULL, *tx = NULL, *tn = NULL;
        int rc = prep_scale(src_xyz, src_nrm, ns, voxels[s], sor_k, sor_std, normal_k, &sx, &sn, &stats[s].n_voxel[0], &stats[s].n_clean[0]);
        if (rc != ORC_OK) return rc;
        rc = prep_scale(tgt_xyz, tgt_nrm, nt, voxels[s], sor_k, sor_std, normal_k, &tx, &tn, &stats[s].n_voxel[1], &stats[s].n_clean[1]);
        if (rc != ORC_OK) { free(sx); free(sn); return rc; }
        rc = orc_registration_gicp(sx, sn, NULL, stats[s].n_clean[0], tx, tn, NULL, stats[s].n_clean[1], dists[s], T, loss,
                                   loss_k, eps, rel_fitness, rel_rmse, max_it, &stats[s].icp,
                                   (s == n_scales - 1) ? corr_last : NULL, NULL);
        free(sx); free(sn); free(tx); free(tn);
        if (rc != ORC_OK) return rc;
        memcpy(T, stats[s].icp.T, sizeof T);
    }
    return ORC_OK;
}

/* EvaluateRegistration: transform source by T, 1-NN within max_dist           */
int orc_evaluate_registration(const double *src_xyz, int64_t ns, const double *tgt_xyz, int64_t nt, double max_dist,
                              const double *T, orc_result *out, int32_t *corr) {
    if (max_dist <= 0) return ORC_EINVAL;
    double *P = (double *)malloc(sizeof(double) * 3 * (size_t)(ns > 0 ? ns : 1));
    memcpy(P, src_xyz, sizeof(double) * 3 * (size_t)ns);
    if (!is_identity4(T)) transform_cloud(T, P, NULL, ns);
    orc_kdtree *tree = orc_kdtree_build(tgt_xyz, nt, 3);
    search_corr(tree, P, ns, max_dist, corr, &out->n_corr, &out->fitness, &out->inlier_rmse);
    memcpy(out->T, T, 16 * sizeof(double)); out->iterations = 0; out->converged = 0;
    orc_kdtree_free(tree); free(P);
    return ORC_OK;
}

/* GetInformationMatrixFromPointClouds: correspondences as EvaluateRegistration,
 * then sum over matched TARGET points t of G^T G with rows
 * [0, z, -y, 1,0,0], [-z, 0, x, 0,1,0], [y, -x, 0, 0,0,1]       */
int orc_information_matrix(const double *src_xyz, int64_t ns, const double *tgt_xyz, int64_t nt, double max_dist,
                           const double *T, double *info36) {
    orc_result r;
    int32_t *corr = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(ns > 0 ? ns : 1));
    int rc = orc_evaluate_registration(src_xyz, ns, tgt_xyz, nt, max_dist, T, &r, corr);
    if (rc != ORC_OK) { free(corr); return rc; }
    double G[36] = { 0 };
    for (int64_t c = 0; c < r.n_corr; c++) {
        const double *t = tgt_xyz + (int64_t)corr[c * 2 + 1] * 3;
        double x = t[0], y = t[1], z = t[2];
        double rows[3][6] = { { 0, z, -y, 1, 0, 0 }, { -z, 0, x, 0, 1, 0 }, { y, -x, 0, 0, 0, 1 } };
        for (int rr = 0; rr < 3; rr++)
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) G[a * 6 + b] += rows[rr][a] * rows[rr][b];
    }
    memcpy(info36, G, sizeof G);
    free(corr);
    return ORC_OK;
}
