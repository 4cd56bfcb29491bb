/*
 * fgr.c -- CPU ORACLE (test infrastructure).
 * Restates Open3D pipelines::registration::FastGlobalRegistrationBasedOnFeatureMatching
 * (FastGlobalRegistration.cpp: NormalizePointCloud, AdvancedMatching, OptimizePairwiseRegistration,
 * GetTransformationOriginalScale) as called at ALL_FUNCTIONS.py:198-202 and
 * 1_FGR_pairwise_registration_in_NCLT_dataset.py:61-65.
 *
 * Open3D seeds the tuple sampler non-deterministically; here the sampler is a counter-based generator
 * (splitmix64 of seed + 3*trial + k) so that the HIP path can draw the SAME triples in parallel.
 */
#include "oracle_internal.h"
#include <omp.h>

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static void nn_feature(const double *fa, int64_t na, const double *fq, int64_t nq, int32_t *out) {
    /* for every query row of fq its exact nearest row of fa in 33-D (ties -> smallest index) */
    orc_kdtree *t = orc_kdtree_build(fa, na, 33);
    size_t sb = orc_kdtree_scratch_bytes(t, 1);
#pragma omp parallel
    {
        void *scratch = malloc(sb);
#pragma omp for schedule(dynamic, 64)
        for (int64_t q = 0; q < nq; q++) {
            int32_t id = -1; double d2;
            int c = orc_kdtree_knn(t, fq + q * 33, 1, INFINITY, &id, &d2, scratch);
            out[q] = c > 0 ? id : -1;
        }
        free(scratch);
    }
    orc_kdtree_free(t);
}

int orc_registration_fgr(const double *src_xyz, const double *src_feat, int64_t ns, const double *tgt_xyz,
                         const double *tgt_feat, int64_t nt, const orc_fgr_option *opt, orc_result *out,
                         int64_t *n_cross_out, int64_t *n_tuple_out) {
    if (ns < 0 || nt < 0) return ORC_EINVAL;
    m4_identity(out->T); out->fitness = 0; out->inlier_rmse = 0; out->n_corr = 0; out->iterations = 0; out->converged = 0;
    if (n_cross_out) *n_cross_out = 0; if (n_tuple_out) *n_tuple_out = 0;
    const double *xyz[2] = { src_xyz, tgt_xyz }, *feat[2] = { src_feat, tgt_feat };
    int64_t np[2] = { ns, nt };
    /* ---- NormalizePointCloud */
    double *P[2]; double mean[2][3]; double scale = 0;
    for (int c = 0; c < 2; c++) {
        P[c] = (double *)malloc(sizeof(double) * 3 * (size_t)(np[c] > 0 ? np[c] : 1));
        double m[3] = { 0, 0, 0 };
        for (int64_t i = 0; i < np[c]; i++) for (int d = 0; d < 3; d++) m[d] += xyz[c][i * 3 + d];
        for (int d = 0; d < 3; d++) mean[c][d] = np[c] > 0 ? m[d] / (double)np[c] : 0.0;
        double mx = 0;
        for (int64_t i = 0; i < np[c]; i++) {
            for (int d = 0; d < 3; d++) P[c][i * 3 + d] = xyz[c][i * 3 + d] - mean[c][d];
            double nn = sqrt(dot3(P[c] + i * 3, P[c] + i * 3));
            if (nn > mx) mx = nn;
        }
        if (mx > scale) scale = mx;
    }
    double scale_global, scale_start;
    if (opt->use_absolute_scale) { scale_global = 1.0; scale_start = scale; } else { scale_global = scale; scale_start = 1.0; }
    for (int c = 0; c < 2; c++) for (int64_t i = 0; i < np[c] * 3; i++) P[c][i] /= scale_global;

    /* ---- AdvancedMatching: mutual nearest neighbours in feature space.  fi = the larger cloud. */
    int fi = 0, fj = 1, swapped = 0;
    if (np[fj] > np[fi]) { fi = 1; fj = 0; swapped = 1; }
    int64_t nPti = np[fi], nPtj = np[fj];
    int32_t *j_to_i = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nPtj > 0 ? nPtj : 1));
    int32_t *i_to_j = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nPti > 0 ? nPti : 1));
    int64_t ncross = 0, ntuple = 0;
    int32_t *cross = NULL, *tuples = NULL;
    if (nPti > 0 && nPtj > 0) {
        nn_feature(feat[fi], nPti, feat[fj], nPtj, j_to_i);
        nn_feature(feat[fj], nPtj, feat[fi], nPti, i_to_j);     /* Open3D evaluates this lazily for the i's that are hit */
        /* cross check, ordered by i */
        uint8_t *hit = (uint8_t *)calloc((size_t)nPti, 1);
        for (int64_t j = 0; j < nPtj; j++) hit[j_to_i[j]] = 1;
        cross = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)nPti);
        for (int64_t i = 0; i < nPti; i++)
            if (hit[i]) { int32_t j = i_to_j[i]; if (j_to_i[j] == (int32_t)i) { cross[ncross * 2] = (int32_t)i; cross[ncross * 2 + 1] = j; ncross++; } }
        free(hit);
    }
    /* ---- tuple test */
    int64_t max_tuples = opt->maximum_tuple_count;
    if (opt->tuple_test && ncross > 0) {
        tuples = (int32_t *)malloc(sizeof(int32_t) * 2 * 3 * (size_t)(max_tuples > 0 ? max_tuples : 1));
        int64_t trials = ncross * 100, cnt = 0;
        double ts = opt->tuple_scale;
        for (int64_t t = 0; t < trials && cnt < max_tuples; t++) {
            int64_t r[3];
            for (int k = 0; k < 3; k++) r[k] = (int64_t)(splitmix64(opt->seed + 3 * (uint64_t)t + (uint64_t)k) % (uint64_t)ncross);
            const double *pi[3], *pj[3];
            for (int k = 0; k < 3; k++) { pi[k] = P[fi] + (int64_t)cross[r[k] * 2] * 3; pj[k] = P[fj] + (int64_t)cross[r[k] * 2 + 1] * 3; }
            int ok = 1;
            for (int k = 0; k < 3 && ok; k++) {
                const double *a = pi[k], *b = pi[(k + 1) % 3], *c = pj[k], *d = pj[(k + 1) % 3];
                double li = sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
                double lj = sqrt((c[0] - d[0]) * (c[0] - d[0]) + (c[1] - d[1]) * (c[1] - d[1]) + (c[2] - d[2]) * (c[2] - d[2]));
                ok = (li * ts < lj) && (lj < li / ts);
            }
            if (ok) {
                for (int k = 0; k < 3; k++) { tuples[(cnt * 3 + k) * 2] = cross[r[k] * 2]; tuples[(cnt * 3 + k) * 2 + 1] = cross[r[k] * 2 + 1]; }
                cnt++;
            }
        }
        ntuple = cnt * 3;
    } else if (ncross > 0) {
        tuples = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)ncross);
        memcpy(tuples, cross, sizeof(int32_t) * 2 * (size_t)ncross);
        ntuple = ncross;
    }
    if (n_cross_out) *n_cross_out = ncross; if (n_tuple_out) *n_tuple_out = ntuple;
    /* pairs as (index in cloud 0 = source, index in cloud 1 = target) */
    if (swapped) for (int64_t c = 0; c < ntuple; c++) { int32_t t = tuples[c * 2]; tuples[c * 2] = tuples[c * 2 + 1]; tuples[c * 2 + 1] = t; }

    /* ---- OptimizePairwiseRegistration: moves cloud 1 (target) onto cloud 0 (source) */
    double trans[16]; m4_identity(trans);
    if (ntuple >= 10) {
        double par = scale_start;
        double *Q = (double *)malloc(sizeof(double) * 3 * (size_t)nt);
        memcpy(Q, P[1], sizeof(double) * 3 * (size_t)nt);
        for (int itr = 0; itr < opt->iteration_number; itr++) {
            double JTJ[36] = { 0 }, JTr[6] = { 0 };
            for (int64_t c = 0; c < ntuple; c++) {
                const double *p = P[0] + (int64_t)tuples[c * 2] * 3, *q = Q + (int64_t)tuples[c * 2 + 1] * 3;
                double rpq[3] = { p[0] - q[0], p[1] - q[1], p[2] - q[2] };
                double temp = par / (dot3(rpq, rpq) + par), s = temp * temp;
                double J[3][6] = { { 0, -q[2], q[1], -1, 0, 0 }, { q[2], 0, -q[0], 0, -1, 0 }, { -q[1], q[0], 0, 0, 0, -1 } };
                for (int row = 0; row < 3; row++)
                    for (int a = 0; a < 6; a++) {
                        for (int b = 0; b < 6; b++) JTJ[a * 6 + b] += J[row][a] * J[row][b] * s;
                        JTr[a] += J[row][a] * rpq[row] * s;
                    }
            }
            /* SolveLinearSystemPSD(-JTJ, JTr) */
            double nA[36], x[6], D[16];
            for (int k = 0; k < 36; k++) nA[k] = -JTJ[k];
            if (orc_ldlt6_solve(nA, JTr, x) == ORC_OK) orc_vec6_to_T(x, D); else m4_identity(D);
            m4_mul(D, trans, trans);
            for (int64_t i = 0; i < nt; i++) {
                double *q = Q + i * 3;
                double X = D[0] * q[0] + D[1] * q[1] + D[2] * q[2] + D[3], Y = D[4] * q[0] + D[5] * q[1] + D[6] * q[2] + D[7],
                       Z = D[8] * q[0] + D[9] * q[1] + D[10] * q[2] + D[11];
                q[0] = X; q[1] = Y; q[2] = Z;
            }
            if (opt->decrease_mu && itr % 4 == 0 && par > opt->maximum_correspondence_distance) par /= opt->division_factor;
        }
        free(Q);
    }
    /* ---- GetTransformationOriginalScale, then invert (source -> target) */
    double To[16]; memset(To, 0, sizeof To);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) To[r * 4 + c] = trans[r * 4 + c];
        To[r * 4 + 3] = -(trans[r * 4 + 0] * mean[1][0] + trans[r * 4 + 1] * mean[1][1] + trans[r * 4 + 2] * mean[1][2]) +
                        trans[r * 4 + 3] * scale_global + mean[0][r];
    }
    To[15] = 1;
    double Ti[16]; orc_rigid_inverse(To, Ti);
    int rc = orc_evaluate_registration(src_xyz, ns, tgt_xyz, nt, opt->maximum_correspondence_distance, Ti, out, NULL);
    free(P[0]); free(P[1]); free(j_to_i); free(i_to_j); free(cross); free(tuples);
    return rc;
}
