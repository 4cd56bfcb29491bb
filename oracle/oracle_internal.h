/* oracle_internal.h -- CPU ORACLE (test infrastructure): shared internals. */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "pcr_oracle.h"

typedef struct orc_kdtree orc_kdtree;
orc_kdtree *orc_kdtree_build(const double *pts, int64_t n, int dim);
void orc_kdtree_free(orc_kdtree *t);
size_t orc_kdtree_scratch_bytes(const orc_kdtree *t, int k);
int orc_kdtree_knn(const orc_kdtree *t, const double *q, int k, double radius2,
                   int32_t *idx_out, double *d2_out, void *scratch);
int64_t orc_kdtree_radius(const orc_kdtree *t, const double *q, double r2,
                          int32_t **idx, double **d2o, int64_t *cap);

/* neighbourhood by KDTreeSearchParam semantics; returns count, sorted ascending.
 * buffers idx/d2 are grown as needed (radius mode).                            */
int64_t orc_neighbourhood(const orc_kdtree *t, const double *q, int mode, int knn,
                          double radius, int32_t **idx, double **d2, int64_t *cap,
                          void *scratch);

/* ---- small dense algebra (row-major) ---- */
static inline void m3_mul(const double *a, const double *b, double *c) {
    double r[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
    memcpy(c, r, sizeof r);
}
static inline void m3_mul_bt(const double *a, const double *b, double *c) { /* a * b^T */
    double r[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r[i * 3 + j] = a[i * 3 + 0] * b[j * 3 + 0] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
    memcpy(c, r, sizeof r);
}
static inline void m4_mul(const double *a, const double *b, double *c) {
    double r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
            r[i * 4 + j] = s;
        }
    memcpy(c, r, sizeof r);
}
static inline void m4_identity(double *a) { memset(a, 0, 16 * sizeof(double)); a[0] = a[5] = a[10] = a[15] = 1; }
static inline void cross3(const double *a, const double *b, double *c) {
    double r0 = a[1] * b[2] - a[2] * b[1], r1 = a[2] * b[0] - a[0] * b[2], r2 = a[0] * b[1] - a[1] * b[0];
    c[0] = r0; c[1] = r1; c[2] = r2;
}
static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* symmetric 3x3 eigen-decomposition by cyclic Jacobi: A = V diag(w) V^T */
void orc_sym3_eig(const double A[9], double w[3], double V[9]);
/* W = (M^-1)^(1/2), symmetric principal root */
void orc_sym3_inv_sqrt(const double M[9], double W[9]);
/* 6x6 symmetric solve by pivoted LDL^T; returns 0 on success */
int orc_ldlt6_solve(const double *A36, const double *b6, double *x6);
/* Rz(g)*Ry(b)*Rx(a) | t from x = (a,b,g,tx,ty,tz) */
void orc_vec6_to_T(const double *x6, double *T16);
/* general 4x4 rigid inverse */
void orc_rigid_inverse(const double *T, double *Ti);

#endif
