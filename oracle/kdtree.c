/*
 * kdtree.c -- CPU ORACLE (test infrastructure).  Exact k-d tree in D dimensions.
 *
 * Stands in for Open3D's geometry::KDTreeFlann (nanoflann) as used implicitly by
 * every KDTreeSearchParam* call site of the reference (ALL_FUNCTIONS.py:181,185,
 * 213,301): SearchKNN = exact k nearest sorted ascending by squared
 * distance, query point included when it is a member; SearchHybrid = k-NN then keep
 * d^2 < r^2; SearchRadius = all with d^2 < r^2.
 */
#include "oracle_internal.h"

typedef struct {
    int32_t left, right;   /* children; left < 0 => leaf          */
    int32_t begin, end;    /* leaf: range in perm                 */
    int32_t split_dim;
    double split_lo, split_hi; /* nanoflann-style split plane interval */
} kd_node;

struct orc_kdtree {
    const double *pts;
    int64_t n;
    int dim;
    int32_t *perm;
    kd_node *nodes;
    int32_t n_nodes, cap_nodes;
    double *bb_lo, *bb_hi;  /* root bounding box */
};

#define LEAF_SIZE 12

static int32_t new_node(orc_kdtree *t) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes = (kd_node *)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
    }
    return t->n_nodes++;
}

static void nth_element_dim(const double *pts, int dim, int d, int32_t *a, int64_t n, int64_t k) {
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        double pivot = pts[(int64_t)a[(lo + hi) / 2] * dim + d];
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (pts[(int64_t)a[i] * dim + d] < pivot) i++;
            while (pts[(int64_t)a[j] * dim + d] > pivot) j--;
            if (i <= j) { int32_t tmp = a[i]; a[i] = a[j]; a[j] = tmp; i++; j--; }
        }
        if (k <= j) hi = j; else if (k >= i) lo = i; else return;
    }
}

static int32_t build_rec(orc_kdtree *t, int32_t begin, int32_t end, double *lo, double *hi) {
    int32_t id = new_node(t);
    int dim = t->dim;
    if (end - begin <= LEAF_SIZE) {
        kd_node nd; nd.left = nd.right = -1; nd.begin = begin; nd.end = end;
        nd.split_dim = 0; nd.split_lo = nd.split_hi = 0;
        t->nodes[id] = nd;
        /* tighten bbox to leaf content */
        for (int d = 0; d < dim; d++) { lo[d] = INFINITY; hi[d] = -INFINITY; }
        for (int32_t i = begin; i < end; i++) {
            const double *p = t->pts + (int64_t)t->perm[i] * dim;
            for (int d = 0; d < dim; d++) { if (p[d] < lo[d]) lo[d] = p[d]; if (p[d] > hi[d]) hi[d] = p[d]; }
        }
        return id;
    }
    /* split on the dimension of largest extent of the actual points */
    int sd = 0; double best = -1;
    for (int d = 0; d < dim; d++) {
        double mn = INFINITY, mx = -INFINITY;
        for (int32_t i = begin; i < end; i++) {
            double v = t->pts[(int64_t)t->perm[i] * dim + d];
            if (v < mn) mn = v; if (v > mx) mx = v;
        }
        if (mx - mn > best) { best = mx - mn; sd = d; }
    }
    int32_t mid = begin + (end - begin) / 2;
    nth_element_dim(t->pts, dim, sd, t->perm + begin, end - begin, mid - begin);
    double *llo = (double *)malloc(sizeof(double) * 4 * (size_t)dim);
    double *lhi = llo + dim, *rlo = lhi + dim, *rhi = rlo + dim;
    int32_t l = build_rec(t, begin, mid, llo, lhi);
    int32_t r = build_rec(t, mid, end, rlo, rhi);
    kd_node nd; nd.left = l; nd.right = r; nd.begin = begin; nd.end = end; nd.split_dim = sd;
    nd.split_lo = lhi[sd]; nd.split_hi = rlo[sd];
    t->nodes[id] = nd;
    for (int d = 0; d < dim; d++) { lo[d] = llo[d] < rlo[d] ? llo[d] : rlo[d]; hi[d] = lhi[d] > rhi[d] ? lhi[d] : rhi[d]; }
    free(llo);
    return id;
}

orc_kdtree *orc_kdtree_build(const double *pts, int64_t n, int dim) {
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(orc_kdtree));
    t->pts = pts; t->n = n; t->dim = dim;
    t->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) t->perm[i] = (int32_t)i;
    t->bb_lo = (double *)malloc(sizeof(double) * 2 * (size_t)dim);
    t->bb_hi = t->bb_lo + dim;
    if (n > 0) build_rec(t, 0, (int32_t)n, t->bb_lo, t->bb_hi);
    return t;
}

void orc_kdtree_free(orc_kdtree *t) {
    if (!t) return;
    free(t->perm); free(t->nodes); free(t->bb_lo); free(t);
}

/* ---- bounded max-heap of (d2, idx) ---- */
typedef struct { double d2; int32_t idx; } heap_item;
typedef struct { heap_item *a; int k, size; double radius2; } knn_heap;

static inline int item_less(heap_item x, heap_item y) { /* total order: d2 then idx */
    return x.d2 < y.d2 || (x.d2 == y.d2 && x.idx < y.idx);
}
static inline double heap_worst(const knn_heap *h) { return h->size < h->k ? h->radius2 : h->a[0].d2; }

static inline void heap_push(knn_heap *h, double d2, int32_t idx) {
    heap_item it = { d2, idx };
    if (h->size < h->k) {
        int i = h->size++;
        h->a[i] = it;
        while (i > 0) { int p = (i - 1) / 2; if (item_less(h->a[p], h->a[i])) { heap_item t = h->a[p]; h->a[p] = h->a[i]; h->a[i] = t; i = p; } else break; }
    } else if (item_less(it, h->a[0])) {
        h->a[0] = it; int i = 0;
        for (;;) {
            int l = 2 * i + 1, r = l + 1, m = i;
            if (l < h->size && item_less(h->a[m], h->a[l])) m = l;
            if (r < h->size && item_less(h->a[m], h->a[r])) m = r;
            if (m == i) break;
            heap_item t = h->a[m]; h->a[m] = h->a[i]; h->a[i] = t; i = m;
        }
    }
}

static void search_rec(const orc_kdtree *t, int32_t id, const double *q, knn_heap *h, double mindist2, double *off) {
    const kd_node *nd = &t->nodes[id];
    int dim = t->dim;
    if (nd->left < 0) {
        for (int32_t i = nd->begin; i < nd->end; i++) {
            int32_t pi = t->perm[i];
            const double *p = t->pts + (int64_t)pi * dim;
            double d2 = 0;
            for (int d = 0; d < dim; d++) { double e = q[d] - p[d]; d2 += e * e; }
            if (d2 < h->radius2) heap_push(h, d2, pi);
        }
        return;
    }
    int sd = nd->split_dim;
    double v = q[sd];
    double d1 = v - nd->split_lo, d2_ = v - nd->split_hi;
    int32_t near_, far_; double cut;
    if (d1 + d2_ < 0) { near_ = nd->left; far_ = nd->right; cut = d2_ * d2_; }
    else { near_ = nd->right; far_ = nd->left; cut = d1 * d1; }
    search_rec(t, near_, q, h, mindist2, off);
    double saved = off[sd];
    double md = mindist2 + cut - saved;
    off[sd] = cut;
    if (md <= heap_worst(h)) search_rec(t, far_, q, h, md, off);
    off[sd] = saved;
}

static int cmp_item(const void *a, const void *b) {
    const heap_item *x = (const heap_item *)a, *y = (const heap_item *)b;
    if (x->d2 < y->d2) return -1; if (x->d2 > y->d2) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* k nearest with d2 < radius2 (radius2 = INFINITY for plain kNN). Returns count,
 * results sorted ascending. scratch must hold k heap_items + dim doubles.        */
int orc_kdtree_knn(const orc_kdtree *t, const double *q, int k, double radius2,
                   int32_t *idx_out, double *d2_out, void *scratch) {
    if (t->n == 0 || k <= 0) return 0;
    knn_heap h; h.a = (heap_item *)scratch; h.k = k; h.size = 0; h.radius2 = radius2;
    double *off = (double *)((char *)scratch + sizeof(heap_item) * (size_t)k);
    int dim = t->dim; double mind = 0;
    for (int d = 0; d < dim; d++) {
        off[d] = 0;
        if (q[d] < t->bb_lo[d]) { double e = q[d] - t->bb_lo[d]; off[d] = e * e; }
        else if (q[d] > t->bb_hi[d]) { double e = q[d] - t->bb_hi[d]; off[d] = e * e; }
        mind += off[d];
    }
    search_rec(t, 0, q, &h, mind, off);
    qsort(h.a, (size_t)h.size, sizeof(heap_item), cmp_item);
    for (int i = 0; i < h.size; i++) { idx_out[i] = h.a[i].idx; d2_out[i] = h.a[i].d2; }
    return h.size;
}

size_t orc_kdtree_scratch_bytes(const orc_kdtree *t, int k) {
    return sizeof(heap_item) * (size_t)(k > 0 ? k : 1) + sizeof(double) * (size_t)t->dim;
}

/* ---- radius search: all points with d2 < r2, unsorted, growing buffer ---- */
static void radius_rec(const orc_kdtree *t, int32_t id, const double *q, double r2, double mindist2, double *off,
                       int32_t **idx, double **d2o, int64_t *cnt, int64_t *cap) {
    const kd_node *nd = &t->nodes[id];
    int dim = t->dim;
    if (nd->left < 0) {
        for (int32_t i = nd->begin; i < nd->end; i++) {
            int32_t pi = t->perm[i];
            const double *p = t->pts + (int64_t)pi * dim;
            double d2 = 0;
            for (int d = 0; d < dim; d++) { double e = q[d] - p[d]; d2 += e * e; }
            if (d2 < r2) {
                if (*cnt == *cap) {
                    *cap = *cap ? *cap * 2 : 64;
                    *idx = (int32_t *)realloc(*idx, sizeof(int32_t) * (size_t)*cap);
                    *d2o = (double *)realloc(*d2o, sizeof(double) * (size_t)*cap);
                }
                (*idx)[*cnt] = pi; (*d2o)[*cnt] = d2; (*cnt)++;
            }
        }
        return;
    }
    int sd = nd->split_dim;
    double v = q[sd];
    double d1 = v - nd->split_lo, d2_ = v - nd->split_hi;
    int32_t near_, far_; double cut;
    if (d1 + d2_ < 0) { near_ = nd->left; far_ = nd->right; cut = d2_ * d2_; }
    else { near_ = nd->right; far_ = nd->left; cut = d1 * d1; }
    radius_rec(t, near_, q, r2, mindist2, off, idx, d2o, cnt, cap);
    double saved = off[sd];
    double md = mindist2 + cut - saved;
    off[sd] = cut;
    if (md <= r2) radius_rec(t, far_, q, r2, md, off, idx, d2o, cnt, cap);
    off[sd] = saved;
}

int64_t orc_kdtree_radius(const orc_kdtree *t, const double *q, double r2, int32_t **idx, double **d2o, int64_t *cap) {
    int64_t cnt = 0;
    if (t->n == 0) return 0;
    int dim = t->dim; double mind = 0;
    double *off = (double *)malloc(sizeof(double) * (size_t)dim);
    for (int d = 0; d < dim; d++) {
        off[d] = 0;
        if (q[d] < t->bb_lo[d]) { double e = q[d] - t->bb_lo[d]; off[d] = e * e; }
        else if (q[d] > t->bb_hi[d]) { double e = q[d] - t->bb_hi[d]; off[d] = e * e; }
        mind += off[d];
    }
    radius_rec(t, 0, q, r2, mind, off, idx, d2o, &cnt, cap);
    free(off);
    /* sort ascending by (d2, idx) */
    heap_item *tmp = (heap_item *)malloc(sizeof(heap_item) * (size_t)(cnt > 0 ? cnt : 1));
    for (int64_t i = 0; i < cnt; i++) { tmp[i].d2 = (*d2o)[i]; tmp[i].idx = (*idx)[i]; }
    qsort(tmp, (size_t)cnt, sizeof(heap_item), cmp_item);
    for (int64_t i = 0; i < cnt; i++) { (*d2o)[i] = tmp[i].d2; (*idx)[i] = tmp[i].idx; }
    free(tmp);
    return cnt;
}

/* ---- public batched API ---- */
int orc_knn(const double *pts, int64_t n, int dim, const double *queries, int64_t nq, int k, double radius,
            int64_t *idx, double *d2, int32_t *counts) {
    if (n < 0 || nq < 0 || dim < 1 || k < 1) return ORC_EINVAL;
    orc_kdtree *t = orc_kdtree_build(pts, n, dim);
    double r2 = radius > 0 ? radius * radius : INFINITY;
    size_t sb = orc_kdtree_scratch_bytes(t, k);
#pragma omp parallel
    {
        void *scratch = malloc(sb);
        int32_t *ti = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
        double *td = (double *)malloc(sizeof(double) * (size_t)k);
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < nq; i++) {
            int c = orc_kdtree_knn(t, queries + i * dim, k, r2, ti, td, scratch);
            for (int j = 0; j < k; j++) {
                idx[i * k + j] = j < c ? ti[j] : -1;
                d2[i * k + j] = j < c ? td[j] : INFINITY;
            }
            if (counts) counts[i] = c;
        }
        free(scratch); free(ti); free(td);
    }
    orc_kdtree_free(t);
    return ORC_OK;
}
