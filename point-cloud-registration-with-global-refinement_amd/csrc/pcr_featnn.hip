// pcr_featnn.hip -- exact nearest neighbour in the 33-D FPFH feature space (K7, the one GEMM-shaped step of the path) as a
// SCREEN on the f16 matrix cores followed by an exact float64 re-check of the few survivors.
// Reference behaviour: the two KDTreeFlann searches of Open3D's FastGlobalRegistration AdvancedMatching, reached from
// ALL_FUNCTIONS.py:198-202 / 1_FGR_pairwise_registration_in_NCLT_dataset.py:61-65: for every feature row of
// one cloud the row of the other cloud with the smallest sum_k (a_k - b_k)^2 in float64, ties -> the smaller index.
//
// Why a screen.  All Ns x Nt distances in float64 (v_mfma_f64_16x16x4, 78.6 TFLOP/s peak on MI355X) took 2 x 55 ms at 200k x 200k
// points -- 80 % of registro_FGR.  The f16 matrix pipe is 32x faster but 11 bits wide.  So every centred feature x = (f - mu) * 128
// is split into two halves, hi = f16(x), lo = f16(x - hi) (22 bits together), and the screen's dot product is
//     x_q . x_b  ~=  hi_q.hi_b  +  sum over the COVERED dimensions of (lo_q hi_b + hi_q lo_b)
// -- ONE product of length K between the rows [hi | hi_c | lo_c] of the database and [hi | lo_c | hi_c] of the queries.  Rounds 2-3 covered
// all 33 dimensions (K = 99 -> 128: 4 v_mfma_f32_16x16x32_f16 per 16 x 16 block).  Round 4: K = 64 -- the 15 dimensions of the largest variance
// are covered (33 + 2 * 15 = 63; on NCLT scans they hold 93 % of the variance of the FPFH bins), the other 18 enter with their hi halves
// only: 2 MFMAs per block, half the operand registers (124 VGPRs instead of 188: two workgroups per CU instead of one), half the bytes staged
// and read from LDS.  The price is a wider bound on the uncovered part, which costs candidates, not correctness:
// with the exact float64 norms, d~(q, b) = |x_q|^2 + |x_b|^2 - 2 dot  satisfies  |d~ - d| <= E_q + E_b,
//     E_row = FN_C |x|^2 + FN_CU |x|_U^2 + FN_EABS
// (|x|_U = the norm over the uncovered dimensions; FN_C: split truncation 3 * 2^-22 |x_q||x_b| of a covered dimension, f32 accumulation and
// combination; FN_CU: a hi-only dimension misses hi_q e_b + e_q hi_b + e_q e_b with |e| <= 2^-11 |x|, i.e. (2^-10 + 2^-20) |x_q,d||x_b,d|
// in the dot product, twice that in d~, and |a||b| <= (a^2 + b^2) / 2).  So with L = d~ - E a lower and U = min_b (d~ + E) an upper
// bound of the true minimum, every row with L(b) <= U is a CANDIDATE and the true nearest row (and every exact tie of it) is among
// them (measured on the oracle's features of an NCLT pair: 15.4 records per query against 14.7 with all cross terms; hi halves alone would
// have given 19 with a tail of several hundred on 1 % of the queries).  The per-row terms are prepared by the split (FnRows: nlo = |x|^2 - E,
// nup = 2 E, cq = 2 E with their margins), so the hot loop has no per-row multiply.  The screen keeps U per query in registers and writes
// every candidate as a RECORD (query, row, lower bound) into a pool
// (a few dozen per query: the running minimum of a sequence improves ~ln N times; wavefronts take 64-record chunks of the pool with
// one atomic per chunk and fill them with plain stores, so the hot loop never waits for a returning atomic).  k_fn_exact_min /
// k_fn_exact_arg then evaluate sum_k (a_k - b_k)^2 in float64, in index order with separately rounded products and sums exactly like
// the oracle's kd-tree leaf loop, on the records that survive the FINAL bound of their query: minimum distance first, then the
// smallest row among the records that attain it.  The result is therefore the exact float64 nearest row whatever the screen's
// rounding does inside its bound; if the pool overflows (pathological duplicate structure) the caller falls back to the all-pairs
// float64 path.  PCR_FEATNN_CHECK=1 measures the bound's slack (largest |d~ - d| / (E_q + E_b) over the records: must stay under 1).
//
// Kernel shape (gfx950): workgroup = 8 wavefronts = 512 queries; a wavefront keeps the B operands of its 64 queries (4 blocks of
// 16) in 32 VGPRs for the whole kernel; the database streams through LDS in steps of 64 rows (8 KB), staged once per workgroup,
// double buffered, in a [k-group][row] image with a 40-byte row pitch on which the 8-byte operand reads are conflict free;
// per 16-row tile a wavefront issues 8 MFMAs (128 cycles on its SIMD) and 29 VALU instructions of bound tests and keeps only one
// hit bit per 16 x 16 block; blocks with a hit (a few per cent) are recomputed after the step's four tiles by the candidate path,
// so that nothing but the barrier separates the MFMA streams of consecutive tiles.
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "pcr_device.h"

#define FN_D 33
#ifndef FN_K
#define FN_K 64                  // length of the screen's dot product: 64 (33 hi + both cross terms of the FN_NCOV widest dimensions) or 128 (all cross terms, round 2-3)
#endif
#define FN_NM (FN_K / 32)        // MFMAs per 16 x 16 block
#define FN_NCOV ((FN_K - FN_D - 1) / 2 < FN_D ? (FN_K - FN_D - 1) / 2 : FN_D)     // dimensions whose hi.lo cross terms are in the product: 15 at K = 64, all 33 at K = 128
#ifndef FN_WG
#define FN_WG 256                // 4 wavefronts = 256 queries per workgroup, four workgroups per CU at K = 64 (512 threads: 10.1 ms against 9.6 at 200k x 200k rows, both directions)
#endif
#ifndef FN_GRP
#define FN_GRP 1                 // database tiles staged and computed between two workgroup barriers (LDS: 2 x FN_GRP x 18 KB); 2 / 4 measured: no gain, see below
#endif
#ifndef FN_QB
#define FN_QB 4                  // query blocks of 16 per wavefront
#endif
#define FN_QPW (16 * FN_QB)      // queries per wavefront
#define FN_QPG (FN_QPW * (FN_WG / 64))
#define FN_STEP 64               // database rows per staged step (four 16-row MFMA tiles)
#define FN_SUBS (FN_STEP / 16)
#define FN_MIN_SPS 32              // steps a workgroup of the unpruned screen walks at least (see pcr_feature_nn_mutual)
#define FN_CHUNK 64              // records a wavefront takes from the pool at a time
#define FN_POOL_PER_QUERY 96     // pool capacity = this many records per query (expected: 20-40)
#define FN_SCALE 128.0           // features are centred and scaled by a power of two before the split (keeps lo out of the f16 subnormals)
#define FN_C 2.0e-6              // |d~ - d| <= FN_C (|x_q|^2 + |x_b|^2) over the covered dimensions; worst case of the analysis above is 0.8e-6
#define FN_CU 9.9e-4             // ... + FN_CU (|x_q|_U^2 + |x_b|_U^2) over the dimensions WITHOUT cross terms (hi . hi only: 2 * (2^-10 + 2^-22) / 2, rounded up)
#define FN_EABS 1.0e-6           // ... + an absolute term per row (f16 subnormals of a hi half: 33 * 2^-40 in scaled units, with room)
#define FN_PITCH (FN_NM * 16 + 8)        // LDS image of one 16-row tile: 4 k-groups x 16 rows; row pitch 72 B (K = 128) / 40 B (K = 64): the 8-byte
#define FN_GROUP (16 * FN_PITCH)         //   operand reads of a half-wavefront (16 rows x 2 k-groups) fall on 64 different banks either way
#define FN_SUB_BYTES (4 * FN_GROUP)

// 32-bit LDS byte address of a __shared__ object (what ds_read takes)
__device__ static inline unsigned fn_lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p; }

// order-preserving map float <-> int (its own inverse): atomicMin on the image is a minimum of the floats, negative ones included
__device__ static inline int fn_ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ static inline float fn_unord(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// a + b rounded up: the nearest-rounded sum plus 2^-23 of its magnitude (no directed-rounding add on the device side of this toolchain)
__device__ static inline float fn_add_up(float a, float b) { const float s = a + b; return __fmaf_rn(fabsf(s), 1.2e-7f, s); }

typedef _Float16 fn_h8 __attribute__((ext_vector_type(8)));
typedef float fn_f4 __attribute__((ext_vector_type(4)));

// ---- column means of a feature matrix (the centring vector; any vector is valid, the mean keeps the norms small) -------------
// part: gridDim.x rows of 34 = 33 column sums + the largest |value| (the f16 split needs |f - mu| * 128 < 65504)
#define FN_PC (2 * FN_D + 1)      // [0, 33) column sums, [33] largest |value|, [34, 67) column sums of squares; the mu vector has the same length: [0, 33) means, [33] largest
                                  // |value|, [34, 34 + FN_NCOV) the covered dimensions (ascending, as doubles)
// (round 5: a lane per COLUMN, a wavefront per row -- coalesced 132-byte reads and two accumulators per lane; a row per lane kept 66 float64
// accumulators in 254 VGPRs and read 64 rows 132 bytes apart per instruction)
__device__ static inline void d_fn_colsum(const float *__restrict__ f, int n, double *__restrict__ part) {
    __shared__ double sh[256 / 64][FN_PC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool col = lane < FN_D;
    double s = 0.0, sq = 0.0, mx = 0.0;
    const int stride = gridDim.x * (256 / 64);
    for (int r0 = blockIdx.x * (256 / 64) + w; r0 < n; r0 += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int r = r0 + u * stride; v[u] = (col && r < n) ? f[(size_t)r * FN_D + lane] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const double x = (double)v[u]; s += x; sq += x * x; mx = fmax(mx, fabs(x)); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
    if (col) { sh[w][lane] = s; sh[w][FN_D + 1 + lane] = sq; }
    if (lane == 0) sh[w][FN_D] = mx;
    __syncthreads();
    if (threadIdx.x < FN_PC) {
        double v = 0;
        for (int k = 0; k < 256 / 64; k++) v = threadIdx.x != FN_D ? v + sh[k][threadIdx.x] : fmax(v, sh[k][threadIdx.x]);
        part[blockIdx.x * FN_PC + threadIdx.x] = v;
    }
}
__global__ void __launch_bounds__(256) k_fn_colsum(const float *__restrict__ f, int n, double *__restrict__ part) { d_fn_colsum(f, n, part); }
// batch forms (lockstep FGR groups, pcr_feature_nn_mutual_batch): blockIdx.y picks the matrix / the pair / the (pair, direction) problem
struct FnColsumDesc { const float *f; int n; double *part; };
__global__ void __launch_bounds__(256) k_fn_colsum_g(const FnColsumDesc *d) { const FnColsumDesc a = d[blockIdx.y]; d_fn_colsum(a.f, a.n, a.part); }
// mu[0..32] = column means of the first matrix, mu[33] = largest |value| over both
// (also resets the first-zero-row words and the overflow flags of the call: it runs before the splits that use them)
__device__ static inline void d_fn_mean(const double *__restrict__ part, int nb, int n, double *__restrict__ mu, int *__restrict__ first_zero, int *__restrict__ flags) {
    __shared__ double var[FN_D];
    __shared__ int taken[FN_D];
    if (threadIdx.x == 63) { first_zero[0] = 0x7fffffff; first_zero[1] = 0x7fffffff; flags[0] = 0; flags[1] = 0; }
    if (threadIdx.x < FN_PC) {
        double v = 0;
        for (int k0 = 0; k0 < 2 * nb; k0 += 16) {          // (16 loads in flight: one after the other this loop alone was 80 us)
            double x[16];
#pragma unroll
            for (int u = 0; u < 16; u++) x[u] = k0 + u < 2 * nb ? part[(k0 + u) * FN_PC + threadIdx.x] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; u++) { const int k = k0 + u; if (threadIdx.x == FN_D) v = fmax(v, x[u]); else if (k < nb) v += x[u]; }
        }
        if (threadIdx.x <= FN_D) mu[threadIdx.x] = threadIdx.x == FN_D ? v : (n > 0 ? v / (double)n : 0.0);
        else var[threadIdx.x - FN_D - 1] = v;                              // sum of squares of the column (first matrix)
    }
    __syncthreads();
    // the FN_NCOV columns of the largest variance get their cross terms into the screen's product (any choice is valid: it only sets how tight the
    // screen's bound is; on NCLT scans 15 of the 33 FPFH bins hold 93 % of the variance)
    if (threadIdx.x == 0) {
        for (int k = 0; k < FN_D; k++) { taken[k] = 0; var[k] = n > 0 ? var[k] / (double)n - mu[k] * mu[k] : 0.0; }
        for (int j = 0; j < FN_NCOV; j++) {
            int best = -1;
            for (int k = 0; k < FN_D; k++) if (!taken[k] && (best < 0 || var[k] > var[best])) best = k;
            taken[best] = 1;
        }
        int j = 0;
        for (int k = 0; k < FN_D; k++) if (taken[k]) mu[FN_D + 1 + j++] = (double)k;
        for (; j < FN_D; j++) mu[FN_D + 1 + j] = -1.0;
    }
}
__global__ void k_fn_mean(const double *__restrict__ part, int nb, int n, double *__restrict__ mu, int *__restrict__ first_zero, int *__restrict__ flags) { d_fn_mean(part, nb, n, mu, first_zero, flags); }
struct FnMeanDesc { const double *part; int nb, n; double *mu; int *first_zero, *flags; };
__global__ void k_fn_mean_g(const FnMeanDesc *d) { const FnMeanDesc a = d[blockIdx.x]; d_fn_mean(a.part, a.nb, a.n, a.mu, a.first_zero, a.flags); }

// ---- split: one thread per row.  A-form (database role) [hi | hi | lo | 0], B-form (query role) [hi | lo | hi | 0]; norm of the
// exact centred, scaled row; nlo = (1 - c) |x|^2 rounded down (the lower-bound test needs no per-row multiply in the hot loop).
// Padded rows: zeros with nlo = +huge (can never be a candidate).
// All-zero rows (Open3D leaves the FPFH of a point without neighbours at zero; 1-2 % of an outdoor scan) are exact duplicates in bulk:
// every zero query ties with every zero row and its candidate list would overflow.  Their answer is known -- the first zero row of
// the database, at distance exactly 0 -- so the split marks them (sign bit of nrm) and records the first zero row of each matrix.
// perm (optional): row i of the forms is row perm[i] of f (the tile-pruned screen works on rows in Morton order of their leading
// principal coordinates); first_zero is always an index into f.
__device__ static inline void d_fn_split(const float *__restrict__ f_, int n, int n_pad, const double *__restrict__ mu,
                                         _Float16 *__restrict__ A, _Float16 *__restrict__ B, float *__restrict__ nlo, float *__restrict__ nup, float *__restrict__ nrm,
                                         float *__restrict__ cq, int *__restrict__ first_zero, const uint32_t *__restrict__ perm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    const int src = (i < n && perm) ? (int)perm[i] : i;
    const float *__restrict__ f = f_;
    _Float16 hi[FN_D], lo[FN_D];
    double s = 0.0;
    bool zero = i < n;
#pragma unroll
    for (int k = 0; k < FN_D; k++) {
        zero = zero && (i < n ? f[(size_t)src * FN_D + k] == 0.0f : false);
        const double x = i < n ? ((double)f[(size_t)src * FN_D + k] - mu[k]) * FN_SCALE : 0.0;
        const _Float16 h = (_Float16)(float)x;
        hi[k] = h;
        lo[k] = (_Float16)(float)(x - (double)(float)h);
        s += x * x;
    }
    // covered dimensions (the same for every row of the call): hc / lc = their halves, taken from the row again (it is in the L1 by now; picking
    // them out of hi[] / lo[] by a run-time index costs a thousand selects per row), su = the squared norm over the others
    _Float16 hc[FN_NCOV], lc[FN_NCOV];
    double sc = 0.0;
    if (FN_NCOV == FN_D) {
#pragma unroll
        for (int j = 0; j < FN_NCOV; j++) { hc[j] = hi[j]; lc[j] = lo[j]; }
        sc = s;
    } else {
#pragma unroll
        for (int j = 0; j < FN_NCOV; j++) {
            const int cd = (int)mu[FN_D + 1 + j];
            const double x = i < n ? ((double)f[(size_t)src * FN_D + cd] - mu[cd]) * FN_SCALE : 0.0;      // the same arithmetic as above: the same halves
            const _Float16 h = (_Float16)(float)x;
            hc[j] = h; lc[j] = (_Float16)(float)(x - (double)(float)h); sc += x * x;
        }
    }
    const double su = fmax(s - sc, 0.0) * (1.0 + 1e-12) + (FN_NCOV == FN_D ? 0.0 : 1e-9 * s);    // (s - sc carries the rounding of 33 float64 sums)
    // the two rows are put together in registers and leave as 16-byte stores (element-wise 2-byte stores made this kernel 1.2 ms per
    // lockstep group of 16 NCLT-size pairs: 0.3 TB/s)
    fn_h8 ra[FN_K / 8], rb[FN_K / 8];
#pragma unroll
    for (int k = 0; k < FN_K; k++) {
        const _Float16 z = (_Float16)0.0f;
        const _Float16 va = k < FN_D ? hi[k] : (k < FN_D + FN_NCOV ? hc[k - FN_D] : (k < FN_D + 2 * FN_NCOV ? lc[k - FN_D - FN_NCOV] : z));
        const _Float16 vb = k < FN_D ? hi[k] : (k < FN_D + FN_NCOV ? lc[k - FN_D] : (k < FN_D + 2 * FN_NCOV ? hc[k - FN_D - FN_NCOV] : z));
        ra[k / 8][k % 8] = va; rb[k / 8][k % 8] = vb;
    }
    fn_h8 *a = (fn_h8 *)(A + (size_t)i * FN_K), *b = (fn_h8 *)(B + (size_t)i * FN_K);
#pragma unroll
    for (int c = 0; c < FN_K / 8; c++) { a[c] = ra[c]; b[c] = rb[c]; }
    if (i < n) {
        const double E = FN_C * s + FN_CU * su + FN_EABS;                  // this row's share of the bound |d~ - d| <= E_q + E_b
        const float lo_f = __double2float_rd(s - E);
        nlo[i] = lo_f;
        nup[i] = __double2float_ru(2.0 * E * 1.002 + ((s - E) - (double)lo_f));      // wlo + nup >= d~ + E_b - |x_q|^2 (with what rounding nlo down lost)
        nrm[i] = zero ? -__double2float_ru(s) : __double2float_ru(s);
        cq[i] = __double2float_ru(2.0 * E * 1.001);
        if (zero) atomicMin(first_zero, src);
    } else { nlo[i] = 1.0e30f; nup[i] = 0.0f; nrm[i] = 0.0f; cq[i] = 0.0f; }
}
// The same split with 8 lanes per row (K = 64; round 5).  A row per lane held the whole row in registers (272 VGPRs: one wavefront per SIMD) and
// stored its two 128-byte rows in 16-byte pieces 128 bytes apart; here lane p of an octet makes the p-th 16-byte piece of both rows -- the
// halves of the (up to) 8 dimensions its slots hold, computed by the same expressions as above, so the SAME halves -- and an octet's
// stores are one contiguous 128-byte line per row.  The squared norms are octet sums of per-lane partial sums (another summation order
// than the sequential one above: |x|^2 differs in its last bits, which the bounds' margins cover; the matches are exact either way).
#if FN_K == 64
__device__ static inline void d_fn_split8(const float *__restrict__ f_, int n, int n_pad, const double *__restrict__ mu,
                                          _Float16 *__restrict__ A, _Float16 *__restrict__ B, float *__restrict__ nlo, float *__restrict__ nup, float *__restrict__ nrm,
                                          float *__restrict__ cq, int *__restrict__ first_zero, const uint32_t *__restrict__ perm) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 3, part = t & 7;
    if (i >= n_pad) return;                               // (octet-uniform; n_pad is a multiple of 32)
    const bool real = i < n;
    const int src = (real && perm) ? (int)perm[i] : i;
    const float *__restrict__ row = f_ + (size_t)(real ? src : 0) * FN_D;
    fn_h8 ra, rb;
    double sp = 0.0, scp = 0.0; int nz = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = 8 * part + u;                        // slot of the 64-entry rows: [hi 0..32 | hc / lc 0..14 | lc / hc 0..14 | 0]
        const bool plain = k < FN_D, first = k < FN_D + FN_NCOV, pad = k >= FN_D + 2 * FN_NCOV;
        int dim = 0;
        if (plain) dim = k;
        else if (!pad) dim = (int)mu[FN_D + 1 + (first ? k - FN_D : k - FN_D - FN_NCOV)];
        const float fv = (real && !pad) ? row[dim] : 0.0f;
        const double x = (real && !pad) ? ((double)fv - mu[dim]) * FN_SCALE : 0.0;
        const _Float16 h = (_Float16)(float)x;
        const _Float16 l = (_Float16)(float)(x - (double)(float)h);
        const _Float16 z = (_Float16)0.0f;
        ra[u] = pad ? z : (first ? h : l);                 // A: hi | hc | lc
        rb[u] = pad ? z : (plain ? h : (first ? l : h));   // B: hi | lc | hc
        if (plain) { sp += x * x; nz += (real && fv != 0.0f) ? 1 : 0; }
        else if (first) scp += x * x;
    }
    reinterpret_cast<fn_h8 *>(A + (size_t)i * FN_K)[part] = ra;
    reinterpret_cast<fn_h8 *>(B + (size_t)i * FN_K)[part] = rb;
    const double s = pcr_octet_sum(sp), sc = pcr_octet_sum(scp);
    nz = pcr_octet_sum_i(nz);
    if (part != 0) return;
    if (real) {
        const bool zero = nz == 0;
        const double su = fmax(s - sc, 0.0) * (1.0 + 1e-12) + 1e-9 * s;
        const double E = FN_C * s + FN_CU * su + FN_EABS;
        const float lo_f = __double2float_rd(s - E);
        nlo[i] = lo_f;
        nup[i] = __double2float_ru(2.0 * E * 1.002 + ((s - E) - (double)lo_f));
        nrm[i] = zero ? -__double2float_ru(s) : __double2float_ru(s);
        cq[i] = __double2float_ru(2.0 * E * 1.001);
        if (zero) atomicMin(first_zero, src);
    } else { nlo[i] = 1.0e30f; nup[i] = 0.0f; nrm[i] = 0.0f; cq[i] = 0.0f; }
}
#define FN_SPLIT_LANES 8
#else
#define d_fn_split8 d_fn_split
#define FN_SPLIT_LANES 1
#endif
// the four per-row floats of a cloud: nlo = |x|^2 - E rounded down (database role, lower bounds), nup = 2 E with margins (database role, upper
// bounds), nrm = |x|^2 rounded up with the sign bit marking an all-zero feature row, cq = 2 E with margins (query role, candidate threshold)
struct FnRows { float *nlo, *nup, *nrm, *cq; };
__global__ void __launch_bounds__(256) k_fn_split(const float *__restrict__ f_, int n, int n_pad, const double *__restrict__ mu,
                                                  _Float16 *__restrict__ A, _Float16 *__restrict__ B, FnRows r,
                                                  int *__restrict__ first_zero, const uint32_t *__restrict__ perm) { d_fn_split8(f_, n, n_pad, mu, A, B, r.nlo, r.nup, r.nrm, r.cq, first_zero, perm); }
struct FnSplitDesc { const float *f; int n, n_pad; const double *mu; _Float16 *A, *B; FnRows r; int *first_zero; };
__global__ void __launch_bounds__(256) k_fn_split_g(const FnSplitDesc *d) { const FnSplitDesc a = d[blockIdx.y]; d_fn_split8(a.f, a.n, a.n_pad, a.mu, a.A, a.B, a.r.nlo, a.r.nup, a.r.nrm, a.r.cq, a.first_zero, nullptr); }

// ================================================================================================ tile pruning (round 2)
// The all-pairs screen costs 4 MFMAs per 16 x 16 block whatever the data.  FPFH rows are far from uniform in their 33-D space (four
// principal directions hold 87 % of the variance), so rows are put in Morton order of their four leading principal coordinates
// (8 bits each): 64 consecutive rows -- one staged step of the database, one wavefront of queries -- then sit in a small box, and
//     (distance between the box of a query wavefront and the box of a row tile)^2  >  max over its queries of (best distance so far)
// proves that the tile holds neither the nearest row nor a tie of any of the 64 queries.  The boxes are taken in ALL 33 principal
// coordinates (any orthonormal basis gives a valid bound; this one gives the tightest boxes).  On 200k-point scans 17 % of the
// (wavefront, tile) pairs survive (measured on the oracle's features: 17.4 % with the bound after a pre-pass over the 48 tiles with
// the smallest box distance, 16.9 % with the final bound).  Nothing else changes: survivors go through the same screen, the same
// records and the same exact float64 re-check, and ties still go to the smaller ORIGINAL row index.
#define FN_NG (FN_D * (FN_D + 1) / 2)      // 561 second moments
#define FN_MD 4                            // principal coordinates in the Morton key
#define FN_MB 8                            //   bits of each
#ifndef FN_NPRE
#define FN_NPRE 48                         // pre-pass: tiles per workgroup (round 5, registro_FGR at 200k points: 24 / 32 / 48 / 64 -> 12.45 / 12.56 / 12.27 / 12.20 ms)
#endif
#ifndef FN_WGS_PRUNED
#define FN_WGS_PRUNED 18432                // workgroups of the pruned main pass (query groups x splits of the database)
#endif
#ifndef FN_XCD_ORDER
#define FN_XCD_ORDER 0                     // XCD-aware order of the query groups in the pruned main pass: measured 9.8 ms against 9.1 (200k x 200k rows) -- the hard
#endif                                     //   queries are neighbours and pile up on one XCD, as in the k-NN kernels (DESIGN section 8, round-2 list, item 2); off
#ifndef FN_PRE_SPLIT
#define FN_PRE_SPLIT 3                     // workgroups that share the pre-pass list of a query group (records pass / bound-only sweep)
#endif
#ifndef FN_PRE_SPLIT_B
#define FN_PRE_SPLIT_B 3
#endif
#ifndef FN_NBOUND
#define FN_NBOUND 48                       // of which the bound-only sweep covers the nearest ...
#endif
#ifdef FN_LIST_CAP_OVERRIDE
#define FN_LIST_CAP FN_LIST_CAP_OVERRIDE
#else
#define FN_LIST_CAP 512
#endif
//                  // steps of one (workgroup, split): capacity of its list
#define FN_GRAM_SKIP 4                     // the second moments are taken on a quarter of the rows (the axes only have to be good; the means are exact)

// raw second moments sum_i f_i[a] f_i[b], a <= b (thread t owns one pair; rows staged through LDS 32 at a time)
__global__ void __launch_bounds__(576) k_fn_gram(const float *__restrict__ f, int n, double *__restrict__ part) {
    __shared__ float rows[32][FN_D + 1];
    int pa = 0, pb = 0;
    { int t = threadIdx.x; for (pa = 0; pa < FN_D; pa++) { const int len = FN_D - pa; if (t < len) { pb = pa + t; break; } t -= len; } }
    const bool owner = threadIdx.x < FN_NG;
    double acc = 0.0;
    for (int r0 = blockIdx.x * 32; r0 < n; r0 += gridDim.x * 32 * FN_GRAM_SKIP) {      // every FN_GRAM_SKIP-th block of 32 rows per round of the grid
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * FN_D; e += 576) { const int r = e / FN_D, k = e % FN_D; rows[r][k] = r0 + r < n ? f[(size_t)(r0 + r) * FN_D + k] : 0.0f; }
        __syncthreads();
        if (owner)
#pragma unroll 8
            for (int r = 0; r < 32; r++) acc += (double)rows[r][pa] * (double)rows[r][pb];
    }
    if (owner) part[(size_t)blockIdx.x * FN_NG + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(576) k_fn_gram_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    if (threadIdx.x >= FN_NG) return;
    double v = 0.0;
    for (int k = 0; k < nb; k++) v += part[(size_t)k * FN_NG + threadIdx.x];
    out[threadIdx.x] = v;
}

// cyclic Jacobi on the host (33 x 33, a few hundred microseconds): V columns = eigenvectors, ev descending
static void fn_jacobi(double *C /* 33 x 33, destroyed */, double *V, double *ev) {
    const int N = FN_D;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) V[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
        for (int p_ = 0; p_ < N; p_++) for (int q = p_ + 1; q < N; q++) off += C[p_ * N + q] * C[p_ * N + q];
        double diag = 0.0;
        for (int i = 0; i < N; i++) diag += C[i * N + i] * C[i * N + i];
        if (!(off > 1e-16 * diag)) break;        // V stays orthonormal whatever the residual: the axes only have to be good, not converged
        for (int p_ = 0; p_ < N; p_++)
            for (int q = p_ + 1; q < N; q++) {
                const double apq = C[p_ * N + q];
                if (apq == 0.0) continue;
                const double theta = (C[q * N + q] - C[p_ * N + p_]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < N; k++) { const double a = C[k * N + p_], b = C[k * N + q]; C[k * N + p_] = c * a - sn * b; C[k * N + q] = sn * a + c * b; }
                for (int k = 0; k < N; k++) { const double a = C[p_ * N + k], b = C[q * N + k]; C[p_ * N + k] = c * a - sn * b; C[q * N + k] = sn * a + c * b; }
                for (int k = 0; k < N; k++) { const double a = V[k * N + p_], b = V[k * N + q]; V[k * N + p_] = c * a - sn * b; V[k * N + q] = sn * a + c * b; }
            }
    }
    int order[FN_D];
    for (int i = 0; i < N; i++) order[i] = i;
    std::sort(order, order + N, [&](int a, int b) { return C[a * N + a] > C[b * N + b]; });
    std::vector<double> Vs((size_t)N * N);
    for (int k = 0; k < N; k++) { ev[k] = C[order[k] * N + order[k]]; for (int j = 0; j < N; j++) Vs[(size_t)j * N + k] = V[j * N + order[k]]; }
    std::memcpy(V, Vs.data(), sizeof(double) * N * N);
}

// rot: [0, 1089) V (x_j -> coordinate k: V[j * 33 + k]), then 4 x (lo, inv) of the key quantisation.
// P[i][k] = float32 of the float64 principal coordinate k of the centred, scaled row i (same x as the split); key = Morton code of the four
// leading ones.
__global__ void __launch_bounds__(256) k_fn_rotate(const float *__restrict__ f, int n, const double *__restrict__ mu, const double *__restrict__ rot,
                                                   float *__restrict__ P, uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double x[FN_D];
#pragma unroll
    for (int j = 0; j < FN_D; j++) x[j] = ((double)f[(size_t)i * FN_D + j] - mu[j]) * FN_SCALE;
    uint32_t q[FN_MD];
    for (int k = 0; k < FN_D; k++) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < FN_D; j++) s += rot[j * FN_D + k] * x[j];
        P[(size_t)i * FN_D + k] = (float)s;
        if (k < FN_MD) {
            const double u = (s - rot[FN_D * FN_D + 2 * k]) * rot[FN_D * FN_D + 2 * k + 1];
            q[k] = (uint32_t)(u < 0.0 ? 0.0 : (u > (double)((1 << FN_MB) - 1) ? (double)((1 << FN_MB) - 1) : u));
        }
    }
    uint64_t m = 0;
    for (int b = FN_MB - 1; b >= 0; b--)
#pragma unroll
        for (int k = 0; k < FN_MD; k++) m = (m << 1) | ((q[k] >> b) & 1u);
    key[i] = m; val[i] = (uint32_t)i;
}

// box of every 64-row tile (rows in sorted order) in all 33 principal coordinates: lo / hi[k * tile_stride + tile]; one wavefront per tile
__global__ void __launch_bounds__(256) k_fn_boxes(const float *__restrict__ P, const uint32_t *__restrict__ perm, int n, int n_tiles, int tile_stride,
                                                  float *__restrict__ lo, float *__restrict__ hi) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    const int i = tile * 64 + lane;
    const float *row = i < n ? P + (size_t)perm[i] * FN_D : nullptr;
    for (int k = 0; k < FN_D; k++) {
        const float v = row ? row[k] : 0.0f;
        const float mn = pcr_wave_min(row ? v : 1.0e30f), mx = pcr_wave_max(row ? v : -1.0e30f);
        if (lane == 0) { lo[(size_t)k * tile_stride + tile] = mn; hi[(size_t)k * tile_stride + tile] = mx; }
    }
}

// L[qt][t] = lower bound of the squared distance between any row of query tile qt and any row of database tile t.  eps2 = twice the
// largest float32 rounding error of a stored coordinate; the factor covers the float32 summation.
__global__ void __launch_bounds__(256) k_fn_boxlb(const float *__restrict__ qlo, const float *__restrict__ qhi, int q_stride, const float *__restrict__ tlo,
                                                  const float *__restrict__ thi, int t_stride, int n_t, float eps2, float *__restrict__ L, int L_stride) {
    const int t = blockIdx.x * 256 + threadIdx.x, qt = blockIdx.y;
    if (t >= n_t) return;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < FN_D; k++) {
        const float a = tlo[(size_t)k * t_stride + t] - qhi[(size_t)k * q_stride + qt], b = qlo[(size_t)k * q_stride + qt] - thi[(size_t)k * t_stride + t];
        const float g = fmaxf(fmaxf(a, b) - eps2, 0.0f);
        acc = __fmaf_rn(g, g, acc);
    }
    L[(size_t)qt * L_stride + t] = acc * 0.99999f;
}

// pre-pass list of a query workgroup (8 query tiles): the FN_NPRE database tiles with the smallest box distance to any of them (ties,
// typically many tiles at distance 0, are thinned evenly).  One workgroup of 256 threads per query workgroup.
#define FN_LM_CAP 12288                    // tiles whose minimum box distance fits the LDS cache of k_fn_prelist (786k rows); beyond: recomputed
__global__ void __launch_bounds__(256) k_fn_prelist(const float *__restrict__ L, int L_stride, int n_qt, int n_t, int *__restrict__ prelist) {
    __shared__ int cnt[2];
    __shared__ int scan[256][2];
    __shared__ float lm[FN_LM_CAP];
    __shared__ int sel_t[FN_NPRE];
    __shared__ float sel_l[FN_NPRE];
    const int g = blockIdx.x, qt0 = g * (FN_WG / 64);
    const int nw = n_qt - qt0 < FN_WG / 64 ? n_qt - qt0 : FN_WG / 64;
    int *out = prelist + (size_t)g * FN_NPRE;
    if (nw <= 0) { if (threadIdx.x < FN_NPRE) out[threadIdx.x] = -1; return; }
    if (threadIdx.x < FN_NPRE) { sel_t[threadIdx.x] = -1; sel_l[threadIdx.x] = __builtin_inff(); }
    auto lmin_g = [&](int t) { float m = L[(size_t)qt0 * L_stride + t]; for (int w = 1; w < nw; w++) m = fminf(m, L[(size_t)(qt0 + w) * L_stride + t]); return m; };
    const bool cached = n_t <= FN_LM_CAP;
    if (cached) for (int t = threadIdx.x; t < n_t; t += 256) lm[t] = lmin_g(t);
    __syncthreads();
    auto lmin = [&](int t) { return cached ? lm[t] : lmin_g(t); };
    const int per = (n_t + 255) / 256, t_lo = threadIdx.x * per, t_hi = t_lo + per < n_t ? t_lo + per : n_t;
    // smallest tau (as bits of a non-negative float) with at least min(FN_NPRE, n_t) tiles at or under it
    const int want = n_t < FN_NPRE ? n_t : FN_NPRE;
    unsigned lo_b = 0u, hi_b = 0x7f7fffffu;
    while (lo_b < hi_b) {
        const unsigned mid = lo_b + (hi_b - lo_b) / 2;
        if (threadIdx.x == 0) cnt[0] = 0;
        __syncthreads();
        int c = 0;
        for (int t = t_lo; t < t_hi; t++) c += __float_as_uint(lmin(t)) <= mid ? 1 : 0;
        if (c) atomicAdd(&cnt[0], c);
        __syncthreads();
        const int total = cnt[0];
        __syncthreads();
        if (total >= want) hi_b = mid; else lo_b = mid + 1;
    }
    const unsigned tau = lo_b;
    // ranks in index order: tiles under tau are all taken, tiles at tau are thinned to the remaining quota
    int c_lt = 0, c_eq = 0;
    for (int t = t_lo; t < t_hi; t++) { const unsigned b = __float_as_uint(lmin(t)); c_lt += b < tau; c_eq += b == tau; }
    scan[threadIdx.x][0] = c_lt; scan[threadIdx.x][1] = c_eq;
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 256; k++) { const int x = scan[k][0], y = scan[k][1]; scan[k][0] = a; scan[k][1] = b; a += x; b += y; }
        cnt[0] = a; cnt[1] = b;
    }
    __syncthreads();
    const int n_lt = cnt[0], n_eq = cnt[1];
    const int quota = want - n_lt;                                     // > 0 by the choice of tau
    const int stride = quota > 0 ? (n_eq + quota - 1) / quota : 1;
    int r_lt = scan[threadIdx.x][0], r_eq = scan[threadIdx.x][1];
    for (int t = t_lo; t < t_hi; t++) {
        const float v = lmin(t);
        const unsigned b = __float_as_uint(v);
        if (b < tau) { if (r_lt < FN_NPRE) { sel_t[r_lt] = t; sel_l[r_lt] = v; } r_lt++; }
        else if (b == tau) { const int slot = n_lt + r_eq / stride; if (r_eq % stride == 0 && slot < FN_NPRE) { sel_t[slot] = t; sel_l[slot] = v; } r_eq++; }
    }
    __syncthreads();
    // nearest first (the screen's bound then falls at once): rank sort of the 48 entries, unused ones (-1, +inf) last
    if (threadIdx.x < FN_NPRE) {
        const float v = sel_l[threadIdx.x]; const int t = sel_t[threadIdx.x];
        int rank = 0;
        for (int k = 0; k < FN_NPRE; k++) { const float u = sel_l[k]; rank += (u < v || (u == v && k < (int)threadIdx.x)) ? 1 : 0; }
        out[rank] = t;
    }
}

struct FnnArgs {
    const _Float16 *dbA; const float *db_nlo, *db_nup; int n_db_pad;      // database rows (A-form), |x|^2 - E and 2 E (FnRows)
    const _Float16 *qB; const float *q_nrm, *q_cq; int n_q, n_q_pad;      // query rows (B-form), |x|^2 (sign bit set: all-zero feature row) and 2 E
    const int *db_first_zero;                                      // first all-zero row of the database (INT_MAX: none)
    int step0, steps_per_split, step_end;                          // 64-row steps [step0 + split * sps, +sps) clipped to step_end
    // tile pruning (null L: every step of the range is processed)
    const float *L; int L_stride; int n_qt;                        // L[query tile][database tile]: lower bound of the squared distance between their boxes
    const int *prelist;                                            // FN_NPRE steps per query workgroup (-1: unused)
    int pre_mode;                                                  // 1: process the workgroup's prelist; 0: the split's range minus the prelist
    int n_bound;                                                   // BOUND_ONLY launch: entries of the prelist it covers
    int xcd_chunk;                                                 // > 0: query groups in XCD-aware order, grid.x = 8 * xcd_chunk (see the kernel)
    unsigned long long *stats;                                     // diagnostics (PCR_FEATNN_CHECK): [0] steps staged, [1] (wavefront, step) pairs computed, [2] of all
    int *Ug;                                                       // per query: upper bound of (minimum - |x_q|^2): read when a workgroup starts, lowered
                                                                   //   (atomicMin on the order-preserving int image) when it ends
    // record pool: chunk c holds records [c * FN_CHUNK, +chunk_fill[c]); pool_used counts allocated records; flags[0] = overflow
    int *pool_used; int pool_cap; int *chunk_fill; int *rec_q; int *rec_row; float *rec_w; int *flags;
    int seeded = 0;                                                // 1 (second direction of a MUTUAL search): Ug comes seeded by k_fn_seed; a query still at +inf cannot be
                                                                   //   a mutual match and produces no candidates
};

// ---- the screen.  grid = (query groups of 512, splits of the database)
// BOUND_ONLY: no candidates and no records, only the queries' upper bounds are lowered (over the first a.n_bound entries of the pre-pass list)
template <bool BOUND_ONLY>
__device__ static inline void d_feature_nn_screen(const FnnArgs &a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * FN_GRP][FN_SUBS * FN_SUB_BYTES];
    __shared__ __attribute__((aligned(16))) float lnlo[2 * FN_GRP][FN_STEP];
    __shared__ __attribute__((aligned(16))) float lnup[2 * FN_GRP][FN_STEP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    // XCD-aware order of the query groups (xcd_chunk > 0: the grid is 8 * xcd_chunk wide): workgroup ids go round-robin over the 8 XCDs, so XCD x
    // takes the contiguous groups [x * chunk, (x + 1) * chunk) -- neighbours in the Morton order of the queries, whose step lists overlap: the
    // part of the database one L2 has to hold is what THEY need, not what all the queries of the cloud need
    const int bx = a.xcd_chunk > 0 ? (int)(blockIdx.x & 7u) * a.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (bx * FN_QPG >= a.n_q_pad) return;             // (batch launches: the grid is the largest problem's)
    const int q0 = (bx * (FN_WG / 64) + wv) * FN_QPW;
    const int s0 = a.pre_mode ? 0 : a.step0 + blockIdx.y * a.steps_per_split;
    const int s1 = a.pre_mode ? 0 : min(a.step_end, s0 + a.steps_per_split);
    if (!a.pre_mode && s0 >= s1) return;
    __shared__ int slist[FN_LIST_CAP];
    __shared__ float slw[FN_WG / 64][FN_LIST_CAP];         // box distance of list entry i from wavefront w's queries (pruned form; else zeros are never read)
    __shared__ int n_list_s;
    __shared__ float sD[FN_WG / 64];
    __shared__ unsigned pre_bits[FN_LIST_CAP / 32];
    // B operands of the wavefront's 64 queries: block b, MFMA m <- k-chunk (g, m) of query q0 + 16 b + col
    fn_h8 qb[FN_QB][FN_NM];
    float cq[FN_QB], U[FN_QB], thrh[FN_QB], nqv[FN_QB];        // thrh = -(U + cq) / 2: the hit threshold on the accumulator scale
#pragma unroll
    for (int b = 0; b < FN_QB; b++) {
        const int q = q0 + 16 * b + col;
        const int qc = q < a.n_q_pad ? q : 0;
        const uint4 *src = (const uint4 *)(a.qB + (size_t)qc * FN_K + g * (FN_K / 4));
#pragma unroll
        for (int m = 0; m < FN_NM; m++) { union { uint4 u; fn_h8 h; } x; x.u = src[m]; qb[b][m] = x.h; }
        // candidate iff (lower bound of d~ - nq) < U + 2 E_q ; queries beyond n_q and zero queries with a known answer never produce candidates
        const float nr = a.q_nrm[qc];
        const bool known = __builtin_signbit(nr) && *a.db_first_zero != 0x7fffffff;
        cq[b] = (q < a.n_q && !known) ? a.q_cq[qc] : -__builtin_inff();
        nqv[b] = fabsf(nr);
        U[b] = fn_unord(a.Ug[qc]);                        // what the pre-pass has established
        if (a.seeded && U[b] == __builtin_inff()) cq[b] = -__builtin_inff();
        thrh[b] = -0.5f * (U[b] + cq[b]);
    }
    // D = the largest (upper bound of the best squared distance) over the wavefront's live queries: a tile whose box is farther than
    // that from the wavefront's box holds neither a nearest row nor a tie of it
    auto wave_D = [&]() {
        float d = -__builtin_inff();
#pragma unroll
        for (int b = 0; b < FN_QB; b++) {       // true best distance <= U + |x_q|^2 + E_q (cq = 2 E_q with a margin), rounded up
            const float t = U[b] + nqv[b] + 0.5f * cq[b];
            d = fmaxf(d, cq[b] == -__builtin_inff() ? -__builtin_inff() : __fmaf_rn(4.0e-7f, fabsf(U[b]) + nqv[b] + cq[b], t));
        }
        return pcr_wave_max_all(d);                     // (DPP + permlane swaps: no LDS round trips -- this runs after every step with a candidate)
    };
    float Dw = wave_D();
    const int qt = bx * (FN_WG / 64) + wv;         // the wavefront's query tile (row of L)
    const bool prune = a.L != nullptr;
    // ---- the workgroup's list of steps
    if (tid == 0) n_list_s = 0;
    if (lane == 0) sD[wv] = (prune && qt < a.n_qt) ? Dw : (prune ? -__builtin_inff() : __builtin_inff());
    if (tid < FN_LIST_CAP / 32) pre_bits[tid] = 0u;
    __syncthreads();
    if (a.pre_mode) {               // the list comes nearest-first with its unused entries at the end
        if (tid < 64) {
            // (gridDim.y workgroups share the list, entry e to workgroup e % gridDim.y: each gets some of the very nearest tiles)
            const int t = (tid < (BOUND_ONLY ? a.n_bound : FN_NPRE) && tid % (int)gridDim.y == (int)blockIdx.y) ? a.prelist[(size_t)bx * FN_NPRE + tid] : -1;
            const bool ok = t >= 0 && t < a.step_end;
            const unsigned long long okm = __ballot(ok);
            if (ok) slist[__builtin_popcountll(okm & ((1ull << tid) - 1ull))] = t;
            if (tid == 0) n_list_s = __builtin_popcountll(okm);
        }
    } else {
        if (a.prelist && tid < FN_NPRE) { const int t = a.prelist[(size_t)bx * FN_NPRE + tid]; if (t >= s0 && t < s1) atomicOr(&pre_bits[(t - s0) >> 5], 1u << ((t - s0) & 31)); }
        __syncthreads();
        for (int t = s0 + tid; t < s1; t += FN_WG) {
            bool need = !((pre_bits[(t - s0) >> 5] >> ((t - s0) & 31)) & 1u);
            float lwv[FN_WG / 64];
#pragma unroll
            for (int w = 0; w < FN_WG / 64; w++) lwv[w] = 0.0f;
            if (need && prune) {
                need = false;
#pragma unroll
                for (int w = 0; w < FN_WG / 64; w++) {
                    const int r = bx * (FN_WG / 64) + w;
                    lwv[w] = r < a.n_qt ? a.L[(size_t)r * a.L_stride + t] : __builtin_inff();
                    if (lwv[w] <= sD[w]) need = true;
                }
            }
            if (need) {
                const int pos = atomicAdd(&n_list_s, 1);
                slist[pos] = t;
                if (prune)
#pragma unroll
                    for (int w = 0; w < FN_WG / 64; w++) slw[w][pos] = lwv[w];
            }
        }
    }
    __syncthreads();
    const int n_list = n_list_s;
    if (a.stats && tid == 0) atomicAdd(&a.stats[0], (unsigned long long)n_list);
    if (n_list == 0) return;                               // nothing in this range can improve any of the 512 queries (their bounds are in Ug already)
    // staging: thread t moves FN_NCH 16-byte k-chunks of the step (rows r, r + FN_WG / 16, ..., chunk j); norms by the first 64 threads
    constexpr int FN_CPR = FN_K / 8;                                          // 16-byte chunks per row: k-group j / FN_NM, MFMA j % FN_NM
    constexpr int FN_NCH = FN_STEP * FN_CPR / FN_WG, FN_RSTEP = FN_WG / FN_CPR;   // chunks per thread (K = 64: 1 at 512 threads, 2 at 256), rows per round
    static_assert(FN_NCH >= 1 && FN_NCH * FN_RSTEP == FN_STEP, "staging: the workgroup must cover a step in whole rounds");
#ifndef FN_STASH_REMAP
#define FN_STASH_REMAP 1
#endif
#if FN_STASH_REMAP && FN_K == 64
    // K = 64 (8 chunks per row): the two 32-lane halves of a wavefront write k-groups {0, 1} and {2, 3} of 8 rows each.  With lane = (row, chunk) in
    // plain order all four k-groups met in one half, and groups g and g + 2 share LDS banks at the 40-byte pitch (2 x 640 B = 0 mod 256):
    // SQ_LDS_BANK_CONFLICT 1.55e9 -> 0.52e9 cycles per six calls at 200k x 200k rows, 7.36 -> 7.24 ms
    const int st_r = (tid >> 6) * 8 + ((lane >> 2) & 7), st_j = (lane >> 5) * 4 + (lane & 3);
#else
    const int st_r = tid / FN_CPR, st_j = tid % FN_CPR;     // r = 0 .. FN_RSTEP - 1
#endif
    auto st_dst = [&](int r) { return (r >> 4) * FN_SUB_BYTES + (st_j / FN_NM) * FN_GROUP + (r & 15) * FN_PITCH + (st_j % FN_NM) * 16; };
    struct Slot { uint4 v[FN_NCH]; float nv, nu, lw; };
    auto fetch = [&](int step, float lw, Slot &x) {
        x.lw = lw;
        const _Float16 *src = a.dbA + ((size_t)step * FN_STEP + st_r) * FN_K + st_j * 8;
#pragma unroll
        for (int c = 0; c < FN_NCH; c++) x.v[c] = *(const uint4 *)(src + (size_t)c * FN_RSTEP * FN_K);
        x.nv = tid < FN_STEP ? a.db_nlo[(size_t)step * FN_STEP + tid] : 0.0f;
        x.nu = tid < FN_STEP ? a.db_nup[(size_t)step * FN_STEP + tid] : 0.0f;
    };
    auto stash = [&](int buf, const Slot &x) {
#pragma unroll
        for (int c = 0; c < FN_NCH; c++) {
            unsigned char *d = &lds[buf][st_dst(st_r + c * FN_RSTEP)];
            *(uint2 *)d = make_uint2(x.v[c].x, x.v[c].y); *(uint2 *)(d + 8) = make_uint2(x.v[c].z, x.v[c].w);
        }
        // the accumulators start at -nlo / 2 (the BOUND_ONLY form: at -(nlo + nup) / 2, rounded towards the safe side with room for the
        // roundings of the accumulation): the MFMAs then leave acc = dot - nlo / 2 = -w / 2 and the test w < thr is acc > -thr / 2 -- one max3, one
        // max and one compare per block instead of four fused multiply-adds and three minimums
        if (tid < FN_STEP) { lnlo[buf][tid] = BOUND_ONLY ? -0.5f * (fn_add_up(x.nv, x.nu) * 1.0000006f) : -0.5f * x.nv; lnup[buf][tid] = x.nu; }
    };
    // The database does not fit the L2s (51 MB at 200k rows): a row block comes from the Infinity Cache or HBM, 1-2 us away.  Steps are
    // staged FN_GRP at a time with ONE workgroup barrier per group: the LDS image is double buffered by group, the global loads of group
    // g + 2 are issued once group g + 1 has been written to LDS and land while group g + 1 is computed on.  Round 4 measured where a
    // wavefront's cycles go in this loop (tools/build_variant.sh stamps pcr_featnn -DFN_STAMPS, 200k x 200k rows, per direction): barrier
    // waits 37 %, MFMA path 36 %, candidate path 10 %, LDS stash 10 %, load issue 7 % -- and then that the barriers are NOT the lever: groups
    // of 2 / 4 tiles per barrier (13.0 / 12.9 ms against 12.6 for both directions) and workgroups of 4 / 2 wavefronts (12.1 / 12.0 against
    // 13.0 before the other changes of the round) leave the time where it is.  A wavefront computes in 62 % of the staged steps of its
    // workgroup, its SIMD partner likewise, and whichever way the waiting is cut the workgroup lasts as long as its busiest wavefront:
    // the matrix pipes idle because the work of a step cannot move between wavefronts (the B operands of a wavefront's 64 queries live
    // in its registers).  FN_GRP = 1 is the form kept.  (Those figures are of the K = 128 kernel with workgroups of 8: 41 KB of LDS, 184 VGPRs.  The
    // K = 64 form -- 32 KB of LDS, 123 VGPRs, four workgroups of 4 wavefronts per CU -- fills the stalls of one workgroup with the others; its own
    // phase shares are in DESIGN.md section 8.)
    Slot slot[FN_GRP];
    float lw_now[FN_GRP], lw_next[FN_GRP];
    const int n_groups = (n_list + FN_GRP - 1) / FN_GRP;
    const bool lw_live = prune && !a.pre_mode;             // (a query tile beyond n_qt got +inf in slw: never computed)
    auto list_lw = [&](int i) { return lw_live ? slw[wv][i] : 0.0f; };
#pragma unroll
    for (int t = 0; t < FN_GRP; t++) { lw_now[t] = 0.0f; lw_next[t] = 0.0f; if (t < n_list) { fetch(slist[t], list_lw(t), slot[t]); } }
#pragma unroll
    for (int t = 0; t < FN_GRP; t++) if (t < n_list) { stash(t, slot[t]); lw_now[t] = slot[t].lw; }
#pragma unroll
    for (int t = 0; t < FN_GRP; t++) if (FN_GRP + t < n_list) fetch(slist[FN_GRP + t], list_lw(FN_GRP + t), slot[t]);
    // the list entries of the group after next are read from LDS one iteration ahead of their fetch (no LDS round trip in front of the global loads)
    int pf_step[FN_GRP]; float pf_lw[FN_GRP];
#pragma unroll
    for (int t = 0; t < FN_GRP; t++) { const int i = 2 * FN_GRP + t; pf_step[t] = i < n_list ? slist[i] : 0; pf_lw[t] = i < n_list ? list_lw(i) : 0.0f; }
    unsigned long long n_comp = 0;
#ifdef FN_STAMPS          // diagnostics build (tools/build_variant.sh ... -DFN_STAMPS): shader cycles of a wavefront per phase of the step loop
    unsigned long long tc_bar = 0, tc_fast = 0, tc_cand = 0, tc_stash = 0, tc_fetch = 0, n_hitblk = 0;
#define FN_T(...) __VA_ARGS__
#else
#define FN_T(...)
#endif
    const unsigned a_off = (unsigned)(g * FN_GROUP + col * FN_PITCH);
    int chunk_base = -1, chunk_fill = FN_CHUNK;            // wave-uniform: current chunk of the record pool (none yet)
    bool dead = false;                                      // the pool overflowed: flags[0] is set and the caller falls back
    // one staged tile: LDS buffer `buf`, database step `step`, box distance `lw` of the tile from this wavefront's queries
    auto tile = [&](const int buf, const int step, const float lw) {
        FN_T(const unsigned long long t2 = __builtin_amdgcn_s_memtime();)
        const bool wave_on = !(lw > Dw);                  // wave-uniform
        if (wave_on) {
        n_comp++;
        // ---- fast path: 4 tiles x 4 query blocks, one hit bit per 16 x 16 block, nothing else kept
        unsigned wave_hits = 0;                           // bit (sub, b): some lane saw one of its four rows of the block under its threshold (scalar register)
#pragma unroll
        for (int sub = 0; sub < FN_SUBS; sub++) {
            // A operands: 4 k-chunks x 16 bytes per lane, 8-byte LDS reads on the conflict-free image (plain loads: the compiler keeps
            // counted waits for them and runs the next tile's reads under this tile's MFMAs)
            union { uint2 u[2]; fn_h8 h; } av[FN_NM];
            const unsigned char *base = &lds[buf][sub * FN_SUB_BYTES + a_off];
#pragma unroll
            for (int m = 0; m < FN_NM; m++) { av[m].u[0] = *(const uint2 *)(base + m * 16); av[m].u[1] = *(const uint2 *)(base + m * 16 + 8); }
            const fn_f4 nl = *(const fn_f4 *)(&lnlo[buf][sub * 16 + 4 * g]);       // rows 4 g .. 4 g + 3 of the tile: this lane's accumulator rows
#pragma unroll
            for (int b = 0; b < FN_QB; b++) {
                fn_f4 acc = nl;
#pragma unroll
                for (int m = 0; m < FN_NM; m++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[m].h, qb[b][m], acc, 0, 0, 0);
                if (BOUND_ONLY) {
                    const float mx = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));    // = -(smallest w of the four rows) / 2
                    U[b] = fminf(U[b], -2.0f * mx);       // upper bounds d~ + E_b - nq of the four rows
                } else {    // four compares into scalar masks: no per-lane state, the union over the wavefront costs nothing
                    const unsigned long long m = __ballot(acc[0] > thrh[b]) | __ballot(acc[1] > thrh[b]) | __ballot(acc[2] > thrh[b]) | __ballot(acc[3] > thrh[b]);
                    wave_hits |= m != 0ull ? (1u << (sub * FN_QB + b)) : 0u;
                }
            }
        }
        FN_T(const unsigned long long t3 = __builtin_amdgcn_s_memtime(); tc_fast += t3 - t2;)
        // ---- candidate path: blocks in which some lane saw a value under its threshold are recomputed from the LDS image
        if (!BOUND_ONLY && wave_hits != 0u && !dead) {
            while (wave_hits != 0u) {
                FN_T(n_hitblk++;)
                const int blk = __builtin_ctz(wave_hits);
                wave_hits &= wave_hits - 1u;
                const int sub = blk / FN_QB, b = blk % FN_QB;
                union { uint2 u[2]; fn_h8 h; } av[FN_NM];
                const unsigned char *base = &lds[buf][sub * FN_SUB_BYTES + a_off];
#pragma unroll
                for (int m = 0; m < FN_NM; m++) { av[m].u[0] = *(const uint2 *)(base + m * 16); av[m].u[1] = *(const uint2 *)(base + m * 16 + 8); }
                const fn_f4 nl = *(const fn_f4 *)(&lnlo[buf][sub * 16 + 4 * g]);
                const fn_f4 nu = *(const fn_f4 *)(&lnup[buf][sub * 16 + 4 * g]);
                // block b is a run-time index here: select its operands and state with uniform branches (registers stay registers)
                fn_f4 acc = nl;
                float Ub = 0.0f, cqb = 0.0f;
#pragma unroll
                for (int bb = 0; bb < FN_QB; bb++) if (b == bb) {
#pragma unroll
                    for (int m = 0; m < FN_NM; m++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[m].h, qb[bb][m], acc, 0, 0, 0);
                    Ub = U[bb]; cqb = cq[bb];
                }
                float tb = Ub + cqb;
                const int q = q0 + 16 * b + col;
                const int row0 = step * FN_STEP + sub * 16 + 4 * g;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float w = -2.0f * acc[i];
                    const bool h = w < tb;
                    const unsigned long long mask = __ballot(h);
                    const int n_new = __builtin_popcountll(mask);
                    if (n_new) {
                        if (chunk_fill + n_new > FN_CHUNK) {          // wave-uniform: close the chunk, take a new one (one atomic per chunk)
                            int nb_ = 0;
                            if (lane == 0) {
                                if (chunk_base >= 0) a.chunk_fill[chunk_base / FN_CHUNK] = chunk_fill;
                                nb_ = atomicAdd(a.pool_used, FN_CHUNK);
                            }
                            chunk_base = __builtin_amdgcn_readfirstlane(nb_); chunk_fill = 0;
                            if (chunk_base + FN_CHUNK > a.pool_cap) { if (lane == 0) a.flags[0] = 1; dead = true; chunk_base = -1; chunk_fill = FN_CHUNK; }
                        }
                        if (!dead) {
                            if (h) {
                                const int slot_ = chunk_base + chunk_fill + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                                a.rec_q[slot_] = q; a.rec_row[slot_] = row0 + i; a.rec_w[slot_] = w;
                            }
                            chunk_fill += n_new;
                        }
                    }
                    if (h) { Ub = fminf(Ub, fn_add_up(w, nu[i])); tb = Ub + cqb; }
                }
                // share the improved bound between the 4 row groups of a query, then store it back into the block's state
                Ub = pcr_xrow_min(Ub);
#pragma unroll
                for (int bb = 0; bb < FN_QB; bb++) if (b == bb) { U[bb] = Ub; thrh[bb] = -0.5f * (Ub + cq[bb]); }
            }
            if (prune) Dw = wave_D();
        }
        FN_T(tc_cand += __builtin_amdgcn_s_memtime() - t3;)
        }
    };
    for (int gi = 0; gi < n_groups; gi++) {
        const int half = (gi & 1) * FN_GRP;
        FN_T(const unsigned long long t1 = __builtin_amdgcn_s_memtime();)
        __syncthreads();                                  // the buffers of this group are complete; the other group's are no longer read by anyone
        FN_T(tc_bar += __builtin_amdgcn_s_memtime() - t1;)
#pragma unroll
        for (int t = 0; t < FN_GRP; t++) if (gi * FN_GRP + t < n_list) tile(half + t, slist[gi * FN_GRP + t], lw_now[t]);
        FN_T(const unsigned long long t4 = __builtin_amdgcn_s_memtime();)
        if (gi + 1 < n_groups) {
#pragma unroll
            for (int t = 0; t < FN_GRP; t++) if ((gi + 1) * FN_GRP + t < n_list) { stash((FN_GRP - half) + t, slot[t]); lw_next[t] = slot[t].lw; }
        }
        FN_T(const unsigned long long t5 = __builtin_amdgcn_s_memtime(); tc_stash += t5 - t4;)
        if (gi + 2 < n_groups) {
#pragma unroll
            for (int t = 0; t < FN_GRP; t++) if ((gi + 2) * FN_GRP + t < n_list) fetch(pf_step[t], pf_lw[t], slot[t]);
        }
#pragma unroll
        for (int t = 0; t < FN_GRP; t++) { const int i = (gi + 3) * FN_GRP + t; pf_step[t] = i < n_list ? slist[i] : 0; pf_lw[t] = i < n_list ? list_lw(i) : 0.0f; }
        FN_T(tc_fetch += __builtin_amdgcn_s_memtime() - t5;)
#pragma unroll
        for (int t = 0; t < FN_GRP; t++) lw_now[t] = lw_next[t];
    }
    if (lane == 0 && chunk_base >= 0) a.chunk_fill[chunk_base / FN_CHUNK] = chunk_fill;
    if (a.stats && lane == 0) { atomicAdd(&a.stats[1], n_comp); atomicAdd(&a.stats[2], (unsigned long long)n_list); }
    FN_T(if (a.stats && lane == 0) { atomicAdd(&a.stats[4], tc_fetch); atomicAdd(&a.stats[5], tc_bar); atomicAdd(&a.stats[6], tc_fast); atomicAdd(&a.stats[7], tc_cand); atomicAdd(&a.stats[8], tc_stash); atomicAdd(&a.stats[9], n_hitblk); })
    if (BOUND_ONLY)
#pragma unroll
        for (int b = 0; b < FN_QB; b++) U[b] = pcr_xrow_min(U[b]);
    if (g == 0)
#pragma unroll
        for (int b = 0; b < FN_QB; b++) {
            const int q = q0 + 16 * b + col;
            if (q < a.n_q) atomicMin(&a.Ug[q], fn_ord(U[b]));
        }
}
// (waves_per_eu: with at most 128 / 168 registers per lane the compiler keeps the MFMA accumulators in ordinary VGPRs -- no v_accvgpr_read before every bound test)
#ifndef FN_WAVES_ATTR
#define FN_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
template <bool BOUND_ONLY> __global__ void __launch_bounds__(FN_WG) FN_WAVES_ATTR k_feature_nn_screen(FnnArgs a) { d_feature_nn_screen<BOUND_ONLY>(a); }
template <bool BOUND_ONLY> __global__ void __launch_bounds__(FN_WG) FN_WAVES_ATTR k_feature_nn_screen_g(const FnnArgs *a) { d_feature_nn_screen<BOUND_ONLY>(a[blockIdx.z]); }

// ---- exact float64 distance exactly as the oracle's kd-tree leaf loop computes it: d2 += e * e with the product and the sum
// rounded separately, k = 0 .. 32 in order
__device__ static inline double fn_exact_d2(const float *__restrict__ x, const float *__restrict__ y) {
    double d2 = 0.0;
#pragma unroll
    for (int k = 0; k < FN_D; k++) { const double e = (double)x[k] - (double)y[k]; d2 = __dadd_rn(d2, __dmul_rn(e, e)); }
    return d2;
}

struct FnxArgs {
    const float *q; const float *q_nrm, *q_cq; int n_q;  // original float32 query rows; scaled centred norms (sign bit: zero row), 2 E_q
    const float *db; int n_db; const float *db_nlo, *db_nup, *db_nrm; const int *db_first_zero;
    const int *Ug;                                      // final bound of the screen per query
    const int *pool_used; const int *chunk_fill; const int *rec_q; const int *rec_row; const float *rec_w;
    const uint32_t *perm_q, *perm_db;                   // tile-pruned screen: query q / row r of the screen is row perm[.] of q / db (null: identity)
    double *rec_d;                                      // per record: its exact distance, written by pass 1 and read by pass 2 (no second gather of the rows)
    unsigned long long *best_d;                         // per (screen) query: bits of the smallest exact distance (non-negative doubles order like their bits)
    int32_t *out;                                       // per ORIGINAL query: smallest ORIGINAL row attaining it (INT_MAX until found)
    float *dbg;                                         // PCR_FEATNN_CHECK: [0] max |d~ - d| / (nq + nb) seen on records, [1] records that survived the final bound
};
__device__ static inline bool fn_record(const FnxArgs &a, int r, int *q, int *row) {
    if (r >= *a.pool_used || (r % FN_CHUNK) >= a.chunk_fill[r / FN_CHUNK]) return false;
    *q = a.rec_q[r]; *row = a.rec_row[r];
    if (*q >= a.n_q || *row >= a.n_db) return false;
    const float thr = fn_unord(a.Ug[*q]) + a.q_cq[*q];
    return a.rec_w[r] <= thr;                             // else: could not be the minimum given the final bound
}
// pass 1: smallest exact distance per query
__device__ static inline void d_fn_exact_min(const FnxArgs &a, int n_rec_cap, const int r) {
    int q, row;
    if (r >= n_rec_cap || !fn_record(a, r, &q, &row)) return;
    const size_t oq = a.perm_q ? a.perm_q[q] : (size_t)q, orow = a.perm_db ? a.perm_db[row] : (size_t)row;
    const double d = fn_exact_d2(a.q + oq * FN_D, a.db + orow * FN_D);
    a.rec_d[r] = d;
    atomicMin(&a.best_d[q], (unsigned long long)__double_as_longlong(d));
    if (a.dbg) {          // the screen's value of this pair against the exact one, in units of the bound
        const double nq = fabs((double)a.q_nrm[q]), nb = fabs((double)a.db_nrm[row]);
        const double dt = (double)a.rec_w[r] - (double)a.db_nlo[row] + nb + nq;   // d~ in scaled units (to the rounding of nrm)
        const double budget = 0.5 * ((double)a.q_cq[q] + (double)a.db_nup[row]);  // E_q + E_b
        atomicMax((unsigned int *)&a.dbg[0], __float_as_uint((float)(fabs(dt - d * FN_SCALE * FN_SCALE) / budget)));
        atomicAdd(&a.dbg[1], 1.0f);
    }
}
// pass 2: smallest row among the records that attain it (exact ties -> smaller index, as the oracle's heap orders them)
// (a fixed grid strides over the USED part of the pool -- a fifth of its capacity; one thread per slot of the capacity was 80 000 workgroups that exit at once)
__global__ void __launch_bounds__(256) k_fn_exact_min(FnxArgs a, int n_rec_cap) {
    const int used = *a.pool_used < n_rec_cap ? *a.pool_used : n_rec_cap;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < used; r += gridDim.x * 256) d_fn_exact_min(a, n_rec_cap, r);
}
__device__ static inline void d_fn_exact_arg(const FnxArgs &a, int n_rec_cap, const int r) {
    int q, row;
    if (r >= n_rec_cap || !fn_record(a, r, &q, &row)) return;
    if ((unsigned long long)__double_as_longlong(a.rec_d[r]) == a.best_d[q]) atomicMin(&a.out[a.perm_q ? a.perm_q[q] : q], (int)(a.perm_db ? a.perm_db[row] : row));
}
__global__ void __launch_bounds__(256) k_fn_exact_arg(FnxArgs a, int n_rec_cap) {
    const int used = *a.pool_used < n_rec_cap ? *a.pool_used : n_rec_cap;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < used; r += gridDim.x * 256) d_fn_exact_arg(a, n_rec_cap, r);
}
// pass 3: zero queries take the first zero row; a query without any record (empty database) gets -1
__device__ static inline void d_fn_finish(const FnxArgs &a) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.n_q) return;
    const int oq = a.perm_q ? (int)a.perm_q[q] : q;
    if (__builtin_signbit(a.q_nrm[q]) && *a.db_first_zero != 0x7fffffff) a.out[oq] = *a.db_first_zero;
    else if (a.out[oq] == 0x7fffffff) a.out[oq] = -1;
}
__global__ void __launch_bounds__(256) k_fn_finish(FnxArgs a) { d_fn_finish(a); }
struct FnxDesc { FnxArgs x; int n_rec_cap; };
// (batch forms: a fixed grid strides over the USED part of a problem's pool -- a fifth of its capacity -- instead of one thread per slot of the capacity)
__global__ void __launch_bounds__(256) k_fn_exact_min_g(const FnxDesc *d) {
    const FnxDesc &a = d[blockIdx.y];
    const int used = *a.x.pool_used < a.n_rec_cap ? *a.x.pool_used : a.n_rec_cap;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < used; r += gridDim.x * 256) d_fn_exact_min(a.x, a.n_rec_cap, r);
}
__global__ void __launch_bounds__(256) k_fn_exact_arg_g(const FnxDesc *d) {
    const FnxDesc &a = d[blockIdx.y];
    const int used = *a.x.pool_used < a.n_rec_cap ? *a.x.pool_used : a.n_rec_cap;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < used; r += gridDim.x * 256) d_fn_exact_arg(a.x, a.n_rec_cap, r);
}
__global__ void __launch_bounds__(256) k_fn_finish_g(const FnxDesc *d) { const FnxDesc &a = d[blockIdx.y]; if ((int)blockIdx.x * 256 < a.x.n_q) d_fn_finish(a.x); }

// one launch instead of five memsets per direction (registro_FGR on NCLT-size clouds is bound by the rate of small dependent launches)
__device__ static inline void d_fn_init(int *pool_used, int *chunk_fill, int n_chunks, int *Ug, int n_ug, unsigned long long *best_d, int32_t *out, int n_q,
                                        int *first_zero, int *flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { *pool_used = 0; if (first_zero) { first_zero[0] = 0x7fffffff; first_zero[1] = 0x7fffffff; flags[0] = 0; flags[1] = 0; } }
    if (i < n_chunks) chunk_fill[i] = 0;
    if (i < n_ug) Ug[i] = 0x7f800000;                      // +inf
    if (i < n_q) { best_d[i] = ~0ull; if (out) out[i] = 0x7fffffff; }
}
__global__ void __launch_bounds__(256) k_fn_init(int *pool_used, int *chunk_fill, int n_chunks, int *Ug, int n_ug, unsigned long long *best_d, int32_t *out, int n_q,
                                                 int *first_zero, int *flags) { d_fn_init(pool_used, chunk_fill, n_chunks, Ug, n_ug, best_d, out, n_q, first_zero, flags); }
struct FnInitDesc { int *pool_used, *chunk_fill; int n_chunks; int *Ug; int n_ug; unsigned long long *best_d; int32_t *out; int n_q; };
__global__ void __launch_bounds__(256) k_fn_init_g(const FnInitDesc *d) { const FnInitDesc a = d[blockIdx.y]; d_fn_init(a.pool_used, a.chunk_fill, a.n_chunks, a.Ug, a.n_ug, a.best_d, a.out, a.n_q, nullptr, nullptr); }

// ---- mutual search, second direction (round 5).  FGR keeps a pair (i, j) only if j is the nearest row of i AND i the nearest row of j
// (cross check, pcr_fgr.hip k_cross_flags), so once the first direction has given every j its nearest i, the second direction matters only
// for the rows i that some j points at, and for those an upper bound of the answer is known: the exact distance of that j.  k_fn_seed writes
// it into the bound array -- U = d* - |x_i|^2 in the screen's units, rounded up: every row at distance <= d* (the true nearest row of i and
// its exact ties) passes the candidate test L(b) < U + 2 E_q, since L(b) <= d_b - |x_i|^2 + E_q -- and the screen starts with its FINAL bounds:
// no pre-passes, and the tile lists keep only the tiles within those bounds.  Rows nobody points at stay at +inf and are dead.
struct FnSeedArgs {
    const int32_t *out_prev; const unsigned long long *best_prev; const uint32_t *perm_prev; int n_prev;      // first direction: answers (original indices), exact distances (screen order)
    const uint32_t *inv_cur; const float *q_nrm_cur; int *Ug;                                                 // second direction: original row -> screen query, |x|^2, bounds
};
__global__ void __launch_bounds__(256) k_fn_invperm(const uint32_t *__restrict__ perm, int n, uint32_t *__restrict__ inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[perm[i]] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) k_fn_seed(FnSeedArgs a) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_prev) return;
    const int oi = a.out_prev[a.perm_prev ? a.perm_prev[s] : (uint32_t)s];
    if (oi < 0) return;
    const unsigned long long bits = a.best_prev[s];
    const double d = bits == ~0ull ? 0.0 : __longlong_as_double((long long)bits);      // (no record but an answer: an all-zero row matched to the first all-zero row)
    const int pos = a.inv_cur ? (int)a.inv_cur[oi] : oi;
    const double nq = fabs((double)a.q_nrm_cur[pos]), ds = d * (FN_SCALE * FN_SCALE);
    const float u = (float)(ds - nq + 2.0e-6 * (ds + nq) + 1.0e-6);
    atomicMin(&a.Ug[pos], fn_ord(fn_add_up(u, 0.0f)));
}

// The live rows of the second direction as a query set of their own (same order): B operands, norms, bounds, the map to the original row.
struct FnCompactArgs {
    const int *seedU; const uint8_t *live; const int *cpos; int n;        // per screen position of the cloud: seeded bound, live flag, position among the live rows
    const _Float16 *B; const float *nrm, *cq; const uint32_t *perm;       // the full query set (perm null: identity)
    _Float16 *Bc; float *nrm_c, *cq_c; uint32_t *perm_c; int *Ug_c; int32_t *out;
};
__global__ void __launch_bounds__(256) k_fn_live(const int *__restrict__ seedU, int n, uint8_t *__restrict__ live) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) live[i] = seedU[i] != 0x7f800000 ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_fn_compact(FnCompactArgs a) {      // 8 lanes per row: 16 bytes of the 128-byte B row each
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (i >= a.n || !a.live[i]) return;
    const int c = a.cpos[i];
    static_assert(FN_K * sizeof(_Float16) == 128, "k_fn_compact copies 128-byte rows");
    reinterpret_cast<uint4 *>(a.Bc + (size_t)c * FN_K)[part] = reinterpret_cast<const uint4 *>(a.B + (size_t)i * FN_K)[part];
    if (part == 0) {
        const uint32_t o = a.perm ? a.perm[i] : (uint32_t)i;
        a.nrm_c[c] = a.nrm[i]; a.cq_c[c] = a.cq[i]; a.perm_c[c] = o; a.Ug_c[c] = a.seedU[i]; a.out[o] = 0x7fffffff;
    }
}

// batch forms (lockstep FGR groups): blockIdx.y = pair
struct FnLiveDesc { const int *seedU; int n; uint8_t *live; int32_t *out; };
__global__ void __launch_bounds__(256) k_fn_seed_g(const FnSeedArgs *d) {
    const FnSeedArgs a = d[blockIdx.y];
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_prev) return;
    const int oi = a.out_prev[a.perm_prev ? a.perm_prev[s] : (uint32_t)s];
    if (oi < 0) return;
    const unsigned long long bits = a.best_prev[s];
    const double d2 = bits == ~0ull ? 0.0 : __longlong_as_double((long long)bits);
    const int pos = a.inv_cur ? (int)a.inv_cur[oi] : oi;
    const double nq = fabs((double)a.q_nrm_cur[pos]), ds = d2 * (FN_SCALE * FN_SCALE);
    const float u = (float)(ds - nq + 2.0e-6 * (ds + nq) + 1.0e-6);
    atomicMin(&a.Ug[pos], fn_ord(fn_add_up(u, 0.0f)));
}
__global__ void __launch_bounds__(256) k_fn_live_g(const FnLiveDesc *d) {     // live flags; the second direction's answers start at -1 (no mutual match possible)
    const FnLiveDesc a = d[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) { a.live[i] = a.seedU[i] != 0x7f800000 ? 1 : 0; a.out[i] = -1; }
}
__global__ void __launch_bounds__(256) k_fn_compact_g(const FnCompactArgs *d) {
    const FnCompactArgs a = d[blockIdx.y];
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (i >= a.n || !a.live[i]) return;
    const int c = a.cpos[i];
    reinterpret_cast<uint4 *>(a.Bc + (size_t)c * FN_K)[part] = reinterpret_cast<const uint4 *>(a.B + (size_t)i * FN_K)[part];
    if (part == 0) {
        const uint32_t o = a.perm ? a.perm[i] : (uint32_t)i;
        a.nrm_c[c] = a.nrm[i]; a.cq_c[c] = a.cq[i]; a.perm_c[c] = o; a.Ug_c[c] = a.seedU[i]; a.out[o] = 0x7fffffff;
    }
}

// ------------------------------------------------------------------------------------------------------------------ driver
size_t pcr_feature_nn_scratch_bytes(int64_t n0, int64_t n1, int prune_mode) {
    const size_t p0 = ((size_t)n0 + FN_QPG) / FN_QPG * FN_QPG, p1 = ((size_t)n1 + FN_QPG) / FN_QPG * FN_QPG;
    const size_t per_row = 2 * FN_K * sizeof(_Float16) + 4 * sizeof(float);                 // both forms + the four floats of FnRows
    const size_t per_query = (size_t)FN_POOL_PER_QUERY * 21 + 8 + 4 + 16;                   // records (+ exact distances), best distance, bound, chunk table share
    // tile pruning: principal coordinates, keys / values (in, out) and sort scratch per row; boxes; the (query tile x row tile) bound matrix
    const size_t prune_rows = (p0 + p1) * (FN_D * sizeof(float) + 2 * (8 + 4)) + pcr_sort_temp_bytes(p0 > p1 ? p0 : p1) + (p0 + p1) / 64 * (2 * FN_D * 4 + 64);
    const size_t lmat = (p0 / 64 + 64) * (p1 / 64 + 64) * sizeof(float);
    // chunks the wavefronts of the screen may leave partly filled: workgroups of the widest records pass + those of the pre-pass.  Only pairs the
    // size rule prunes (or a forced PCR_FEATNN_PRUNE / debug mode) cut the database into FN_WGS_PRUNED workgroups; the others into ~2048-3072
    static const bool forced = getenv("PCR_FEATNN_PRUNE") && atoi(getenv("PCR_FEATNN_PRUNE")) > 0;
    const size_t groups = (p0 > p1 ? p0 : p1) / FN_QPG, steps = (p0 > p1 ? p0 : p1) / FN_STEP;
    const bool may_prune = forced || prune_mode > 0 || (double)n0 * (double)n1 >= 5.0e9;
    const size_t wide = may_prune ? std::min<size_t>(FN_WGS_PRUNED + groups, groups * std::min<size_t>(256, steps ? steps : 1)) : 3072 + groups;
    const size_t waves = (wide + (2 + FN_PRE_SPLIT) * groups + 512) * (FN_WG / 64);
    return (p0 + p1) * (per_row + 12 + 160) + (p0 > p1 ? p0 : p1) * per_query + waves * FN_CHUNK * 21 + (1u << 22) + prune_rows
           + (lmat <= ((size_t)512 << 20) ? lmat : 0) + (1u << 20);
}

// For every row of cloud 1 its exact nearest row of cloud 0 (out_1to0, n1 entries) and vice versa (out_0to1, n0 entries).
// f0 / f1: device float32 (n x 33).  Scratch from the arena above the current mark.  PCR_ECAPACITY: feature values outside the
// f16 range or record pool exhausted -- the caller takes the all-pairs float64 path.
int pcr_feature_nn_mutual(pcr_context *ctx, const float *f0, int n0, const float *f1, int n1, int32_t *out_1to0, int32_t *out_0to1, int prune_mode, int mutual_only) {
    if (n0 <= 0 || n1 <= 0) return PCR_OK;
    if (!pcr_options().featnn_mutual.load(std::memory_order_relaxed)) mutual_only = 0;      // (test switch: both directions in full)
    ArenaMark mark(ctx);
    const float *f[2] = {f0, f1}; const int n[2] = {n0, n1}; int np[2];
    _Float16 *A[2], *B[2]; FnRows rows[2];
    static const bool check = getenv("PCR_FEATNN_CHECK") != nullptr;
    const int nbm = 64;
    double *part = arena<double>(ctx, (size_t)2 * nbm * FN_PC + FN_PC);
    if (!part) return PCR_ENOMEM;
    double *mu = part + (size_t)2 * nbm * FN_PC;
    PCR_LAUNCH(ctx, k_fn_colsum, dim3(nbm), dim3(256), 0, ctx->stream, f0, n0, part);
    PCR_LAUNCH(ctx, k_fn_colsum, dim3(nbm), dim3(256), 0, ctx->stream, f1, n1, part + (size_t)nbm * FN_PC);
    int *first_zero = arena<int>(ctx, 2), *flags = arena<int>(ctx, 2);
    if (!first_zero || !flags) return PCR_ENOMEM;
    PCR_LAUNCH(ctx, k_fn_mean, dim3(1), dim3(128), 0, ctx->stream, part, nbm, n0, mu, first_zero, flags);
    for (int c = 0; c < 2; c++) np[c] = (n[c] + FN_QPG - 1) / FN_QPG * FN_QPG;   // multiple of 512 (queries per workgroup) and of 64 (rows per step)
    // ---- tile pruning: worth its set-up (two small sorts, a host eigen-decomposition) from ~70k x 70k rows (60k x 60k: 4.1 ms with, 3.5 ms
    // without; 80k: 4.6 / 5.4; 100k: 5.6 / 7.5; 200k x 200k: 12.1 / 25.9 ms); the bound matrix must fit
    static const int prune_env_ = getenv("PCR_FEATNN_PRUNE") ? atoi(getenv("PCR_FEATNN_PRUNE")) : -1;
    const int prune_env = prune_mode >= 0 ? prune_mode : prune_env_;
    const size_t lmat_bytes = (size_t)(np[0] / 64 + 64) * (size_t)(np[1] / 64 + 64) * sizeof(float);
    const bool prune = prune_env == 0 ? false : ((prune_env > 0 || (double)n0 * (double)n1 >= 5.0e9) && lmat_bytes <= ((size_t)512 << 20) && n0 >= 64 && n1 >= 64
                                                   && np[0] / 64 < 65536 && np[1] / 64 < 65536);        // query tiles are blockIdx.y of k_fn_boxlb
    const int nbg = 128;
    double *gpart = nullptr, *gsum = nullptr;
    std::vector<double> hcs((size_t)2 * nbm * FN_PC), hg((size_t)2 * FN_NG);
    double hmu[FN_PC];
    if (prune) {   // raw second moments of both feature matrices (the principal axes come from them on the host)
        gpart = arena<double>(ctx, (size_t)2 * nbg * FN_NG + 2 * FN_NG);
        if (!gpart) return PCR_ENOMEM;
        gsum = gpart + (size_t)2 * nbg * FN_NG;
        for (int c = 0; c < 2; c++) {
            PCR_LAUNCH(ctx, k_fn_gram, dim3(nbg), dim3(576), 0, ctx->stream, f[c], n[c], gpart + (size_t)c * nbg * FN_NG);
            PCR_LAUNCH(ctx, k_fn_gram_final, dim3(1), dim3(576), 0, ctx->stream, gpart + (size_t)c * nbg * FN_NG, nbg, gsum + (size_t)c * FN_NG);
        }
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(hg.data(), gsum, sizeof(double) * hg.size(), hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(hcs.data(), part, sizeof(double) * hcs.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    {   // the f16 split holds |f - mu| * 128 < 65504: true for FPFH (bins <= 200); anything else takes the float64 path
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(hmu, mu, sizeof hmu, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        double mm = 0; for (int k = 0; k < FN_D; k++) mm = fmax(mm, fabs(hmu[k]));
        if (!((hmu[FN_D] + mm) * FN_SCALE < 60000.0)) return PCR_ECAPACITY;
    }
    uint32_t *perm[2] = {nullptr, nullptr};
    float *Pc[2] = {nullptr, nullptr};                  // principal coordinates (original row order)
    float *blo[2] = {nullptr, nullptr}, *bhi[2] = {nullptr, nullptr};
    int tile_stride[2] = {0, 0};
    float eps2 = 0.0f;
    if (prune) {
        // principal axes of the centred features of both clouds together: 33 x 33 Jacobi on the host
        const double N = (double)n0 + (double)n1;
        double Ng = 0.0;                 // rows that entered the second moments: blocks of 32 rows, block b taken iff (b / nbg) % FN_GRAM_SKIP == 0
        for (int c = 0; c < 2; c++) for (long long b = 0; b * 32 < n[c]; b++) if ((b / nbg) % FN_GRAM_SKIP == 0) Ng += (double)std::min<long long>(32, n[c] - b * 32);
        double m[FN_D];
        for (int k = 0; k < FN_D; k++) { double v = 0; for (int b = 0; b < 2 * nbm; b++) v += hcs[(size_t)b * FN_PC + k]; m[k] = v / N; }
        std::vector<double> C((size_t)FN_D * FN_D), V((size_t)FN_D * FN_D);
        double ev[FN_D];
        {
            int t = 0;
            for (int a_ = 0; a_ < FN_D; a_++)
                for (int b_ = a_; b_ < FN_D; b_++, t++) { const double v = (hg[t] + hg[FN_NG + t]) / Ng - m[a_] * m[b_]; C[(size_t)a_ * FN_D + b_] = v; C[(size_t)b_ * FN_D + a_] = v; }
        }
        fn_jacobi(C.data(), V.data(), ev);
        double hrot[FN_D * FN_D + 2 * FN_MD];
        std::memcpy(hrot, V.data(), sizeof(double) * FN_D * FN_D);
        for (int k = 0; k < FN_MD; k++) {
            // key range: mean of the coordinate (the split centres on the FIRST cloud's mean, the axes on the common one) +- 3 sigma
            double centre = 0.0;
            for (int j = 0; j < FN_D; j++) centre += V[(size_t)j * FN_D + k] * (m[j] - hmu[j]) * FN_SCALE;
            const double sg = std::sqrt(ev[k] > 0 ? ev[k] : 0.0) * FN_SCALE;
            hrot[FN_D * FN_D + 2 * k] = centre - 3.0 * sg;
            hrot[FN_D * FN_D + 2 * k + 1] = sg > 0 ? (double)(1 << FN_MB) / (6.0 * sg) : 0.0;
        }
        const double *rot = pcr_desc_upload<double>(ctx, hrot, FN_D * FN_D + 2 * FN_MD);              // through the pinned ring: no second wait
        if (!rot) return PCR_EHIP;
        // float32 rounding of a stored coordinate: |P| <= |x| <= sqrt(33) * largest |x_k|
        double mm = 0; for (int k = 0; k < FN_D; k++) mm = fmax(mm, fabs(hmu[k]));
        eps2 = (float)(2.0 * 1.2e-7 * std::sqrt((double)FN_D) * (hmu[FN_D] + mm) * FN_SCALE * 1.01);
        for (int c = 0; c < 2; c++) {
            const int nt = np[c] / 64;
            tile_stride[c] = nt;
            float *P = arena<float>(ctx, (size_t)n[c] * FN_D);
            uint64_t *k_in = arena<uint64_t>(ctx, n[c]), *k_out = arena<uint64_t>(ctx, n[c]);
            uint32_t *v_in = arena<uint32_t>(ctx, n[c]);
            perm[c] = arena<uint32_t>(ctx, n[c]);
            const size_t tb = pcr_sort_temp_bytes(n[c]);
            void *temp = arena<char>(ctx, tb);
            blo[c] = arena<float>(ctx, (size_t)FN_D * nt); bhi[c] = arena<float>(ctx, (size_t)FN_D * nt);
            if (!P || !k_in || !k_out || !v_in || !perm[c] || !temp || !blo[c] || !bhi[c]) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fn_rotate, dim3((n[c] + 255) / 256), dim3(256), 0, ctx->stream, f[c], n[c], mu, rot, P, k_in, v_in);
            PCR_TRY(pcr_sort_pairs(ctx, temp, tb, k_in, k_out, v_in, perm[c], (size_t)n[c], FN_MD * FN_MB));
            PCR_LAUNCH(ctx, k_fn_boxes, dim3((nt + 3) / 4), dim3(256), 0, ctx->stream, P, perm[c], n[c], nt, nt, blo[c], bhi[c]);
            Pc[c] = P;
        }
    }
    for (int c = 0; c < 2; c++) {
        A[c] = arena<_Float16>(ctx, (size_t)np[c] * FN_K); B[c] = arena<_Float16>(ctx, (size_t)np[c] * FN_K);
        float *r4 = arena<float>(ctx, (size_t)4 * np[c]);
        if (!A[c] || !B[c] || !r4) return PCR_ENOMEM;
        rows[c] = FnRows{r4, r4 + np[c], r4 + (size_t)2 * np[c], r4 + (size_t)3 * np[c]};
        PCR_LAUNCH(ctx, k_fn_split, dim3((unsigned)(((size_t)np[c] * FN_SPLIT_LANES + 255) / 256)), dim3(256), 0, ctx->stream, f[c], n[c], np[c], mu, A[c], B[c], rows[c], first_zero + c, (const uint32_t *)perm[c]);
    }
    float *dbg = nullptr;
    if (check) { dbg = arena<float>(ctx, 2); if (!dbg) return PCR_ENOMEM; }
    // mutual_only: out_0to1 is complete only where it can matter to the cross check (-1 elsewhere); the first direction's exact distances stay
    unsigned long long *best_first = nullptr; uint32_t *inv0 = nullptr;
    if (mutual_only) {
        best_first = arena<unsigned long long>(ctx, n[1]);
        if (!best_first) return PCR_ENOMEM;
        if (perm[0]) {
            inv0 = arena<uint32_t>(ctx, n[0]);
            if (!inv0) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fn_invperm, dim3((n[0] + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t *)perm[0], n[0], inv0);
        }
    }
    for (int dir = 0; dir < 2; dir++) {
        const bool seeded = mutual_only && dir == 1;
        const int qc = dir == 0 ? 1 : 0, dc = 1 - qc;                          // dir 0: queries = cloud 1, database = cloud 0
        int32_t *out = dir == 0 ? out_1to0 : out_0to1;
        ArenaMark m2(ctx);
        int nq = n[qc], nqp = np[qc];
        const int steps = np[dc] / FN_STEP;
        // ---- seeded direction: the live rows (some row of the other cloud points at them) become a query set of their own
        const _Float16 *qB = B[qc]; const float *q_nrm = rows[qc].nrm, *q_cq = rows[qc].cq; const uint32_t *q_perm = perm[qc];
        const float *q_blo = blo[qc], *q_bhi = bhi[qc]; int q_tile_stride = tile_stride[qc];
        int *Ug_seeded = nullptr;
        if (seeded) {
            int *seedU = arena<int>(ctx, np[0]); uint8_t *live = arena<uint8_t>(ctx, n[0]); int *cpos = arena<int>(ctx, n[0]), *n_live_dev = arena<int>(ctx, 1);
            if (!seedU || !live || !cpos || !n_live_dev) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fn_init, dim3((np[0] + 255) / 256), dim3(256), 0, ctx->stream, n_live_dev, (int *)nullptr, 0, seedU, np[0], (unsigned long long *)nullptr, (int32_t *)nullptr, 0,
                       (int *)nullptr, (int *)nullptr);
            FnSeedArgs sa;
            sa.out_prev = out_1to0; sa.best_prev = best_first; sa.perm_prev = perm[1]; sa.n_prev = n[1]; sa.inv_cur = inv0; sa.q_nrm_cur = rows[0].nrm; sa.Ug = seedU;
            PCR_LAUNCH(ctx, k_fn_seed, dim3((n[1] + 255) / 256), dim3(256), 0, ctx->stream, sa);
            PCR_LAUNCH(ctx, k_fn_live, dim3((n[0] + 255) / 256), dim3(256), 0, ctx->stream, (const int *)seedU, n[0], live);
            PCR_TRY(pcr_dev_flag_scan(ctx, live, nullptr, n[0], cpos, n_live_dev));
            PCR_HIP_CHECK(ctx, hipMemsetAsync(out, 0xff, sizeof(int32_t) * (size_t)n[0], ctx->stream));          // -1: no mutual match possible
            int64_t n_live = 0;
            PCR_TRY(pcr_read_count(ctx, n_live_dev, &n_live));
            if (n_live == 0) continue;
            nq = (int)n_live; nqp = (nq + FN_QPG - 1) / FN_QPG * FN_QPG;
            _Float16 *Bc = arena<_Float16>(ctx, (size_t)nqp * FN_K); float *r2c = arena<float>(ctx, (size_t)2 * nqp); uint32_t *perm_c = arena<uint32_t>(ctx, nqp);
            Ug_seeded = arena<int>(ctx, nqp);
            if (!Bc || !r2c || !perm_c || !Ug_seeded) return PCR_ENOMEM;
            PCR_HIP_CHECK(ctx, hipMemsetAsync(Bc + (size_t)nq * FN_K, 0, sizeof(_Float16) * (size_t)(nqp - nq) * FN_K, ctx->stream));
            PCR_HIP_CHECK(ctx, hipMemsetAsync(r2c, 0, sizeof(float) * 2 * (size_t)nqp, ctx->stream));
            if (nqp > nq) PCR_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)(Ug_seeded + nq), 0x7f800000, (size_t)(nqp - nq), ctx->stream));
            FnCompactArgs ca;
            ca.seedU = seedU; ca.live = live; ca.cpos = cpos; ca.n = n[0]; ca.B = B[0]; ca.nrm = rows[0].nrm; ca.cq = rows[0].cq; ca.perm = perm[0];
            ca.Bc = Bc; ca.nrm_c = r2c; ca.cq_c = r2c + nqp; ca.perm_c = perm_c; ca.Ug_c = Ug_seeded; ca.out = out;
            PCR_LAUNCH(ctx, k_fn_compact, dim3((n[0] + 31) / 32), dim3(256), 0, ctx->stream, ca);
            qB = Bc; q_nrm = r2c; q_cq = r2c + nqp; q_perm = perm_c;
            if (prune) {
                const int ntc = nqp / 64;
                float *lo = arena<float>(ctx, (size_t)FN_D * ntc), *hi = arena<float>(ctx, (size_t)FN_D * ntc);
                if (!lo || !hi) return PCR_ENOMEM;
                PCR_LAUNCH(ctx, k_fn_boxes, dim3((ntc + 3) / 4), dim3(256), 0, ctx->stream, (const float *)Pc[0], (const uint32_t *)perm_c, nq, ntc, ntc, lo, hi);
                q_blo = lo; q_bhi = hi; q_tile_stride = ntc;
            }
        }
        const int groups = nqp / FN_QPG;
        // a pre-pass seeds the bound of every query (first 4096 rows; with tile pruning the 48 tiles nearest to the workgroup's queries);
        // the main pass splits the steps so that the grid has ~8 workgroups per CU (one workgroup = 512 queries x one split)
        const int pre = prune ? 0 : (steps <= 128 ? steps : 64);
        const int rest = steps - pre;
        int splits = rest > 0 ? (2048 + groups - 1) / groups : 1;
        if (prune && rest > 0) splits = (FN_WGS_PRUNED + groups - 1) / groups;                // lists hold ~1/5 of their range and differ in length: more, shorter
                                                                                     //   ranges balance better (200k x 200k: 13.4 / 12.6 / 12.1 / 12.3 ms for 4 / 8 / 12 / 16)
        if (splits > 256) splits = 256;
        if (splits < (rest + FN_LIST_CAP - 1) / FN_LIST_CAP) splits = (rest + FN_LIST_CAP - 1) / FN_LIST_CAP;
        if (splits > rest) splits = rest > 0 ? rest : 1;
        const int sps = rest > 0 ? (rest + splits - 1) / splits : 0;
        if (sps > FN_LIST_CAP || pre > FN_LIST_CAP) return PCR_ECAPACITY;
        int splits_all = (2048 + groups - 1) / groups;                                // the pass with records over ALL steps (unpruned form)
        if (splits_all > 256) splits_all = 256;
        // a workgroup first loads the B operands of its 512 queries (128 KB): under FN_MIN_SPS steps of 16 KB each that load is most of its
        // traffic (NCLT-size clouds: 40 groups x 52 splits of 6 steps read 8x what 9 splits of 35 read; the matches are exact either way)
        if (splits_all > (steps + FN_MIN_SPS - 1) / FN_MIN_SPS) splits_all = (steps + FN_MIN_SPS - 1) / FN_MIN_SPS;
        if (splits_all < (steps + FN_LIST_CAP - 1) / FN_LIST_CAP) splits_all = (steps + FN_LIST_CAP - 1) / FN_LIST_CAP;
        if (splits_all > steps) splits_all = steps;
        const int sps_all = (steps + splits_all - 1) / splits_all;
        if (sps_all > FN_LIST_CAP) return PCR_ECAPACITY;
        // capacity: records per query plus the chunk every wavefront of either pass may leave partly filled
        const size_t waves = (size_t)groups * (FN_WG / 64) * (size_t)(FN_PRE_SPLIT + (splits > splits_all ? splits : splits_all));
        const int pool_cap = (int)(((size_t)nqp * FN_POOL_PER_QUERY + waves * FN_CHUNK + FN_CHUNK - 1) / FN_CHUNK * FN_CHUNK);
        int *pool_used = arena<int>(ctx, 1), *chunk_fill = arena<int>(ctx, pool_cap / FN_CHUNK);
        int *rec_q = arena<int>(ctx, pool_cap), *rec_row = arena<int>(ctx, pool_cap); float *rec_w = arena<float>(ctx, pool_cap);
        int *Ug = seeded ? Ug_seeded : arena<int>(ctx, nqp);
        unsigned long long *best_d = (mutual_only && dir == 0) ? best_first : arena<unsigned long long>(ctx, nq);
        double *rec_d = arena<double>(ctx, pool_cap);
        if (!pool_used || !chunk_fill || !rec_q || !rec_row || !rec_w || !Ug || !best_d || !rec_d) return PCR_ENOMEM;
        {
            const int n_chunks = pool_cap / FN_CHUNK, n_init = std::max(std::max(n_chunks, nqp), nq);
            // (seeded: the bounds and the live rows' INT_MAX in `out` come from k_fn_compact; rows beyond the live ones are dead by their index)
            PCR_LAUNCH(ctx, k_fn_init, dim3((n_init + 255) / 256), dim3(256), 0, ctx->stream, pool_used, chunk_fill, n_chunks, Ug, seeded ? 0 : nqp, best_d, seeded ? (int32_t *)nullptr : out, nq,
                       (int *)nullptr, (int *)nullptr);
        }
        if (dbg) PCR_HIP_CHECK(ctx, hipMemsetAsync(dbg, 0, 8, ctx->stream));
        FnnArgs a;
        a.dbA = A[dc]; a.db_nlo = rows[dc].nlo; a.db_nup = rows[dc].nup; a.n_db_pad = np[dc]; a.qB = qB; a.q_nrm = q_nrm; a.q_cq = q_cq; a.n_q = nq; a.n_q_pad = nqp;
        a.db_first_zero = first_zero + dc; a.Ug = Ug;
        a.pool_used = pool_used; a.pool_cap = pool_cap; a.chunk_fill = chunk_fill; a.rec_q = rec_q; a.rec_row = rec_row; a.rec_w = rec_w; a.flags = flags;
        a.L = nullptr; a.L_stride = 0; a.n_qt = 0; a.prelist = nullptr; a.pre_mode = 0; a.n_bound = 0; a.xcd_chunk = 0; a.stats = nullptr;
        a.seeded = seeded ? 1 : 0;
        unsigned long long *stats = nullptr;
        if (check || ctx->profiling) { stats = arena<unsigned long long>(ctx, 16); if (!stats) return PCR_ENOMEM; PCR_HIP_CHECK(ctx, hipMemsetAsync(stats, 0, 128, ctx->stream)); a.stats = stats; }
        if (prune && seeded) {      // the bounds are final already: one pass over the tiles within them
            const int nqt = (nq + 63) / 64, nbt = steps;
            float *L = arena<float>(ctx, (size_t)nqt * nbt);
            if (!L) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fn_boxlb, dim3((nbt + 255) / 256, nqt), dim3(256), 0, ctx->stream, q_blo, q_bhi, q_tile_stride,
                       (const float *)blo[dc], (const float *)bhi[dc], tile_stride[dc], nbt, eps2, L, nbt);
            a.L = L; a.L_stride = nbt; a.n_qt = nqt; a.prelist = nullptr;
            a.pre_mode = 0; a.step0 = 0; a.steps_per_split = sps; a.step_end = steps;
            a.xcd_chunk = FN_XCD_ORDER ? (groups + 7) / 8 : 0;
            PCR_LAUNCH(ctx, k_feature_nn_screen<false>, dim3(a.xcd_chunk > 0 ? 8 * a.xcd_chunk : groups, splits), dim3(FN_WG), 0, ctx->stream, a);
            a.xcd_chunk = 0;
        } else if (prune) {
            const int nqt = (nq + 63) / 64, nbt = steps;
            float *L = arena<float>(ctx, (size_t)nqt * nbt);
            int *prelist = arena<int>(ctx, (size_t)groups * FN_NPRE);
            if (!L || !prelist) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fn_boxlb, dim3((nbt + 255) / 256, nqt), dim3(256), 0, ctx->stream, (const float *)blo[qc], (const float *)bhi[qc], tile_stride[qc],
                       (const float *)blo[dc], (const float *)bhi[dc], tile_stride[dc], nbt, eps2, L, nbt);
            PCR_LAUNCH(ctx, k_fn_prelist, dim3(groups), dim3(256), 0, ctx->stream, (const float *)L, nbt, nqt, nbt, prelist);
            a.L = L; a.L_stride = nbt; a.n_qt = nqt; a.prelist = prelist;
            // (i) bounds only over the 16 nearest tiles (in a sorted database the running minimum improves row after row: with records
            // that is a record per improvement), (ii) the 48 nearest tiles with records under that bound, (iii) everything else that survives
            const int n_bound = FN_NBOUND;
            a.step0 = 0; a.steps_per_split = 0; a.step_end = steps; a.pre_mode = 1; a.n_bound = n_bound < FN_NPRE ? n_bound : FN_NPRE;
            if (a.n_bound > 0) PCR_LAUNCH(ctx, k_feature_nn_screen<true>, dim3(groups, FN_PRE_SPLIT_B), dim3(FN_WG), 0, ctx->stream, a);
            PCR_LAUNCH(ctx, k_feature_nn_screen<false>, dim3(groups, FN_PRE_SPLIT), dim3(FN_WG), 0, ctx->stream, a);
            if (check) {    // the pre-pass by itself (diagnostics: a wait and a read-back in the middle of the direction)
                unsigned long long hs[16] = {0}; int used = 0;
                PCR_HIP_CHECK(ctx, hipMemcpyAsync(hs, stats, 128, hipMemcpyDeviceToHost, ctx->stream));
                PCR_HIP_CHECK(ctx, hipMemcpyAsync(&used, pool_used, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                fprintf(stderr, "featnn dir %d pre-pass (%d nearest tiles per workgroup): %d records allocated (%.1f per query); stamps (M ticks): fetch %.1f, barrier %.1f, fast path %.1f, candidate path %.1f (%llu hit blocks), stash %.1f\n",
                        dir, FN_NPRE, used, (double)used / nq, hs[4] * 1e-6, hs[5] * 1e-6, hs[6] * 1e-6, hs[7] * 1e-6, hs[9], hs[8] * 1e-6);
            }
            a.pre_mode = 0; a.step0 = 0; a.steps_per_split = sps; a.step_end = steps;
            a.xcd_chunk = FN_XCD_ORDER ? (groups + 7) / 8 : 0;
            PCR_LAUNCH(ctx, k_feature_nn_screen<false>, dim3(a.xcd_chunk > 0 ? 8 * a.xcd_chunk : groups, splits), dim3(FN_WG), 0, ctx->stream, a);
            a.xcd_chunk = 0;
        } else {
            // Seeding the bounds: a sweep over the first 4096 rows that only lowers the queries' upper bounds (no candidates, no records)
            // can be split over many workgroups -- the bounds meet in Ug by atomicMin -- where a pre-pass WITH records had to be one
            // workgroup per 512 queries walking its 64 steps one after the other (200 us per direction when the grid is small: 20k rows are
            // 40 workgroups).  The pass with records then covers every step, under those bounds.
            {
                const int pre_b = steps < 64 ? steps : 64, pb_splits = pre_b < 8 ? pre_b : 8, pb_sps = (pre_b + pb_splits - 1) / pb_splits;
                a.step0 = 0; a.steps_per_split = pb_sps; a.step_end = pre_b;
                if (!seeded) PCR_LAUNCH(ctx, k_feature_nn_screen<true>, dim3(groups, pb_splits), dim3(FN_WG), 0, ctx->stream, a);
                a.step0 = 0; a.steps_per_split = sps_all; a.step_end = steps;
                PCR_LAUNCH(ctx, k_feature_nn_screen<false>, dim3(groups, splits_all), dim3(FN_WG), 0, ctx->stream, a);
            }
        }
        FnxArgs x;
        x.q = f[qc]; x.q_nrm = q_nrm; x.q_cq = q_cq; x.n_q = nq; x.db = f[dc]; x.n_db = n[dc]; x.db_nlo = rows[dc].nlo; x.db_nup = rows[dc].nup; x.db_nrm = rows[dc].nrm;
        x.db_first_zero = first_zero + dc; x.Ug = Ug;
        x.pool_used = pool_used; x.chunk_fill = chunk_fill; x.rec_q = rec_q; x.rec_row = rec_row; x.rec_w = rec_w; x.best_d = best_d; x.out = out; x.dbg = dbg;
        x.perm_q = q_perm; x.perm_db = perm[dc]; x.rec_d = rec_d;
        const int xgrid = std::min((pool_cap + 255) / 256, 4096);
        PCR_LAUNCH(ctx, k_fn_exact_min, dim3(xgrid), dim3(256), 0, ctx->stream, x, pool_cap);
        PCR_LAUNCH(ctx, k_fn_exact_arg, dim3(xgrid), dim3(256), 0, ctx->stream, x, pool_cap);
        PCR_LAUNCH(ctx, k_fn_finish, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, x);
        if (ctx->profiling && !check) {     // bench.py: (wavefront, 64-row step) pairs the screen computed, of all -- what it executes of the all-pairs flops
            unsigned long long hs[4] = {0, 0, 0, 0};
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(hs, stats, 32, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->prof[12] += (double)hs[1]; ctx->prof[15] += (double)groups * (FN_WG / 64) * (double)steps;
        }
        if (check) {
            int h[2] = {0, 0}; float hd[2] = {0, 0};
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(&h[0], pool_used, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(&h[1], flags, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(hd, dbg, 8, hipMemcpyDeviceToHost, ctx->stream));
            unsigned long long hs[16] = {0};
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(hs, stats, 128, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (hs[4] + hs[5] + hs[6]) fprintf(stderr, "featnn dir %d: wavefront cycles per phase of the step loop (FN_STAMPS, summed over wavefronts, M cycles): fetch issue %.1f, barrier wait %.1f, fast path %.1f, "
                                                       "candidate path %.1f (%llu hit blocks), stash %.1f\n", dir, hs[4] * 1e-6, hs[5] * 1e-6, hs[6] * 1e-6, hs[7] * 1e-6, hs[9], hs[8] * 1e-6);
            fprintf(stderr, "featnn dir %d: tile pruning %s: %llu steps staged of %lld (workgroup x step), %llu (wavefront, step) pairs computed of %lld\n", dir, prune ? "on" : "off",
                    hs[0], (long long)groups * steps, hs[1], (long long)groups * (FN_WG / 64) * steps);
            fprintf(stderr, "featnn dir %d: %d queries x %d rows, pre %d + %d splits x %d steps of %d rows; pool %d of %d records allocated (%.1f per query)%s, %.0f survived the final bound (%.2f per query); "
                            "max |d~-d| / (E_q + E_b) = %.3f (must stay under 1; K = %d, %d covered dimensions)\n", dir, nq, n[dc], pre, rest > 0 ? splits : 0, sps, FN_STEP, h[0], pool_cap, (double)h[0] / nq,
                    h[1] ? " OVERFLOW" : "", hd[1], hd[1] / nq, hd[0], FN_K, FN_NCOV);
        }
    }
    {   // pool overflow (either direction): the results are incomplete, the caller recomputes on the float64 path
        int h = 0;
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(&h, flags, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (h) return PCR_ECAPACITY;
    }
    return PCR_OK;
}

// ---- the mutual search of G pairs through the same launches (lockstep FGR groups).  Only the form WITHOUT tile pruning (every pair under
// 5e9 row pairs: NCLT-size clouds, where the chain of small launches per pair is the whole cost); per pair the arithmetic is that of
// pcr_feature_nn_mutual (the steps are cut differently -- the grid is shared by 2 G problems -- but the matches are the exact float64 nearest rows
// whatever the cut: the same bits).  Returns PCR_ECAPACITY when a
// pair wants the pruned form or its features leave the f16 range (the caller then runs the pairs one by one); `overflow_dev[g]` = a device
// int (the words of all pairs lie in ONE array, two per pair, so overflow_dev[0] + 2 g is pair g's) the caller reads with its next read-back: != 0 means the record pool of pair g overflowed (that pair alone is redone on the float64 path).
int pcr_feature_nn_mutual_batch(pcr_context *ctx, int G, const float *const *f0, const int *n0, const float *const *f1, const int *n1, int32_t *const *out_1to0,
                                int32_t *const *out_0to1, const int **overflow_dev, int mutual_only) {
    if (G < 1) return PCR_OK;
    for (int g = 0; g < G; g++) {
        if (n0[g] < 64 || n1[g] < 64) return PCR_ECAPACITY;
        if ((double)n0[g] * (double)n1[g] >= 5.0e9) return PCR_ECAPACITY;
    }
    const int nbm = 64;
    struct Pair { const float *f[2]; int n[2], np[2]; double *part, *mu; int *first_zero, *flags; _Float16 *A[2], *B[2]; FnRows rows[2]; };
    std::vector<Pair> P((size_t)G);
    std::vector<FnColsumDesc> cs((size_t)2 * G); std::vector<FnMeanDesc> ms((size_t)G);
    double *mu_all = arena<double>(ctx, (size_t)G * FN_PC);              // per-group arrays: one read-back each
    int *flags_all = arena<int>(ctx, (size_t)2 * G), *zero_all = arena<int>(ctx, (size_t)2 * G);
    if (!mu_all || !flags_all || !zero_all) return PCR_ENOMEM;
    for (int g = 0; g < G; g++) {
        Pair &p = P[g];
        p.f[0] = f0[g]; p.f[1] = f1[g]; p.n[0] = n0[g]; p.n[1] = n1[g];
        p.part = arena<double>(ctx, (size_t)2 * nbm * FN_PC);
        p.first_zero = zero_all + 2 * g; p.flags = flags_all + 2 * g;
        if (!p.part) return PCR_ENOMEM;
        p.mu = mu_all + (size_t)g * FN_PC;
        cs[2 * g] = FnColsumDesc{p.f[0], p.n[0], p.part}; cs[2 * g + 1] = FnColsumDesc{p.f[1], p.n[1], p.part + (size_t)nbm * FN_PC};
        ms[g] = FnMeanDesc{p.part, nbm, p.n[0], p.mu, p.first_zero, p.flags};
        for (int c = 0; c < 2; c++) p.np[c] = (p.n[c] + FN_QPG - 1) / FN_QPG * FN_QPG;
    }
    {
        const FnColsumDesc *dc = pcr_desc_upload(ctx, cs.data(), 2 * G);
        const FnMeanDesc *dm = pcr_desc_upload(ctx, ms.data(), G);
        if (!dc || !dm) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_fn_colsum_g, dim3(nbm, 2 * G), dim3(256), 0, ctx->stream, dc);
        PCR_LAUNCH(ctx, k_fn_mean_g, dim3(G), dim3(128), 0, ctx->stream, dm);
    }
    {   // the f16 split holds |f - mu| * 128 < 65504 (pcr_feature_nn_mutual): one read-back for the group
        std::vector<double> hmu((size_t)G * FN_PC);
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(hmu.data(), mu_all, sizeof(double) * hmu.size(), hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (int g = 0; g < G; g++) {
            double mm = 0; for (int k = 0; k < FN_D; k++) mm = fmax(mm, fabs(hmu[(size_t)g * FN_PC + k]));
            if (!((hmu[(size_t)g * FN_PC + FN_D] + mm) * FN_SCALE < 60000.0)) return PCR_ECAPACITY;
        }
    }
    std::vector<FnSplitDesc> ss((size_t)2 * G);
    int max_np = 0;
    for (int g = 0; g < G; g++) {
        Pair &p = P[g];
        for (int c = 0; c < 2; c++) {
            p.A[c] = arena<_Float16>(ctx, (size_t)p.np[c] * FN_K); p.B[c] = arena<_Float16>(ctx, (size_t)p.np[c] * FN_K);
            float *r4 = arena<float>(ctx, (size_t)4 * p.np[c]);
            if (!p.A[c] || !p.B[c] || !r4) return PCR_ENOMEM;
            p.rows[c] = FnRows{r4, r4 + p.np[c], r4 + (size_t)2 * p.np[c], r4 + (size_t)3 * p.np[c]};
            ss[2 * g + c] = FnSplitDesc{p.f[c], p.n[c], p.np[c], p.mu, p.A[c], p.B[c], p.rows[c], p.first_zero + c};
            max_np = p.np[c] > max_np ? p.np[c] : max_np;
        }
    }
    {
        const FnSplitDesc *ds = pcr_desc_upload(ctx, ss.data(), 2 * G);
        if (!ds) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_fn_split_g, dim3((unsigned)(((size_t)max_np * FN_SPLIT_LANES + 255) / 256), 2 * G), dim3(256), 0, ctx->stream, ds);
    }
    // problems = (pair, direction): direction 0 = queries of cloud 1 against the rows of cloud 0.  A problem's queries may be a subset of its cloud
    // (the live rows of a seeded second direction): qB / q_nrm / q_cq / perm_q describe them, Ug_seeded holds their bounds.
    struct Prob { int g, dir; const _Float16 *qB; const float *q_nrm, *q_cq; const uint32_t *perm_q; int nq, nqp; int *Ug_seeded; unsigned long long *best_d; };
    auto run = [&](const std::vector<Prob> &probs, const bool seeded) -> int {
        const int NP = (int)probs.size();
        if (NP == 0) return PCR_OK;
        std::vector<FnInitDesc> is((size_t)NP); std::vector<FnnArgs> bound((size_t)NP), rec((size_t)NP); std::vector<FnxDesc> xs((size_t)NP);
        int max_init = 0, max_groups = 0, max_pb = 0, max_splits = 0, max_pool = 0, max_nq = 0;
        for (int k = 0; k < NP; k++) {
            const Prob &pb = probs[k];
            Pair &p = P[pb.g];
            const int dir = pb.dir, qc = dir == 0 ? 1 : 0, dc = 1 - qc;
            int32_t *out = dir == 0 ? out_1to0[pb.g] : out_0to1[pb.g];
            const int nq = pb.nq, nqp = pb.nqp, steps = p.np[dc] / FN_STEP, groups = nqp / FN_QPG;
            // the grid is shared by the problems of the launch: ~3072 workgroups in all, and no workgroup under FN_MIN_SPS steps (it loads 128 KB of query
            // operands first; a group of 16 NCLT-size pairs cut as the one-pair call cuts them spent 26 ms per launch on 66 000 workgroups of 6 steps)
            int splits_all = (3072 + groups * NP - 1) / (groups * NP);
            if (splits_all > 256) splits_all = 256;
            if (splits_all > (steps + FN_MIN_SPS - 1) / FN_MIN_SPS) splits_all = (steps + FN_MIN_SPS - 1) / FN_MIN_SPS;
            if (splits_all < 1) splits_all = 1;
            if (splits_all < (steps + FN_LIST_CAP - 1) / FN_LIST_CAP) splits_all = (steps + FN_LIST_CAP - 1) / FN_LIST_CAP;
            if (splits_all > steps) splits_all = steps;
            const int sps_all = (steps + splits_all - 1) / splits_all;
            if (sps_all > FN_LIST_CAP) return PCR_ECAPACITY;
            // (the pool is sized as the one-pair call sizes it: its `splits` of the pruned form never exceeds splits_all when pruning is off)
            const int pre = steps <= 128 ? steps : 64, rest = steps - pre;
            int splits = rest > 0 ? (2048 + groups - 1) / groups : 1;
            if (splits > 256) splits = 256;
            if (splits < (rest + FN_LIST_CAP - 1) / FN_LIST_CAP) splits = (rest + FN_LIST_CAP - 1) / FN_LIST_CAP;
            if (splits > rest) splits = rest > 0 ? rest : 1;
            const size_t waves = (size_t)groups * (FN_WG / 64) * (size_t)(1 + (splits > splits_all ? splits : splits_all));
            const int pool_cap = (int)(((size_t)nqp * FN_POOL_PER_QUERY + waves * FN_CHUNK + FN_CHUNK - 1) / FN_CHUNK * FN_CHUNK);
            int *pool_used = arena<int>(ctx, 1), *chunk_fill = arena<int>(ctx, pool_cap / FN_CHUNK);
            int *rec_q = arena<int>(ctx, pool_cap), *rec_row = arena<int>(ctx, pool_cap); float *rec_w = arena<float>(ctx, pool_cap);
            int *Ug = pb.Ug_seeded ? pb.Ug_seeded : arena<int>(ctx, nqp);
            unsigned long long *best_d = pb.best_d ? pb.best_d : arena<unsigned long long>(ctx, nq);
            double *rec_d = arena<double>(ctx, pool_cap);
            if (!pool_used || !chunk_fill || !rec_q || !rec_row || !rec_w || !Ug || !best_d || !rec_d) return PCR_ENOMEM;
            const int n_chunks = pool_cap / FN_CHUNK, n_init = std::max(std::max(n_chunks, nqp), nq);
            // (seeded: the bounds and the live rows' INT_MAX in `out` come from k_fn_compact_g)
            is[k] = FnInitDesc{pool_used, chunk_fill, n_chunks, Ug, pb.Ug_seeded ? 0 : nqp, best_d, pb.Ug_seeded ? (int32_t *)nullptr : out, nq};
            FnnArgs a;
            a.dbA = p.A[dc]; a.db_nlo = p.rows[dc].nlo; a.db_nup = p.rows[dc].nup; a.n_db_pad = p.np[dc]; a.qB = pb.qB; a.q_nrm = pb.q_nrm; a.q_cq = pb.q_cq; a.n_q = nq; a.n_q_pad = nqp;
            a.db_first_zero = p.first_zero + dc; a.Ug = Ug;
            a.pool_used = pool_used; a.pool_cap = pool_cap; a.chunk_fill = chunk_fill; a.rec_q = rec_q; a.rec_row = rec_row; a.rec_w = rec_w; a.flags = p.flags;
            a.L = nullptr; a.L_stride = 0; a.n_qt = 0; a.prelist = nullptr; a.pre_mode = 0; a.n_bound = 0; a.xcd_chunk = 0; a.stats = nullptr;
            a.seeded = seeded ? 1 : 0;
            int pb_want = (2048 + groups * NP - 1) / (groups * NP);                // the bound-only sweep over the first 64 steps: as many cuts as fill the chip once
            if (pb_want > 8) pb_want = 8;
            const int pre_b = steps < 64 ? steps : 64, pb_splits = pre_b < pb_want ? pre_b : pb_want, pb_sps = (pre_b + pb_splits - 1) / pb_splits;
            a.step0 = 0; a.steps_per_split = pb_sps; a.step_end = pre_b;
            bound[k] = a;
            a.step0 = 0; a.steps_per_split = sps_all; a.step_end = steps;
            rec[k] = a;
            FnxArgs x;
            x.q = p.f[qc]; x.q_nrm = pb.q_nrm; x.q_cq = pb.q_cq; x.n_q = nq; x.db = p.f[dc]; x.n_db = p.n[dc]; x.db_nlo = p.rows[dc].nlo; x.db_nup = p.rows[dc].nup; x.db_nrm = p.rows[dc].nrm;
            x.db_first_zero = p.first_zero + dc; x.Ug = Ug;
            x.pool_used = pool_used; x.chunk_fill = chunk_fill; x.rec_q = rec_q; x.rec_row = rec_row; x.rec_w = rec_w; x.best_d = best_d; x.out = out; x.dbg = nullptr;
            x.perm_q = pb.perm_q; x.perm_db = nullptr; x.rec_d = rec_d;
            xs[k] = FnxDesc{x, pool_cap};
            max_init = n_init > max_init ? n_init : max_init; max_groups = groups > max_groups ? groups : max_groups; max_pb = pb_splits > max_pb ? pb_splits : max_pb;
            max_splits = splits_all > max_splits ? splits_all : max_splits; max_pool = pool_cap > max_pool ? pool_cap : max_pool; max_nq = nq > max_nq ? nq : max_nq;
        }
        const FnInitDesc *di = pcr_desc_upload(ctx, is.data(), NP);
        const FnnArgs *db = pcr_desc_upload(ctx, bound.data(), NP), *dr = pcr_desc_upload(ctx, rec.data(), NP);
        const FnxDesc *dx = pcr_desc_upload(ctx, xs.data(), NP);
        if (!di || !db || !dr || !dx) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_fn_init_g, dim3((max_init + 255) / 256, NP), dim3(256), 0, ctx->stream, di);
        if (!seeded) PCR_LAUNCH(ctx, k_feature_nn_screen_g<true>, dim3(max_groups, max_pb, NP), dim3(FN_WG), 0, ctx->stream, db);
        PCR_LAUNCH(ctx, k_feature_nn_screen_g<false>, dim3(max_groups, max_splits, NP), dim3(FN_WG), 0, ctx->stream, dr);
        const int xgrid = std::min((max_pool + 255) / 256, std::max(64, 4096 / NP));
        PCR_LAUNCH(ctx, k_fn_exact_min_g, dim3(xgrid, NP), dim3(256), 0, ctx->stream, dx);
        PCR_LAUNCH(ctx, k_fn_exact_arg_g, dim3(xgrid, NP), dim3(256), 0, ctx->stream, dx);
        PCR_LAUNCH(ctx, k_fn_finish_g, dim3((max_nq + 255) / 256, NP), dim3(256), 0, ctx->stream, dx);
        return PCR_OK;
    };
    const bool mutual = mutual_only && pcr_options().featnn_mutual.load(std::memory_order_relaxed) != 0;
    std::vector<Prob> first, second;
    std::vector<unsigned long long *> best_first((size_t)G, nullptr);
    for (int g = 0; g < G; g++) {
        Pair &p = P[g];
        if (mutual) { best_first[g] = arena<unsigned long long>(ctx, p.n[1]); if (!best_first[g]) return PCR_ENOMEM; }
        first.push_back(Prob{g, 0, p.B[1], p.rows[1].nrm, p.rows[1].cq, nullptr, p.n[1], p.np[1], nullptr, best_first[g]});
        if (!mutual) first.push_back(Prob{g, 1, p.B[0], p.rows[0].nrm, p.rows[0].cq, nullptr, p.n[0], p.np[0], nullptr, nullptr});
    }
    PCR_TRY(run(first, false));
    if (mutual) {
        // ---- the second direction, seeded by the first (k_fn_seed): only the rows some row of the other cloud points at, under the bound that row gives
        std::vector<int *> seedU((size_t)G), cpos((size_t)G), n_live_ptr((size_t)G); std::vector<uint8_t *> live((size_t)G); std::vector<int> caps((size_t)G);
        int *n_live_all = arena<int>(ctx, G);
        if (!n_live_all) return PCR_ENOMEM;
        std::vector<FnInitDesc> si((size_t)G); std::vector<FnSeedArgs> sa((size_t)G); std::vector<FnLiveDesc> sl((size_t)G);
        int max_np0 = 0, max_n1 = 0, max_n0 = 0;
        for (int g = 0; g < G; g++) {
            Pair &p = P[g];
            seedU[g] = arena<int>(ctx, p.np[0]); live[g] = arena<uint8_t>(ctx, p.n[0]); cpos[g] = arena<int>(ctx, p.n[0]);
            if (!seedU[g] || !live[g] || !cpos[g]) return PCR_ENOMEM;
            n_live_ptr[g] = n_live_all + g; caps[g] = p.n[0];
            si[g] = FnInitDesc{n_live_all + g, nullptr, 0, seedU[g], p.np[0], nullptr, nullptr, 0};
            FnSeedArgs a;
            a.out_prev = out_1to0[g]; a.best_prev = best_first[g]; a.perm_prev = nullptr; a.n_prev = p.n[1]; a.inv_cur = nullptr; a.q_nrm_cur = p.rows[0].nrm; a.Ug = seedU[g];
            sa[g] = a;
            sl[g] = FnLiveDesc{seedU[g], p.n[0], live[g], out_0to1[g]};
            max_np0 = std::max(max_np0, p.np[0]); max_n1 = std::max(max_n1, p.n[1]); max_n0 = std::max(max_n0, p.n[0]);
        }
        const FnInitDesc *dsi = pcr_desc_upload(ctx, si.data(), G);
        const FnSeedArgs *dsa = pcr_desc_upload(ctx, sa.data(), G);
        const FnLiveDesc *dsl = pcr_desc_upload(ctx, sl.data(), G);
        if (!dsi || !dsa || !dsl) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_fn_init_g, dim3((max_np0 + 255) / 256, G), dim3(256), 0, ctx->stream, dsi);
        PCR_LAUNCH(ctx, k_fn_seed_g, dim3((max_n1 + 255) / 256, G), dim3(256), 0, ctx->stream, dsa);
        PCR_LAUNCH(ctx, k_fn_live_g, dim3((max_n0 + 255) / 256, G), dim3(256), 0, ctx->stream, dsl);
        PCR_TRY(pcr_dev_flag_scan_batch(ctx, G, live.data(), nullptr, caps.data(), cpos.data(), n_live_ptr.data()));
        std::vector<int> n_live((size_t)G);
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(n_live.data(), n_live_all, sizeof(int) * (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<FnCompactArgs> ca;
        for (int g = 0; g < G; g++) {
            Pair &p = P[g];
            if (n_live[g] <= 0) continue;
            const int nq = n_live[g], nqp = (nq + FN_QPG - 1) / FN_QPG * FN_QPG;
            _Float16 *Bc = arena<_Float16>(ctx, (size_t)nqp * FN_K); float *r2c = arena<float>(ctx, (size_t)2 * nqp); uint32_t *perm_c = arena<uint32_t>(ctx, nqp);
            int *Ug_c = arena<int>(ctx, nqp);
            if (!Bc || !r2c || !perm_c || !Ug_c) return PCR_ENOMEM;
            PCR_HIP_CHECK(ctx, hipMemsetAsync(Bc + (size_t)nq * FN_K, 0, sizeof(_Float16) * (size_t)(nqp - nq) * FN_K, ctx->stream));
            PCR_HIP_CHECK(ctx, hipMemsetAsync(r2c, 0, sizeof(float) * 2 * (size_t)nqp, ctx->stream));
            if (nqp > nq) PCR_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)(Ug_c + nq), 0x7f800000, (size_t)(nqp - nq), ctx->stream));
            FnCompactArgs c;
            c.seedU = seedU[g]; c.live = live[g]; c.cpos = cpos[g]; c.n = p.n[0]; c.B = p.B[0]; c.nrm = p.rows[0].nrm; c.cq = p.rows[0].cq; c.perm = nullptr;
            c.Bc = Bc; c.nrm_c = r2c; c.cq_c = r2c + nqp; c.perm_c = perm_c; c.Ug_c = Ug_c; c.out = out_0to1[g];
            ca.push_back(c);
            second.push_back(Prob{g, 1, Bc, r2c, r2c + nqp, perm_c, nq, nqp, Ug_c, nullptr});
        }
        if (!ca.empty()) {
            const FnCompactArgs *dca = pcr_desc_upload(ctx, ca.data(), (int)ca.size());
            if (!dca) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fn_compact_g, dim3((max_n0 + 31) / 32, (unsigned)ca.size()), dim3(256), 0, ctx->stream, dca);
            PCR_TRY(run(second, true));
        }
    }
    for (int g = 0; g < G; g++) overflow_dev[g] = P[g].flags;         // pool overflow of a pair (either direction): its results are incomplete
    return PCR_OK;
}
