// pcr_fgr.hip -- FPFH features (K6) and Fast Global Registration (K7 feature matching, K8 tuple test,
// K9 graduated-non-convexity Gauss-Newton) on gfx950.
// Reference behaviour: Open3D ComputeFPFHFeature and FastGlobalRegistrationBasedOnFeatureMatching as called at
// ALL_FUNCTIONS.py:186-187,198-202 / 1_FGR_pairwise_registration_in_NCLT_dataset.py:49-50,61-65.
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <array>
#include <vector>
#include "pcr_octree.h"

#define FB 256
#define PI_D 3.14159265358979323846

// =============================================================================================== FPFH (K6)
// Neighbour lists come from the octet k-NN (hybrid radius / max_nn, pcr_cloud.hip) in its distributed layout:
// list[q*K + slot], slot = lane + 8*j.  The same 8 lanes then build the histogram of the point.

__device__ static inline int bin11(double x) {
    int h = (int)floor(x);
    return h < 0 ? 0 : (h > 10 ? 10 : h);
}

// Open3D ComputePairFeatures(p1,n1,p2,n2) -> the three histogram bins (degenerate pairs vote in the bins of f = 0)
__device__ static inline void pair_bins(const double *p1, const double *n1, const double *p2, const double *n2, int *b0, int *b1, int *b2) {
    // Products and sums rounded one by one, as the x86 build of Open3D (and the oracle) rounds them: a direction exactly along the frame's
    // normal gives v = 0 -- the degenerate vote in the bins of f = 0 -- only if dy * az - dz * ay cancels as two rounded products do; a fused
    // multiply-add leaves a residue of 1e-17 there and the vote lands three bins away (seen at 200k points: 26 histograms of 400 000).
#pragma clang fp contract(off)
    double dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = p2[2] - p1[2];
    double f0 = 0, f1 = 0, f2 = 0;
    const double len = sqrt(dx * dx + dy * dy + dz * dz);
    if (len != 0.0) {
        double ax = n1[0], ay = n1[1], az = n1[2], bx = n2[0], by = n2[1], bz = n2[2];
        const double angle1 = (ax * dx + ay * dy + az * dz) / len, angle2 = (bx * dx + by * dy + bz * dz) / len;
        double g2;
        if (acos(fabs(angle1)) > acos(fabs(angle2))) {
            ax = n2[0]; ay = n2[1]; az = n2[2]; bx = n1[0]; by = n1[1]; bz = n1[2];
            dx = -dx; dy = -dy; dz = -dz;
            g2 = -angle2;
        } else g2 = angle1;
        double vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;      // dp x n1
        const double vn = sqrt(vx * vx + vy * vy + vz * vz);
        if (vn != 0.0) {
            vx /= vn; vy /= vn; vz /= vn;
            const double wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;   // n1 x v
            f2 = g2;
            f1 = vx * bx + vy * by + vz * bz;
            f0 = atan2(wx * bx + wy * by + wz * bz, ax * bx + ay * by + az * bz);
        }
    }
    *b0 = bin11(11.0 * (f0 + PI_D) / (2.0 * PI_D));
    *b1 = bin11(11.0 * (f1 + 1.0) * 0.5) + 11;
    *b2 = bin11(11.0 * (f2 + 1.0) * 0.5) + 22;
}

// The same three bins from float arithmetic, or `false` when float cannot decide them: a bin coordinate within SPFH_MARGIN of a bin
// edge (float error here: <= 1e-4 of a bin under the two conditioning guards), the two angles too close to order, a direction nearly
// parallel to the frame's normal, the second normal nearly parallel to the frame's v.  The caller then evaluates pair_bins (float64),
// so the histogram is the float64 one bin for bin ("spfh_float64" = 1 evaluates every pair that way: tests compare the two).
#define SPFH_MARGIN 1e-3f
__device__ static inline bool pair_bins_fast(const float4 p1, const float4 n1, const float4 p2, const float4 n2, int *b0, int *b1, int *b2) {
    float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
    const float l2 = dx * dx + dy * dy + dz * dz;
    if (!(l2 > 1e-20f)) return false;
    const float il = __builtin_amdgcn_rsqf(l2);
    const float a1 = (n1.x * dx + n1.y * dy + n1.z * dz) * il, a2 = (n2.x * dx + n2.y * dy + n2.z * dz) * il;
    const float fa1 = fabsf(a1), fa2 = fabsf(a2);
    // (equal normals -- the (0, 0, 1) of points without a neighbourhood, by the hundred on sparse scans: the float64 angles are the same
    // expression of the same operands, equal, and the frame stays with the first point)
    const bool same = n1.x == n2.x && n1.y == n2.y && n1.z == n2.z;
    if (!same && fabsf(fa1 - fa2) < 1e-5f) return false;
    const bool sw = !same && fa1 < fa2;  // acos|angle1| > acos|angle2|
    const float ax = sw ? n2.x : n1.x, ay = sw ? n2.y : n1.y, az = sw ? n2.z : n1.z;
    const float bx = sw ? n1.x : n2.x, by = sw ? n1.y : n2.y, bz = sw ? n1.z : n2.z;
    if (sw) { dx = -dx; dy = -dy; dz = -dz; }
    const float g2 = sw ? -a2 : a1;
    float vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;
    const float vn2 = vx * vx + vy * vy + vz * vz;
    if (!(vn2 * il * il > 0.01f)) return false;
    const float iv = __builtin_amdgcn_rsqf(vn2);
    vx *= iv; vy *= iv; vz *= iv;
    const float wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;
    const float f1 = vx * bx + vy * by + vz * bz;
    const float y = wx * bx + wy * by + wz * bz, x = ax * bx + ay * by + az * bz;
    if (!(x * x + y * y > 0.01f)) return false;
    const float c0 = atan2f(y, x) * (float)(5.5 / PI_D) + 5.5f, c1 = 5.5f * f1 + 5.5f, c2 = 5.5f * g2 + 5.5f;
    const float r0 = rintf(c0), r1 = rintf(c1), r2 = rintf(c2);
    if (fabsf(c0 - r0) < SPFH_MARGIN) return false;                              // 0 and 11 are edges too: the angle wraps there
    if (fabsf(c1 - r1) < SPFH_MARGIN && r1 >= 1.0f && r1 <= 10.0f) return false;
    if (fabsf(c2 - r2) < SPFH_MARGIN && r2 >= 1.0f && r2 <= 10.0f) return false;
    const int h0 = (int)floorf(c0), h1 = (int)floorf(c1), h2 = (int)floorf(c2);
    *b0 = h0 < 0 ? 0 : (h0 > 10 ? 10 : h0);
    *b1 = (h1 < 0 ? 0 : (h1 > 10 ? 10 : h1)) + 11;
    *b2 = (h2 < 0 ? 0 : (h2 > 10 ? 10 : h2)) + 22;
    return true;
}

// SPFH of a point as it is kept between the two kernels: the 33 bin COUNTS (<= max_nn <= 200: a byte each) and 100 / (m - 1); the value
// Open3D holds, (double)count * inc, is formed where it is used (the same product, the same bits) -- 48 B per neighbour gathered by the
// FPFH pass instead of 264.
struct __attribute__((aligned(16))) SpfhRow { uint8_t bins[40]; double inc; };
static_assert(sizeof(SpfhRow) == 48, "SpfhRow");

struct FpfhArgs {
    const float4 *pts, *nrm; const int *n_ptr;
    const int32_t *nbr; int k; double r2;       // neighbour lists (sorted-cloud indices, -1 padded)
    const int32_t *cnt;                         // optional: entries of each row (pcr_dev_radius_lists_batch; -1 or null: all k slots)
    SpfhRow *spfh;                              // n rows
    const uint32_t *perm; float *feat;          // output rows in caller order
    int float64_only;                           // option "spfh_float64"
    int *verify;
    // float pass (k_spfh_fast): the pairs float cannot decide go to a queue of (point, slot), which k_spfh_slow evaluates in float64 and adds to
    // the byte counters; a queue that overflowed (slow_count > slow_cap) makes k_spfh redo every row in float64 (only_if_over points at the count)
    uint2 *slowq; int *slow_count; int slow_cap;
    const int *only_if_over; int over_cap;
};

// The float pass: light enough in registers for eight wavefronts per SIMD (the gathers of a lane's neighbours are a chain of dependent
// loads: the pass is bound by how many of them are in flight; with the float64 evaluation inline the kernel held 178 VGPRs, two wavefronts).
// Below this many points the float pass does not pay: NCLT-size clouds (20-30k points, 46 neighbours on average) spend their SPFH time on launches and
// gather latency, not on float64 arithmetic, and three launches (float pass, queue, gate) are slower than one (script-1 stage in lockstep groups: 2290
// pairs/s all float64 against 2120-2200 with the float pass); at 175k points of 100 neighbours it is 0.7 -> 0.3 ms per cloud.
#define SPFH_SPLIT_MIN_POINTS 60000
#define SPFH_WGQ 192
__device__ static inline void d_spfh_fast(const FpfhArgs &a) {
    __shared__ int hist[FB / OCT][33];
    __shared__ unsigned wgq[SPFH_WGQ];          // the workgroup's undecided pairs (octet << 16 | slot): ONE reservation in the global queue per workgroup
    __shared__ int nwgq, wgq_base;
    const int n = *a.n_ptr;
    const int ol = threadIdx.x & 7, ob = threadIdx.x >> 3;
    const int qi = blockIdx.x * (FB / OCT) + ob;
    for (int b = ol; b < 33; b += OCT) hist[ob][b] = 0;
    if (threadIdx.x == 0) nwgq = 0;
    __syncthreads();
    int cnt = 0;
    if (qi < n) {
        const float4 pf = a.pts[qi], nf = a.nrm[qi];
        const int kk = a.cnt ? (a.cnt[qi] < 0 ? a.k : a.cnt[qi]) : a.k;
        for (int slot = ol; slot < kk; slot += OCT) {
            const int id = a.nbr[(size_t)qi * a.k + slot];
            if (id < 0 || id == qi) continue;
            const float4 qf = a.pts[id], mf = a.nrm[id];
            const double dx = (double)qf.x - (double)pf.x, dy = (double)qf.y - (double)pf.y, dz = (double)qf.z - (double)pf.z;
            if (!(dx * dx + dy * dy + dz * dz < a.r2)) continue;
            cnt++;
            int b0, b1, b2;
            if (!pair_bins_fast(pf, nf, qf, mf, &b0, &b1, &b2)) {
                const int lq = atomicAdd(&nwgq, 1);
                if (lq < SPFH_WGQ) wgq[lq] = (unsigned)ob << 16 | (unsigned)slot;
                else { const int at = atomicAdd(a.slow_count, 1); if (at < a.slow_cap) a.slowq[at] = make_uint2((unsigned)qi, (unsigned)slot); }
                continue;
            }
            atomicAdd(&hist[ob][b0], 1); atomicAdd(&hist[ob][b1], 1); atomicAdd(&hist[ob][b2], 1);
        }
    }
    cnt = pcr_octet_sum_i(cnt);
    __syncthreads();
    const int nl = nwgq < SPFH_WGQ ? nwgq : SPFH_WGQ;
    if (threadIdx.x == 0 && nl > 0) wgq_base = atomicAdd(a.slow_count, nl);
    if (qi < n) {
        SpfhRow *row = a.spfh + qi;
        for (int b = ol; b < 40; b += OCT) row->bins[b] = b < 33 ? (uint8_t)hist[ob][b] : (uint8_t)0;
        if (ol == 0) row->inc = cnt > 0 ? 100.0 / (double)cnt : 0.0;       // 100 / (m - 1), m counts the point itself
    }
    if (nl == 0) return;
    __syncthreads();
    for (int e = threadIdx.x; e < nl; e += FB) {
        const int at = wgq_base + e;
        if (at < a.slow_cap) a.slowq[at] = make_uint2((unsigned)(blockIdx.x * (FB / OCT)) + (wgq[e] >> 16), wgq[e] & 0xffffu);
    }
}
__device__ static inline void d_spfh_slow(const FpfhArgs &a) {
    const int total = *a.slow_count;
    if (total > a.slow_cap) return;                    // k_spfh redoes everything
    for (int e = blockIdx.x * FB + threadIdx.x; e < total; e += gridDim.x * FB) {
        const uint2 w = a.slowq[e];
        const int q2 = (int)w.x, id = a.nbr[(size_t)q2 * a.k + w.y];
        const float4 pf = a.pts[q2], nf = a.nrm[q2], qf = a.pts[id], mf = a.nrm[id];
        const double p1[3] = {pf.x, pf.y, pf.z}, n1[3] = {nf.x, nf.y, nf.z}, p2[3] = {qf.x, qf.y, qf.z}, n2[3] = {mf.x, mf.y, mf.z};
        int b0, b1, b2;
        pair_bins(p1, n1, p2, n2, &b0, &b1, &b2);
        unsigned *words = reinterpret_cast<unsigned *>(a.spfh[q2].bins);       // byte counters (<= 200 each): +1 in byte b of its 32-bit word
        atomicAdd(words + (b0 >> 2), 1u << (8 * (b0 & 3))); atomicAdd(words + (b1 >> 2), 1u << (8 * (b1 & 3))); atomicAdd(words + (b2 >> 2), 1u << (8 * (b2 & 3)));
    }
}
__global__ void __launch_bounds__(FB) k_spfh_fast(FpfhArgs a) { d_spfh_fast(a); }
__global__ void __launch_bounds__(FB) k_spfh_slow(FpfhArgs a) { d_spfh_slow(a); }
__global__ void __launch_bounds__(FB) k_spfh_fast_g(const FpfhArgs *a) { d_spfh_fast(a[blockIdx.y]); }
__global__ void __launch_bounds__(FB) k_spfh_slow_g(const FpfhArgs *a) { d_spfh_slow(a[blockIdx.y]); }

// Every pair in float64 (option "spfh_float64" = 1, the overflow path of the float pass, and -- with a.verify -- the comparison of the two).
__device__ static inline void d_spfh(const FpfhArgs &a) {
    if (a.only_if_over && *a.only_if_over <= a.over_cap) return;         // (the float pass and its queue did the work)
    __shared__ int hist[FB / OCT][33];
    const int n = *a.n_ptr;
    const int ol = threadIdx.x & 7, ob = threadIdx.x >> 3;
    // (a fixed grid strides over the blocks of 32 points: the gated launch behind the float pass is a few hundred workgroups that read one word)
    for (int blk = blockIdx.x; blk * (FB / OCT) < n; blk += gridDim.x) {
    const int qi = blk * (FB / OCT) + ob;
    __syncthreads();
    for (int b = ol; b < 33; b += OCT) hist[ob][b] = 0;
    __syncthreads();
    int cnt = 0;
    if (qi < n) {
        const float4 pf = a.pts[qi], nf = a.nrm[qi];
        const int kk = a.cnt ? (a.cnt[qi] < 0 ? a.k : a.cnt[qi]) : a.k;
        for (int slot = ol; slot < kk; slot += OCT) {
            const int id = a.nbr[(size_t)qi * a.k + slot];
            if (id < 0 || id == qi) continue;
            const float4 qf = a.pts[id], mf = a.nrm[id];
            const double p1[3] = {pf.x, pf.y, pf.z}, n1[3] = {nf.x, nf.y, nf.z}, p2[3] = {qf.x, qf.y, qf.z}, n2[3] = {mf.x, mf.y, mf.z};
            const double dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = p2[2] - p1[2];
            if (!(dx * dx + dy * dy + dz * dz < a.r2)) continue;
            cnt++;
            int b0, b1, b2;
            pair_bins(p1, n1, p2, n2, &b0, &b1, &b2);
            if (a.verify) {                       // option "spfh_float64" = 2: where the float form answers, it must give the same bins; disagreements are counted
                int c0, c1, c2;                   //   and the first 32 recorded, with a second evaluation (a transient differs from it: DESIGN.md section 4.3)
                if (pair_bins_fast(pf, nf, qf, mf, &c0, &c1, &c2) && (c0 != b0 || c1 != b1 || c2 != b2)) {
                    const int at = atomicAdd(a.verify, 1);
                    if (at < 32) {
                        int d0, d1, d2; const bool again = pair_bins_fast(pf, nf, qf, mf, &d0, &d1, &d2);
                        float *r = reinterpret_cast<float *>(a.verify) + 16 + at * 32;
                        r[0] = pf.x; r[1] = pf.y; r[2] = pf.z; r[3] = nf.x; r[4] = nf.y; r[5] = nf.z; r[6] = qf.x; r[7] = qf.y; r[8] = qf.z; r[9] = mf.x; r[10] = mf.y; r[11] = mf.z;
                        r[12] = (float)b0; r[13] = (float)b1; r[14] = (float)b2; r[15] = (float)c0; r[16] = (float)c1; r[17] = (float)c2;
                        r[18] = again ? (float)d0 : -1.0f; r[19] = (float)d1; r[20] = (float)d2; r[21] = (float)qi; r[22] = (float)id; r[23] = (float)slot;
                    }
                }
            }
            atomicAdd(&hist[ob][b0], 1); atomicAdd(&hist[ob][b1], 1); atomicAdd(&hist[ob][b2], 1);
        }
    }
    cnt = pcr_octet_sum_i(cnt);
    __syncthreads();
    if (qi < n) {
        SpfhRow *row = a.spfh + qi;
        for (int b = ol; b < 40; b += OCT) row->bins[b] = b < 33 ? (uint8_t)hist[ob][b] : (uint8_t)0;
        if (ol == 0) row->inc = cnt > 0 ? 100.0 / (double)cnt : 0.0;       // 100 / (m - 1), m counts the point itself
    }
    }
}

__global__ void __launch_bounds__(FB) k_spfh(FpfhArgs a) { d_spfh(a); }
__device__ static inline void d_fpfh(const FpfhArgs &a) {
    const int n = *a.n_ptr;
    const int ol = threadIdx.x & 7, ob = threadIdx.x >> 3;
    const int qi = blockIdx.x * (FB / OCT) + ob;
    double acc[33];
#pragma unroll
    for (int j = 0; j < 33; j++) acc[j] = 0.0;
    int cnt = 0;
    if (qi < n) {
        const float4 pf = a.pts[qi];
        const int kk = a.cnt ? (a.cnt[qi] < 0 ? a.k : a.cnt[qi]) : a.k;
        for (int slot = ol; slot < kk; slot += OCT) {
            const int id = a.nbr[(size_t)qi * a.k + slot];
            if (id < 0 || id == qi) continue;
            const float4 qf = a.pts[id];
            const double dx = (double)qf.x - (double)pf.x, dy = (double)qf.y - (double)pf.y, dz = (double)qf.z - (double)pf.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (!(d2 < a.r2)) continue;
            cnt++;
            if (d2 == 0.0) continue;
            const double inv = 1.0 / d2;
            const uint4 *rw = reinterpret_cast<const uint4 *>(a.spfh + id);
            const uint4 w0 = rw[0], w1 = rw[1], w2 = rw[2];
            const unsigned bw[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x};
            union { unsigned u[2]; double d; } ic; ic.u[0] = w2.z; ic.u[1] = w2.w;
#pragma unroll
            for (int j = 0; j < 33; j++) {
                const double sj = (double)((bw[j >> 2] >> (8 * (j & 3))) & 0xffu) * ic.d;      // the SPFH value (d_spfh)
                acc[j] += sj * inv;
            }
        }
    }
    // octet sums by DPP (pcr_octet_sum: the same pairwise tree as a xor-1 / 2 / 4 butterfly, so the same bits): the butterfly by
    // __shfl_xor was 2 x 99 ds_bpermute round trips per wavefront, more than the gathers above
    cnt = pcr_octet_sum_i(cnt);
#pragma unroll
    for (int j = 0; j < 33; j++) acc[j] = pcr_octet_sum(acc[j]);
    if (qi < n && ol == 0) {
        float *out = a.feat + (size_t)a.perm[qi] * 33;
        if (cnt > 0) {
            double sum[3] = {0, 0, 0};
#pragma unroll
            for (int j = 0; j < 33; j++) sum[j / 11] += acc[j];
#pragma unroll
            for (int g = 0; g < 3; g++) if (sum[g] != 0.0) sum[g] = 100.0 / sum[g];
#pragma unroll
            for (int j = 0; j < 33; j++) out[j] = (float)(acc[j] * sum[j / 11] + (double)a.spfh[qi].bins[j] * a.spfh[qi].inc);
        } else {
#pragma unroll
            for (int j = 0; j < 33; j++) out[j] = 0.0f;
        }
    }
}

__global__ void __launch_bounds__(FB) k_fpfh(FpfhArgs a) { d_fpfh(a); }
// batch forms (lockstep FGR groups): blockIdx.y = cloud
__global__ void __launch_bounds__(FB) k_spfh_g(const FpfhArgs *a) { d_spfh(a[blockIdx.y]); }
__global__ void __launch_bounds__(FB) k_fpfh_g(const FpfhArgs *a) { d_fpfh(a[blockIdx.y]); }
// FPFH of an imported cloud (Morton-sorted points + normals + octree); rows of feat33 in CALLER order (perm: sorted -> caller)
static size_t fpfh_scratch_bytes(int64_t n, int knn) { return (size_t)(n > 0 ? n : 1) * ((size_t)knn * 8 + 33 * 8 + 64) + (1u << 16); }
static int fpfh_of_cloud(pcr_context *ctx, const DevCloud &c, const uint32_t *perm, int64_t n, int search_kind, int knn, double radius, float *feat33) {
    if (search_kind == PCR_SEARCH_RADIUS) { ctx->err = "compute_fpfh_feature: pure radius search not implemented (use Hybrid or KNN)"; return PCR_EINVAL; }
    if (knn < 1 || knn > 200) { ctx->err = "compute_fpfh_feature: max_nn must be in 1..200"; return PCR_EINVAL; }
    if (search_kind == PCR_SEARCH_HYBRID && !(radius > 0)) { ctx->err = "radius <= 0"; return PCR_EINVAL; }
    if (n == 0) return PCR_OK;
    ArenaMark mark(ctx);
    int32_t *nbr = arena<int32_t>(ctx, (size_t)n * knn);
    SpfhRow *spfh = arena<SpfhRow>(ctx, (size_t)n);
    if (!nbr || !spfh) return PCR_ENOMEM;
    int32_t *ncnt = nullptr;
    if (search_kind == PCR_SEARCH_HYBRID) {      // every point inside the ball, appended; the k nearest only where a ball is overfull
        ncnt = arena<int32_t>(ctx, n);
        if (!ncnt) return PCR_ENOMEM;
        const DevCloud *cp = &c;
        PCR_TRY(pcr_dev_radius_lists_batch(ctx, &cp, 1, knn, radius, &nbr, &ncnt));
    } else {
        float *nd2 = arena<float>(ctx, (size_t)n * knn);
        if (!nd2) return PCR_ENOMEM;
        PCR_TRY(pcr_dev_knn_debug(ctx, &c, knn, 0.0, nbr, nd2, nullptr));
    }
    FpfhArgs a;
    a.pts = c.pts; a.nrm = c.nrm; a.n_ptr = c.n; a.nbr = nbr; a.k = knn; a.cnt = ncnt;
    a.r2 = search_kind == PCR_SEARCH_HYBRID ? radius * radius : 1e300;
    // 0 product (float pass + float64 queue from SPFH_SPLIT_MIN_POINTS points, else all float64), 1 all float64, 2 float64 + comparison with the float form,
    // 3 float pass with a 16-entry queue (the overflow path), 4 float pass whatever the size
    const int spfh_mode = pcr_options().spfh_float64.load(std::memory_order_relaxed);
    a.spfh = spfh; a.perm = perm; a.feat = feat33; a.float64_only = (spfh_mode == 3 || spfh_mode == 4) ? 0 : (spfh_mode == 0 ? (n >= SPFH_SPLIT_MIN_POINTS ? 0 : 1) : spfh_mode);
    a.verify = nullptr;
    if (a.float64_only == 2) {
        a.verify = arena<int>(ctx, 16 + 32 * 32);
        if (!a.verify) return PCR_ENOMEM;
        PCR_HIP_CHECK(ctx, hipMemsetAsync(a.verify, 0, sizeof(int) * (16 + 32 * 32), ctx->stream));
    }
    const dim3 grid((unsigned)(((size_t)n * OCT + FB - 1) / FB));
    a.slowq = nullptr; a.slow_count = nullptr; a.slow_cap = 0; a.only_if_over = nullptr; a.over_cap = 0;
    if (a.float64_only == 0) {
        a.slow_cap = spfh_mode == 3 ? 16 : (int)std::min<int64_t>(8 * n, 1 << 28);
        a.slowq = arena<uint2>(ctx, (size_t)a.slow_cap); a.slow_count = arena<int>(ctx, 1);
        if (!a.slowq || !a.slow_count) return PCR_ENOMEM;
        PCR_HIP_CHECK(ctx, hipMemsetAsync(a.slow_count, 0, sizeof(int), ctx->stream));
        PCR_LAUNCH(ctx, k_spfh_fast, grid, dim3(FB), 0, ctx->stream, a);
        PCR_LAUNCH(ctx, k_spfh_slow, dim3(std::min<unsigned>(grid.x, 1024u)), dim3(FB), 0, ctx->stream, a);
        a.only_if_over = a.slow_count; a.over_cap = a.slow_cap; a.float64_only = 1;
        static const bool dbg_fgr = getenv("PCR_DEBUG_FGR") != nullptr;
        if (dbg_fgr) {
            int h = 0;
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(&h, a.slow_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            fprintf(stderr, "spfh: %lld points, %d pairs queued for the float64 pass (%.2f per point; queue capacity %d)\n", (long long)n, h, (double)h / (double)n, a.slow_cap);
        }
    }
    PCR_LAUNCH(ctx, k_spfh, a.only_if_over ? dim3(std::min<unsigned>(grid.x, 512u)) : grid, dim3(FB), 0, ctx->stream, a);
    PCR_LAUNCH(ctx, k_fpfh, grid, dim3(FB), 0, ctx->stream, a);
    if (a.verify) {
        std::vector<int> rec(16 + 32 * 32);
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(rec.data(), a.verify, sizeof(int) * rec.size(), hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const int bad = rec[0];
        for (int e = 0; e < bad && e < 32; e++) {
            const float *r = reinterpret_cast<const float *>(rec.data()) + 16 + e * 32;
            fprintf(stderr, "SPFHBAD");
            for (int j = 0; j < 24; j++) fprintf(stderr, " %.9g", r[j]);
            fprintf(stderr, "\n");
        }
        if (bad) { ctx->err = "spfh_float64 = 2: " + std::to_string(bad) + " pairs whose float bins differ from the float64 bins"; return PCR_EINVAL; }
    }
    return PCR_OK;
}

extern "C" int pcr_compute_fpfh_feature(pcr_context *ctx, const float *xyz, const float *normals, int64_t n, int search_kind, int knn,
                                        double radius, float *feat33) {
    return pcr_api_call(ctx, [&]() -> int {
    if (n < 0 || (n > 0 && (!xyz || !normals || !feat33))) return PCR_EINVAL;
    if (n == 0) return PCR_OK;
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n) + fpfh_scratch_bytes(n, knn > 0 && knn <= 200 ? knn : 1)));
    DevCloud c; uint32_t *perm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, xyz, normals, n, &c, &perm, false));
    return fpfh_of_cloud(ctx, c, perm, n, search_kind, knn, radius, feat33);      // no scalar output: asynchronous on the context's stream
    });
}

// ======================================================================== feature matching (K7)
// Exact nearest row in 33-D, (a-b)^2 evaluated directly in float32 (no |a|^2+|b|^2-2ab cancellation), ties -> the
// smaller index.  One query per lane, database rows staged through LDS in tiles and broadcast to all lanes.
#define FEAT_D 33
#define NN_TILE 64
__global__ void __launch_bounds__(FB) k_feature_nn(const float *__restrict__ db, int n_db, const float *__restrict__ q, int n_q, int32_t *__restrict__ out) {
    __shared__ float tile[NN_TILE][FEAT_D + 3];
    const int qi = blockIdx.x * FB + threadIdx.x;
    float f[FEAT_D];
#pragma unroll
    for (int j = 0; j < FEAT_D; j++) f[j] = qi < n_q ? q[(size_t)qi * FEAT_D + j] : 0.0f;
    float best = 3.4e38f; int bi = -1;
    for (int base = 0; base < n_db; base += NN_TILE) {
        __syncthreads();
        for (int t = threadIdx.x; t < NN_TILE * FEAT_D; t += FB) {
            const int r = t / FEAT_D, cidx = t % FEAT_D;
            tile[r][cidx] = base + r < n_db ? db[(size_t)(base + r) * FEAT_D + cidx] : 0.0f;
        }
        __syncthreads();
        const int lim = n_db - base < NN_TILE ? n_db - base : NN_TILE;
        for (int r = 0; r < lim; r++) {
            float d = 0.0f;
#pragma unroll
            for (int j = 0; j < FEAT_D; j++) { const float e = f[j] - tile[r][j]; d = __fmaf_rn(e, e, d); }
            if (d < best) { best = d; bi = base + r; }
        }
    }
    if (qi < n_q) out[qi] = bi;
}

// ---- MFMA path (the 33-D distance contraction is the one GEMM-shaped step of the pipeline) -------------------
// d2(q, b) = |q|^2 + |b|^2 - 2 q.b with S = B_db * Q^T on v_mfma_f64_16x16x4_f64.  In float32 the expanded form
// cancels (|f|^2 ~ 1e5 against d2 ~ 1) and dense feature sets have many neighbours inside the rounding margin, so the
// contraction runs in FLOAT64: exact to ~1e-11 on these inputs, i.e. the true nearest row of the float32 features
// (what Open3D's float64 kd-tree returns), ties -> smaller index.  Rows = 16 database points, columns = 16 queries per
// MFMA; a wavefront keeps 4 column blocks (64 queries) in registers and reuses every database operand 4 times; a
// lane owns one query column per block and 4 database rows, so the running arg-min needs no cross-lane traffic.
#define FK 36                      // 33 padded to a multiple of 4
#define QB 4                       // query blocks of 16 per wavefront
typedef double f64x4 __attribute__((ext_vector_type(4)));

// FT[k][n_pad] (k-major, zero padded, float64) + squared norms (padded rows: +huge, can never win)
__global__ void __launch_bounds__(FB) k_feat_transpose(const float *__restrict__ f, int n, int n_pad, double *__restrict__ ft, double *__restrict__ nrm2) {
    const int i = blockIdx.x * FB + threadIdx.x;
    if (i >= n_pad) return;
    double s = 0.0;
    for (int k = 0; k < FK; k++) {
        const double v = (i < n && k < FEAT_D) ? (double)f[(size_t)i * FEAT_D + k] : 0.0;
        ft[(size_t)k * n_pad + i] = v;
        s += v * v;
    }
    nrm2[i] = i < n ? s : 1.0e300;
}

__global__ void __launch_bounds__(FB) k_feature_nn_mfma(const double *__restrict__ dbT, const double *__restrict__ dbn, int n_db_pad,
                                                        const double *__restrict__ qT, const double *__restrict__ qn, int n_q_pad,
                                                        int tiles_per_split, double *__restrict__ cand_d, int *__restrict__ cand_i) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int q0 = (blockIdx.x * (FB / 64) + wv) * (16 * QB);
    if (q0 >= n_q_pad) return;
    double qb[QB][FK / 4], nq[QB];
#pragma unroll
    for (int b = 0; b < QB; b++) {
#pragma unroll
        for (int t = 0; t < FK / 4; t++) qb[b][t] = qT[(size_t)(4 * t + kk) * n_q_pad + q0 + 16 * b + col];    // B[k = 4t + kk][col]
        nq[b] = qn[q0 + 16 * b + col];
    }
    double bd[QB]; int bi[QB];
#pragma unroll
    for (int b = 0; b < QB; b++) { bd[b] = 1.0e300; bi[b] = -1; }
    const int n_tiles = n_db_pad / 16;
    const int t0 = blockIdx.y * tiles_per_split, t1 = min(n_tiles, t0 + tiles_per_split);
    // software pipeline: operands of tile t+1 are in flight while the MFMAs of tile t issue
    double av[FK / 4], an[FK / 4], nbv[4], nbn[4];
    if (t0 < t1) {
#pragma unroll
        for (int t = 0; t < FK / 4; t++) av[t] = dbT[(size_t)(4 * t + kk) * n_db_pad + t0 * 16 + col];           // A[row = col][k = 4t + kk]
#pragma unroll
        for (int r = 0; r < 4; r++) nbv[r] = dbn[t0 * 16 + kk + 4 * r];                                        // result rows kk + 4r
    }
    for (int tile = t0; tile < t1; tile++) {
        const int r0 = tile * 16, rn = (tile + 1 < t1 ? tile + 1 : tile) * 16;
#pragma unroll
        for (int t = 0; t < FK / 4; t++) an[t] = dbT[(size_t)(4 * t + kk) * n_db_pad + rn + col];
#pragma unroll
        for (int r = 0; r < 4; r++) nbn[r] = dbn[rn + kk + 4 * r];
#pragma unroll
        for (int b = 0; b < QB; b++) {
            f64x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < FK / 4; t++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], qb[b][t], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double v = (nq[b] + nbv[r]) - 2.0 * acc[r];
                const int row = r0 + kk + 4 * r;
                if (v < bd[b]) { bd[b] = v; bi[b] = row; }       // rows ascend within a lane: first (smallest) index kept on ties
            }
        }
#pragma unroll
        for (int t = 0; t < FK / 4; t++) av[t] = an[t];
#pragma unroll
        for (int r = 0; r < 4; r++) nbv[r] = nbn[r];
    }
    // combine the 4 row groups (lanes col, col+16, col+32, col+48): lexicographic (distance, index) minimum
#pragma unroll
    for (int b = 0; b < QB; b++) {
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const double od = __shfl_xor(bd[b], o, 64); const int oi = __shfl_xor(bi[b], o, 64);
            if (od < bd[b] || (od == bd[b] && oi >= 0 && (bi[b] < 0 || oi < bi[b]))) { bd[b] = od; bi[b] = oi; }
        }
        if (kk == 0) {
            const size_t o = (size_t)(q0 + 16 * b + col) * gridDim.y + blockIdx.y;
            cand_d[o] = bd[b]; cand_i[o] = bi[b];
        }
    }
}
__global__ void __launch_bounds__(FB) k_feature_nn_pick(const double *__restrict__ cand_d, const int *__restrict__ cand_i, int n_q, int splits, int32_t *__restrict__ out) {
    const int qi = blockIdx.x * FB + threadIdx.x;
    if (qi >= n_q) return;
    double best = 1.0e300; int bi = -1;
    for (int sidx = 0; sidx < splits; sidx++) {          // splits cover ascending row ranges: strict '<' keeps the smaller index on ties
        const double d = cand_d[(size_t)qi * splits + sidx]; const int id = cand_i[(size_t)qi * splits + sidx];
        if (id >= 0 && d < best) { best = d; bi = id; }
    }
    out[qi] = bi;
}

// nearest database row for every query row (exact, float64 contraction on the matrix cores)
static int feature_nn(pcr_context *ctx, const float *db, int n_db, const float *q, int n_q, int32_t *out) {
    if (n_db < 64 || n_q < 64 || getenv("PCR_FEATURE_NN_BRUTE")) {
        PCR_LAUNCH(ctx, k_feature_nn, dim3((n_q + FB - 1) / FB), dim3(FB), 0, ctx->stream, db, n_db, q, n_q, out);
        return PCR_OK;
    }
    ArenaMark mark(ctx);
    const int ndp = (n_db + 15) / 16 * 16, nqp = (n_q + 255) / 256 * 256;
    const int n_tiles = ndp / 16, waves = nqp / (16 * QB);
    int splits = (4096 + waves - 1) / waves;
    if (splits > 32) splits = 32;
    if (splits > n_tiles) splits = n_tiles;
    if (splits < 1) splits = 1;
    const int tps = (n_tiles + splits - 1) / splits;
    double *dbT = arena<double>(ctx, (size_t)FK * ndp), *dbn = arena<double>(ctx, ndp);
    double *qT = arena<double>(ctx, (size_t)FK * nqp), *qn = arena<double>(ctx, nqp);
    double *cd = arena<double>(ctx, (size_t)nqp * splits); int *ci = arena<int>(ctx, (size_t)nqp * splits);
    if (!dbT || !dbn || !qT || !qn || !cd || !ci) return PCR_ENOMEM;
    PCR_LAUNCH(ctx, k_feat_transpose, dim3((ndp + FB - 1) / FB), dim3(FB), 0, ctx->stream, db, n_db, ndp, dbT, dbn);
    PCR_LAUNCH(ctx, k_feat_transpose, dim3((nqp + FB - 1) / FB), dim3(FB), 0, ctx->stream, q, n_q, nqp, qT, qn);
    PCR_LAUNCH(ctx, k_feature_nn_mfma, dim3(nqp / 256, splits), dim3(FB), 0, ctx->stream, dbT, dbn, ndp, qT, qn, nqp, tps, cd, ci);
    PCR_LAUNCH(ctx, k_feature_nn_pick, dim3((n_q + FB - 1) / FB), dim3(FB), 0, ctx->stream, cd, ci, n_q, splits, out);
    return PCR_OK;
}

// test hook: the mutual nearest-feature search alone.  mode 0: f16-split screen + exact re-check (the production path; tile pruning
// from ~70k rows per side), 1: all-pairs float64 MFMA, 2: float32 brute force, 3: the screen with tile pruning forced on, 4: forced off
extern "C" int pcr_debug_feature_nn(pcr_context *ctx, const float *f0, int64_t n0, const float *f1, int64_t n1, int32_t *out_1to0, int32_t *out_0to1, int mode) {
    return pcr_api_call(ctx, [&]() -> int {
        if (n0 <= 0 || n1 <= 0 || !f0 || !f1 || !out_1to0 || !out_0to1) return PCR_EINVAL;
        PCR_TRY(pcr_arena_reserve(ctx, pcr_feature_nn_scratch_bytes(n0, n1, mode == 3 ? 1 : -1) + (size_t)(n0 + n1) * (FK * 16 + 32 * 12 + 64) + (64u << 20)));
        if (mode == 0 || mode == 3 || mode == 4) return pcr_feature_nn_mutual(ctx, f0, (int)n0, f1, (int)n1, out_1to0, out_0to1, mode == 0 ? -1 : (mode == 3 ? 1 : 0));
        if (mode == 2) {
            PCR_LAUNCH(ctx, k_feature_nn, dim3((unsigned)((n1 + FB - 1) / FB)), dim3(FB), 0, ctx->stream, f0, (int)n0, f1, (int)n1, out_1to0);
            PCR_LAUNCH(ctx, k_feature_nn, dim3((unsigned)((n0 + FB - 1) / FB)), dim3(FB), 0, ctx->stream, f1, (int)n1, f0, (int)n0, out_0to1);
            return PCR_OK;
        }
        PCR_TRY(feature_nn(ctx, f0, (int)n0, f1, (int)n1, out_1to0));
        return feature_nn(ctx, f1, (int)n1, f0, (int)n0, out_0to1);
    });
}

// cross check: pair (i, i_to_j[i]) survives iff j_to_i[i_to_j[i]] == i      
__global__ void __launch_bounds__(FB) k_cross_flags(const int32_t *__restrict__ i_to_j, const int32_t *__restrict__ j_to_i, int n_i, uint8_t *__restrict__ flags) {
    const int i = blockIdx.x * FB + threadIdx.x;
    if (i >= n_i) return;
    const int j = i_to_j[i];
    flags[i] = (j >= 0 && j_to_i[j] == i) ? 1 : 0;
}
__global__ void __launch_bounds__(FB) k_cross_emit(const int32_t *__restrict__ i_to_j, const uint8_t *__restrict__ flags, const int *__restrict__ pos, int n_i, int32_t *__restrict__ cross) {
    const int i = blockIdx.x * FB + threadIdx.x;
    if (i >= n_i || !flags[i]) return;
    cross[2 * (size_t)pos[i]] = i; cross[2 * (size_t)pos[i] + 1] = i_to_j[i];
}

// ======================================================================== tuple test (K8)
__host__ __device__ static inline uint64_t pcr_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct TupleArgs {
    const double *pi, *pj;          // normalised clouds i (larger) and j, n x 3 float64
    const int32_t *cross; int ncross;
    uint64_t seed; double tuple_scale; long long trials;
};
__device__ static inline bool tuple_ok(const TupleArgs &a, long long t, int *r) {
    const double *pi[3], *pj[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r[k] = (int)(pcr_splitmix64(a.seed + 3ull * (uint64_t)t + (uint64_t)k) % (uint64_t)a.ncross);
        pi[k] = a.pi + (size_t)a.cross[2 * r[k]] * 3; pj[k] = a.pj + (size_t)a.cross[2 * r[k] + 1] * 3;
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double *p = pi[k], *q = pi[(k + 1) % 3], *u = pj[k], *v = pj[(k + 1) % 3];
        const double li = sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]));
        const double lj = sqrt((u[0] - v[0]) * (u[0] - v[0]) + (u[1] - v[1]) * (u[1] - v[1]) + (u[2] - v[2]) * (u[2] - v[2]));
        ok = ok && (li * a.tuple_scale < lj) && (lj < li / a.tuple_scale);
    }
    return ok;
}
__global__ void __launch_bounds__(FB) k_tuple_flags(TupleArgs a, uint8_t *__restrict__ flags) {
    const long long t = (long long)blockIdx.x * FB + threadIdx.x;
    if (t >= a.trials) return;
    int r[3];
    flags[t] = tuple_ok(a, t, r) ? 1 : 0;
}
// keep the first max_tuples accepted trials IN TRIAL ORDER (what the serial loop of the reference does)
__global__ void __launch_bounds__(FB) k_tuple_emit(TupleArgs a, const uint8_t *__restrict__ flags, const int *__restrict__ pos, int max_tuples, int swapped,
                                                   int32_t *__restrict__ corr /* (cloud0 idx, cloud1 idx) x 3 per tuple */) {
    const long long t = (long long)blockIdx.x * FB + threadIdx.x;
    if (t >= a.trials || !flags[t]) return;
    const int rank = pos[t];
    if (rank >= max_tuples) return;
    int r[3];
    (void)tuple_ok(a, t, r);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int ii = a.cross[2 * r[k]], jj = a.cross[2 * r[k] + 1];
        corr[((size_t)rank * 3 + k) * 2] = swapped ? jj : ii;
        corr[((size_t)rank * 3 + k) * 2 + 1] = swapped ? ii : jj;
    }
}

// ======================================================================== normalisation
__global__ void __launch_bounds__(FB) k_sum3(const float *__restrict__ xyz, int n, double *__restrict__ part) {
    double s[3] = {0, 0, 0};
    for (int i = blockIdx.x * FB + threadIdx.x; i < n; i += gridDim.x * FB) { s[0] += xyz[i * 3]; s[1] += xyz[i * 3 + 1]; s[2] += xyz[i * 3 + 2]; }
    __shared__ double sh[FB / PCR_WAVE][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; d++) { const double v = pcr_wave_sum(s[d]); if (lane == 0) sh[w][d] = v; }
    __syncthreads();
    if (threadIdx.x < 3) { double v = 0; for (int k = 0; k < FB / PCR_WAVE; k++) v += sh[k][threadIdx.x]; part[blockIdx.x * 3 + threadIdx.x] = v; }
}
__global__ void k_sum3_final(const double *__restrict__ part, int nb, int n, double *__restrict__ mean3) {
    if (threadIdx.x < 3) { double v = 0; for (int k = 0; k < nb; k++) v += part[k * 3 + threadIdx.x]; mean3[threadIdx.x] = n > 0 ? v / (double)n : 0.0; }
}
// centred float64 copy + max norm (per block)
__global__ void __launch_bounds__(FB) k_center(const float *__restrict__ xyz, int n, const double *__restrict__ mean3, double *__restrict__ out, double *__restrict__ part_max) {
    double mx = 0;
    for (int i = blockIdx.x * FB + threadIdx.x; i < n; i += gridDim.x * FB) {
        const double x = (double)xyz[i * 3] - mean3[0], y = (double)xyz[i * 3 + 1] - mean3[1], z = (double)xyz[i * 3 + 2] - mean3[2];
        out[(size_t)i * 3] = x; out[(size_t)i * 3 + 1] = y; out[(size_t)i * 3 + 2] = z;
        mx = fmax(mx, sqrt(x * x + y * y + z * z));
    }
    __shared__ double sh[FB];
    sh[threadIdx.x] = mx;
    __syncthreads();
    for (int o = FB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) part_max[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(FB) k_scale(double *__restrict__ p, long long n3, double inv) {
    const long long i = (long long)blockIdx.x * FB + threadIdx.x;
    if (i < n3) p[i] = p[i] / inv;
}

// ======================================================================== GNC Gauss-Newton (K9)
#define FNV 27
#define FNVP 32
struct FgrState {
    double trans[16];
    double par;
    unsigned int ticket;
    int itr;
    unsigned long long t_dbg[4];     // diagnostics (PCR_DEBUG_FGR), 100 MHz ticks summed over the iterations of the single-workgroup kernel
    unsigned int arrive;             // k_fgr_opt_multi: workgroups arrived, summed over the iterations (monotonic barrier counter)
    int failed;                      // k_fgr_opt_multi: a workgroup gave up waiting (co-residency not granted): the host reruns with one launch per iteration
};
struct FgrOptArgs {
    const double *pq; int stride;    // the correspondences' points gathered ONCE, structure of arrays: pq[k * stride + c], k = 0..2 the
                                     //   cloud-0 (source) point, k = 3..5 the cloud-1 (target) point of correspondence c -- the 300
                                     //   iterations then stream coalesced float64 columns instead of chasing corr[] -> point twice each
    int ncorr;
    FgrState *st; double *partials;
    int decrease_mu; double max_corr_dist, division_factor;
};
__global__ void __launch_bounds__(FB) k_fgr_gather_pairs(const double *__restrict__ p0, const double *__restrict__ q0, const int32_t *__restrict__ corr, int ncorr, int stride,
                                                         double *__restrict__ pq) {
    const int c = blockIdx.x * FB + threadIdx.x;
    if (c >= ncorr) return;
    const double *p = p0 + (size_t)corr[2 * c] * 3, *q = q0 + (size_t)corr[2 * c + 1] * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) { pq[(size_t)k * stride + c] = p[k]; pq[(size_t)(3 + k) * stride + c] = q[k]; }
}
__global__ void k_fgr_init(FgrState *st, double par) {
    if (threadIdx.x == 0) {
        for (int k = 0; k < 16; k++) st->trans[k] = (k % 5 == 0) ? 1.0 : 0.0;
        st->par = par; st->ticket = 0; st->itr = 0;
        for (int k = 0; k < 4; k++) st->t_dbg[k] = 0;
        st->arrive = 0; st->failed = 0;
    }
}
// 6x6 LDL^T without pivoting in registers (as in pcr_gicp.hip); the system here is -JTJ x = JTr
__device__ static bool fgr_solve6(const double *S, const double *b, double *x) {
    double A[6][6];
    { int t = 0;
#pragma unroll
      for (int p = 0; p < 6; p++)
#pragma unroll
          for (int q = p; q < 6; q++) { A[q][p] = S[t]; t++; } }
    double D[6], y[6];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double d = A[k][k];
        ok = ok && (d > 0.0) && isfinite(d);
        D[k] = d;
        const double inv = 1.0 / d;
        double col[6];
#pragma unroll
        for (int i = k + 1; i < 6; i++) col[i] = A[i][k];
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            const double l = col[i] * inv;
#pragma unroll
            for (int j = k + 1; j <= i; j++) A[i][j] -= l * col[j];
            A[i][k] = l;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { double s = b[i];
#pragma unroll
        for (int j = 0; j < i; j++) s -= A[i][j] * y[j];
        y[i] = s; }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] /= D[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) { double s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; j++) s -= A[j][i] * x[j];
        x[i] = s; }
#pragma unroll
    for (int i = 0; i < 6; i++) ok = ok && isfinite(x[i]);
    return ok;
}

// one correspondence of the GNC objective.  The three Jacobian rows are [-[q]x | -I] (q = moved target point), so
// s J^T J and s J^T r have only FC = 16 distinct sums: s q q^T (6), s q (3), s (1) and the six entries of s J^T r.  Accumulating
// those instead of the 27 generic products is 3x fewer float64 operations per correspondence (the generic form multiplies
// the structural zeros: 0 * x cannot be folded without fast-math); fgr_expand() lays them out as the 21 + 6 sums the
// solver takes.
#define FC 16
__device__ static inline void fgr_load(const FgrOptArgs &a, int c, double *v /*6*/) {
    const double *col = a.pq + c;
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = col[(size_t)k * a.stride];
}
__device__ static inline void fgr_math(const double *T, double par, const double *v /*p, q0*/, double *acc /*FC*/) {
    const double qx = T[0] * v[3] + T[1] * v[4] + T[2] * v[5] + T[3];
    const double qy = T[4] * v[3] + T[5] * v[4] + T[6] * v[5] + T[7];
    const double qz = T[8] * v[3] + T[9] * v[4] + T[10] * v[5] + T[11];
    const double rx = v[0] - qx, ry = v[1] - qy, rz = v[2] - qz;
    const double temp = par / (rx * rx + ry * ry + rz * rz + par), s = temp * temp;
    const double sx = s * qx, sy = s * qy, sz = s * qz;
    acc[0] += sx * qx; acc[1] += sy * qy; acc[2] += sz * qz;
    acc[3] += sx * qy; acc[4] += sx * qz; acc[5] += sy * qz;
    acc[6] += sx; acc[7] += sy; acc[8] += sz; acc[9] += s;
    acc[10] += sz * ry - sy * rz;          // s (q x r) with the sign of J^T r: rows [0,-qz,qy], [qz,0,-qx], [-qy,qx,0]
    acc[11] += sx * rz - sz * rx;
    acc[12] += sy * rx - sx * ry;
    acc[13] -= s * rx; acc[14] -= s * ry; acc[15] -= s * rz;
}
// correspondences first, first + step, ... < ncorr: four of them in flight per lane (24 independent coalesced loads, then the math)
__device__ static inline void fgr_accumulate_all(const FgrOptArgs &a, const double *T, double par, int first, int step, double *acc /*FC*/) {
    for (int c0 = first; c0 < a.ncorr; c0 += 4 * step) {
        double v[4][6];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = c0 + u * step; fgr_load(a, c < a.ncorr ? c : c0, v[u]); }
#pragma unroll
        for (int u = 0; u < 4; u++) if (c0 + u * step < a.ncorr) fgr_math(T, par, v[u], acc);
    }
}
// FC sums -> upper triangle of J^T J (21, row-major) followed by J^T r (6)
__device__ static inline void fgr_expand(const double *c, double *S /*27*/) {
    const double A = c[0], B = c[1], C = c[2], D = c[3], E = c[4], F = c[5], G = c[6], H = c[7], I = c[8], W = c[9];
    S[0] = B + C; S[1] = -D; S[2] = -E; S[3] = 0; S[4] = -I; S[5] = H;
    S[6] = A + C; S[7] = -F; S[8] = I; S[9] = 0; S[10] = -G;
    S[11] = A + B; S[12] = -H; S[13] = G; S[14] = 0;
    S[15] = W; S[16] = 0; S[17] = 0; S[18] = W; S[19] = 0; S[20] = W;
#pragma unroll
    for (int k = 0; k < 6; k++) S[21 + k] = c[10 + k];
}
// solve the 6x6 system of one iteration and left-multiply the pose (Open3D: SolveLinearSystemPSD(-JTJ, JTr) == JTJ x = -JTr)
__device__ static inline void fgr_update(const double *S, double *trans /*16*/) {
    double nb6[6], x[6];
#pragma unroll
    for (int p = 0; p < 6; p++) nb6[p] = -S[21 + p];
    double U[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (fgr_solve6(S, nb6, x)) {
        double ca, sa, cb, sb, cg, sg;
        sincos(x[0], &sa, &ca); sincos(x[1], &sb, &cb); sincos(x[2], &sg, &cg);
        U[0] = cg * cb; U[1] = cg * sb * sa - sg * ca; U[2] = cg * sb * ca + sg * sa; U[3] = x[3];
        U[4] = sg * cb; U[5] = sg * sb * sa + cg * ca; U[6] = sg * sb * ca - cg * sa; U[7] = x[4];
        U[8] = -sb;     U[9] = cb * sa;                U[10] = cb * ca;               U[11] = x[5];
    }
    double Tn[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { double s = 0; for (int k = 0; k < 4; k++) s += U[r * 4 + k] * trans[k * 4 + c]; Tn[r * 4 + c] = s; }
    for (int k = 0; k < 16; k++) trans[k] = Tn[k];
}

// ---- small correspondence sets (NCLT-size clouds): ALL iterations in ONE launch of one 1024-thread workgroup -- the
// multi-workgroup version costs a kernel launch per iteration (300 of them), and launches are what bounds the throughput
// NT threads: 1024, or 256 for up to FGR_SINGLE_SMALL correspondences (NCLT-size pairs have ~850: 4 wavefronts synchronise faster than 16
// and their 16 partial rows are summed by one DPP row each -- 4.8 -> 3.x us per iteration of the 300)
#define FSB 1024
#define FGR_SINGLE_SMALL 2048
template <int NT>
__device__ static inline void d_fgr_opt_single(const FgrOptArgs &a, int iterations) {
    __shared__ double red[NT / 16][FNVP];
    __shared__ double S[FNVP];
    __shared__ double trans[16];
    __shared__ double par_s;
    FgrState *st = a.st;
    if (threadIdx.x < 16) trans[threadIdx.x] = st->trans[threadIdx.x];
    if (threadIdx.x == 0) par_s = st->par;
    __syncthreads();
    int itr = st->itr;
    unsigned long long tA = 0, tB = 0, tC = 0;
    for (int it = 0; it < iterations; it++) {
        const unsigned long long t0 = wall_clock64();
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = trans[k];
        const double par = par_s;
        double acc[FC];
#pragma unroll
        for (int k = 0; k < FC; k++) acc[k] = 0.0;
        fgr_accumulate_all(a, T, par, threadIdx.x, NT, acc);
        const unsigned long long t1 = wall_clock64();
#pragma unroll
        for (int k = 0; k < FC; k++) { const double s = pcr_row16_sum(acc[k]); if ((threadIdx.x & 15) == 0) red[threadIdx.x >> 4][k] = s; }
        __syncthreads();
        // the NT / 16 partial rows: sum k is taken by the 16 lanes of DPP row k (lane r0 adds rows r0, r0 + 16, ..., then one row sum)
        // instead of one thread walking all rows (64 dependent LDS reads: 2.4 us of the iteration)
        if (threadIdx.x < 16 * FC) {
            const int k = threadIdx.x >> 4, r0 = threadIdx.x & 15;
            double s = 0;
#pragma unroll
            for (int r = r0; r < NT / 16; r += 16) s += red[r][k];
            s = pcr_row16_sum(s);
            if (r0 == 0) S[k] = s;
        }
        __syncthreads();
        const unsigned long long t2 = wall_clock64();
        tA += t1 - t0; tB += t2 - t1;
        if (threadIdx.x == 0) {
            double S27[FNV];
            fgr_expand(S, S27);
            fgr_update(S27, trans);
            if (a.decrease_mu && (itr % 4 == 0) && par_s > a.max_corr_dist) par_s = par_s / a.division_factor;
        }
        itr++;
        __syncthreads();
        tC += wall_clock64() - t2;
    }
    if (threadIdx.x < 16) st->trans[threadIdx.x] = trans[threadIdx.x];
    if (threadIdx.x == 0) { st->par = par_s; st->itr = itr; st->t_dbg[0] = tA; st->t_dbg[1] = tB; st->t_dbg[2] = tC; }
}
template <int NT> __global__ void __launch_bounds__(NT) k_fgr_opt_single(FgrOptArgs a, int iterations) { d_fgr_opt_single<NT>(a, iterations); }
template <int NT> __global__ void __launch_bounds__(NT) k_fgr_opt_single_g(const FgrOptArgs *a, int iterations) { d_fgr_opt_single<NT>(a[blockIdx.y], iterations); }

// ---- mid-size correspondence sets: ALL iterations in ONE launch of FMG co-resident workgroups.  Per iteration every
// workgroup publishes its row of FC sums (write-through), arrives at a monotonic counter, waits until all FMG have arrived,
// then gathers the FMG rows in fixed order and solves the 6x6 system ITSELF (1.4 us, redundantly: bit-identical poses in every
// workgroup, and one synchronisation per iteration instead of two).  Rows are double-buffered by iteration parity (a
// workgroup can be at most one barrier ahead).  FMG = 8 workgroups of 512 threads always fit next to anything else; should
// they nevertheless not become co-resident, every waiter gives up after `timeout_ticks` (50 ms), sets `failed`, and the host falls
// back to one launch per iteration -- every wave reaches an exit.
#define FMG 8                 // workgroups of the multi-workgroup form up to FGR_MULTI_WIDE correspondences (and of every pair of a lockstep group)
#define FMG_MAX 32            // ... above (200k-point pairs: ~120k correspondences, 29 per thread and iteration with 8 workgroups, 7 with 32); rows of absent workgroups are zeros
#ifndef FGR_MULTI_WIDE
#define FGR_MULTI_WIDE 50000
#endif
#define FMB 512
__device__ static inline void d_fgr_opt_multi(const FgrOptArgs &a, int iterations, double *rows /* 2 x FMG x FNVP */, unsigned long long timeout_ticks) {
    __shared__ double red[FMB / 16][FC];
    __shared__ double S[FC];
    __shared__ double trans[16];
    __shared__ double par_s;
    __shared__ int bail;
    FgrState *st = a.st;
    const int G = gridDim.x;
    if (threadIdx.x < 16) trans[threadIdx.x] = st->trans[threadIdx.x];
    if (threadIdx.x == 0) { par_s = st->par; bail = 0; }
    __syncthreads();
    int itr = st->itr;
    for (int it = 0; it < iterations; it++) {
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = trans[k];
        const double par = par_s;
        double acc[FC];
#pragma unroll
        for (int k = 0; k < FC; k++) acc[k] = 0.0;
        fgr_accumulate_all(a, T, par, blockIdx.x * FMB + threadIdx.x, G * FMB, acc);
#pragma unroll
        for (int k = 0; k < FC; k++) { const double s = pcr_row16_sum(acc[k]); if ((threadIdx.x & 15) == 0) red[threadIdx.x >> 4][k] = s; }
        __syncthreads();
        double *buf = rows + (size_t)(it & 1) * FMG_MAX * FNVP;
        if (threadIdx.x < FC) {
            double s = 0;
            for (int r = 0; r < FMB / 16; r++) s += red[r][threadIdx.x];
            __hip_atomic_store(&buf[(size_t)blockIdx.x * FNVP + threadIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the row is out before the workgroup arrives
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&st->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int want = (unsigned int)(it + 1) * (unsigned int)G;
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(&st->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (wall_clock64() - t0 > timeout_ticks || __hip_atomic_load(&st->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bail = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (bail) {
            if (threadIdx.x == 0) __hip_atomic_store(&st->failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // one coherent load per lane (the compiler issues relaxed-atomic loads one at a time: a loop of FMG of them is FMG round trips)
        if (threadIdx.x < FMG_MAX * FC) {
            const int b = threadIdx.x / FC, col = threadIdx.x % FC;
            red[b][col] = b < G ? __hip_atomic_load(&buf[(size_t)b * FNVP + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
        __syncthreads();
        if (threadIdx.x < FC) {
            double s = 0;
            for (int b = 0; b < FMG_MAX; b++) s += red[b][threadIdx.x];      // (fixed order; s + 0.0 is s: the 8-workgroup form sums as it always did)
            S[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double S27[FNV];
            fgr_expand(S, S27);
            fgr_update(S27, trans);
            if (a.decrease_mu && (itr % 4 == 0) && par_s > a.max_corr_dist) par_s = par_s / a.division_factor;
        }
        itr++;
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < 16) st->trans[threadIdx.x] = trans[threadIdx.x];
        if (threadIdx.x == 0) { st->par = par_s; st->itr = itr; }
    }
}
__global__ void __launch_bounds__(FMB) k_fgr_opt_multi(FgrOptArgs a, int iterations, double *rows, unsigned long long timeout_ticks) { d_fgr_opt_multi(a, iterations, rows, timeout_ticks); }
// batch form: blockIdx.y = pair, every pair with its own state, barrier counter and row buffers (FMG x pairs co-resident workgroups)
__global__ void __launch_bounds__(FMB) k_fgr_opt_multi_g(const FgrOptArgs *a, int iterations, double *rows, unsigned long long timeout_ticks) {
    d_fgr_opt_multi(a[blockIdx.y], iterations, rows + (size_t)blockIdx.y * 2 * FMG_MAX * FNVP, timeout_ticks);
}

__global__ void __launch_bounds__(FB) k_fgr_iter(FgrOptArgs a) {
    __shared__ double red[FB / PCR_WAVE][FNVP];
    __shared__ double fin[8][FNVP];
    __shared__ int is_last;
    FgrState *st = a.st;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = st->trans[k];
    const double par = st->par;
    double acc[FC];
#pragma unroll
    for (int k = 0; k < FC; k++) acc[k] = 0.0;
    fgr_accumulate_all(a, T, par, blockIdx.x * FB + threadIdx.x, gridDim.x * FB, acc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < FC; k++) { const double s = pcr_wave_sum(acc[k]); if (lane == 0) red[wv][k] = s; }
    __syncthreads();
    if (threadIdx.x < FC) {
        double s = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < FB / PCR_WAVE; w++) s += red[w][threadIdx.x];
        __hip_atomic_store(&a.partials[(size_t)blockIdx.x * FNVP + threadIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1);
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    {
        const int vcol = threadIdx.x & 31, chunk = threadIdx.x >> 5;      // 8 chunks x 32 columns; chunk c <- rows c, c+8, ...
        const int nb = gridDim.x;
        double s = 0;
        for (int b0 = 0; b0 < nb; b0 += 64) {          // sc1 loads (coherent at agent scope without an acquire fence), 8 in flight
            const double *p[8]; double v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) p[r] = a.partials + (size_t)(b0 + chunk + 8 * r < nb ? b0 + chunk + 8 * r : 0) * FNVP + vcol;
            asm volatile("global_load_dwordx2 %0, %8, off sc0 sc1\n\tglobal_load_dwordx2 %1, %9, off sc0 sc1\n\t"
                         "global_load_dwordx2 %2, %10, off sc0 sc1\n\tglobal_load_dwordx2 %3, %11, off sc0 sc1\n\t"
                         "global_load_dwordx2 %4, %12, off sc0 sc1\n\tglobal_load_dwordx2 %5, %13, off sc0 sc1\n\t"
                         "global_load_dwordx2 %6, %14, off sc0 sc1\n\tglobal_load_dwordx2 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
#pragma unroll
            for (int r = 0; r < 8; r++) if (vcol < FC && b0 + chunk + 8 * r < nb) s += v[r];
        }
        fin[chunk][vcol] = s;
    }
    __syncthreads();
    if (threadIdx.x < FNVP) {
        double s = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) s += fin[c][threadIdx.x];
        fin[0][threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tr[16], S27[FNV];
        for (int k = 0; k < 16; k++) tr[k] = st->trans[k];
        fgr_expand(fin[0], S27);
        fgr_update(S27, tr);
        for (int k = 0; k < 16; k++) st->trans[k] = tr[k];
        if (a.decrease_mu && (st->itr % 4 == 0) && st->par > a.max_corr_dist) st->par = st->par / a.division_factor;
        st->itr = st->itr + 1;
        __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Which form of the GNC optimiser a correspondence count takes (one rule for the one-pair call and for lockstep groups: a pair must meet the
// same variant -- the same summation order -- in both).  0: one 256-thread workgroup, 1: one 1024-thread workgroup, 2: FMG co-resident
// workgroups in one launch, 3: one launch per iteration.
struct FgrOptRule { int single_max, multi_min, multi_max; unsigned long long multi_timeout; };
static const FgrOptRule &fgr_opt_rule() {
    static const FgrOptRule r = {
        getenv("PCR_FGR_SINGLE_MAX") ? atoi(getenv("PCR_FGR_SINGLE_MAX")) : 11000,   // one CU needs 0.8 us of float64 work per 1000 correspondences and iteration (+1.4 us solve); a launch per iteration costs 11 us
        getenv("PCR_FGR_MULTI_MIN") ? atoi(getenv("PCR_FGR_MULTI_MIN")) : 6000,      // from here on FMG co-resident workgroups in one launch
        getenv("PCR_FGR_MULTI_MAX") ? atoi(getenv("PCR_FGR_MULTI_MAX")) : 400000,
        // ticks of the 100 MHz wall clock a workgroup waits at the barrier (tests set 0 to force the fallback)
        getenv("PCR_FGR_MULTI_TIMEOUT") ? strtoull(getenv("PCR_FGR_MULTI_TIMEOUT"), nullptr, 10) : 5000000ull};
    return r;
}
static int fgr_opt_variant(int64_t ncorr) {
    const FgrOptRule &r = fgr_opt_rule();
    if (ncorr >= r.multi_min && ncorr <= r.multi_max) return 2;
    if (ncorr <= r.single_max) return ncorr <= FGR_SINGLE_SMALL ? 0 : 1;
    return 3;
}

// ======================================================================== driver
static int normalise(pcr_context *ctx, const float *xyz, int64_t n, double *out64, double *mean_host, double *max_host) {
    const int nb = (int)((n + FB - 1) / FB < 256 ? (n + FB - 1) / FB : 256);
    double *part = arena<double>(ctx, (size_t)nb * 3 + 3 + nb);
    if (!part) return PCR_ENOMEM;
    double *mean3 = part + (size_t)nb * 3, *pmax = mean3 + 3;
    PCR_LAUNCH(ctx, k_sum3, dim3(nb), dim3(FB), 0, ctx->stream, xyz, (int)n, part);
    PCR_LAUNCH(ctx, k_sum3_final, dim3(1), dim3(64), 0, ctx->stream, part, nb, (int)n, mean3);
    PCR_LAUNCH(ctx, k_center, dim3(nb), dim3(FB), 0, ctx->stream, xyz, (int)n, mean3, out64, pmax);
    std::vector<double> h((size_t)3 + nb);
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), mean3, sizeof(double) * (3 + (size_t)nb), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int d = 0; d < 3; d++) mean_host[d] = h[d];
    double mx = 0; for (int k = 0; k < nb; k++) mx = h[3 + k] > mx ? h[3 + k] : mx;
    *max_host = mx;
    return PCR_OK;
}


static size_t fgr_scratch_bytes(int64_t ns, int64_t nt, const pcr_fgr_option *opt) {
    const int64_t nmax = ns > nt ? ns : nt;
    const long long trial_cap = opt->tuple_test ? 100ll * nmax : 0;
    return (size_t)(ns + nt) * (24 + 8 + 16) + (size_t)nmax * 32 + (size_t)trial_cap * 5 + (size_t)nmax * 64 + (size_t)nmax * (FK * 16 + 32 * 12 + 64) + (64u << 20)
           + pcr_feature_nn_scratch_bytes(ns, nt);
}

// AdvancedMatching + OptimizePairwiseRegistration: the source -> target pose from the two clouds and their features (caller-order
// device arrays).  Scratch comes from the arena above the current mark (the caller has reserved fgr_scratch_bytes).
static int fgr_pose(pcr_context *ctx, const float *src_xyz, const float *src_feat, int64_t ns, const float *tgt_xyz, const float *tgt_feat, int64_t nt,
                    const pcr_fgr_option *opt, double *Tsrc2tgt) {
    { const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; memcpy(Tsrc2tgt, I, sizeof I); }
    if (ns > 0 && nt > 0) {
        const int64_t nmax = ns > nt ? ns : nt;
        const long long trial_cap = opt->tuple_test ? 100ll * nmax : 0;
        ArenaMark mark(ctx);
        // ---- NormalizePointCloud
        double *P[2] = {arena<double>(ctx, (size_t)ns * 3), arena<double>(ctx, (size_t)nt * 3)};
        if (!P[0] || !P[1]) return PCR_ENOMEM;
        double mean[2][3], mx[2];
        PCR_TRY(normalise(ctx, src_xyz, ns, P[0], mean[0], &mx[0]));
        PCR_TRY(normalise(ctx, tgt_xyz, nt, P[1], mean[1], &mx[1]));
        const double scale = mx[0] > mx[1] ? mx[0] : mx[1];
        const double scale_global = opt->use_absolute_scale ? 1.0 : scale, scale_start = opt->use_absolute_scale ? scale : 1.0;
        if (scale_global != 1.0) {
            PCR_LAUNCH(ctx, k_scale, dim3((unsigned)((ns * 3 + FB - 1) / FB)), dim3(FB), 0, ctx->stream, P[0], (long long)ns * 3, scale_global);
            PCR_LAUNCH(ctx, k_scale, dim3((unsigned)((nt * 3 + FB - 1) / FB)), dim3(FB), 0, ctx->stream, P[1], (long long)nt * 3, scale_global);
        }
        // ---- AdvancedMatching: mutual nearest neighbours in feature space; i = the larger cloud
        const int swapped = nt > ns ? 1 : 0;
        const float *fi = swapped ? tgt_feat : src_feat, *fj = swapped ? src_feat : tgt_feat;
        const int nPti = (int)(swapped ? nt : ns), nPtj = (int)(swapped ? ns : nt);
        int32_t *j_to_i = arena<int32_t>(ctx, nPtj), *i_to_j = arena<int32_t>(ctx, nPti);
        uint8_t *cflags = arena<uint8_t>(ctx, nPti);
        int *cpos = arena<int>(ctx, nPti), *ncross_dev = arena<int>(ctx, 1);
        int32_t *cross = arena<int32_t>(ctx, (size_t)nPti * 2);
        if (!j_to_i || !i_to_j || !cflags || !cpos || !ncross_dev || !cross) return PCR_ENOMEM;
        hipEvent_t pe[2] = {nullptr, nullptr};
        if (ctx->profiling) { PCR_HIP_CHECK(ctx, hipEventCreate(&pe[0])); PCR_HIP_CHECK(ctx, hipEventCreate(&pe[1])); PCR_HIP_CHECK(ctx, hipEventRecord(pe[0], ctx->stream)); }
        // both directions: f16-split screen on the matrix cores + exact float64 re-check (pcr_featnn.hip); PCR_FEATURE_NN=f64 keeps
        // the all-pairs float64 MFMA path (also taken for feature values outside the f16 range), PCR_FEATURE_NN_BRUTE the float32 one
        static const bool nn_f64 = getenv("PCR_FEATURE_NN") && !strcmp(getenv("PCR_FEATURE_NN"), "f64");
        int nn_rc = PCR_ECAPACITY;
        if (!nn_f64 && !getenv("PCR_FEATURE_NN_BRUTE") && nPti >= 64 && nPtj >= 64) nn_rc = pcr_feature_nn_mutual(ctx, fi, nPti, fj, nPtj, j_to_i, i_to_j, -1, 1);      // (only the cross check below reads the two lists)
        if (nn_rc == PCR_ECAPACITY) {
            PCR_TRY(feature_nn(ctx, fi, nPti, fj, nPtj, j_to_i));
            PCR_TRY(feature_nn(ctx, fj, nPtj, fi, nPti, i_to_j));
        } else if (nn_rc != PCR_OK) return nn_rc;
        if (ctx->profiling) {             // bench instrumentation: HIP-event time over the two matching passes (pcr_hip.h, out16[8..10])
            PCR_HIP_CHECK(ctx, hipEventRecord(pe[1], ctx->stream));
            PCR_HIP_CHECK(ctx, hipEventSynchronize(pe[1]));
            float ms = 0; PCR_HIP_CHECK(ctx, hipEventElapsedTime(&ms, pe[0], pe[1]));
            ctx->prof[8] += ms; ctx->prof[9] += 2.0 * (2.0 * 33.0 * (double)nPti * (double)nPtj); ctx->prof[10] += 2.0;
            (void)hipEventDestroy(pe[0]); (void)hipEventDestroy(pe[1]);
        }
        PCR_LAUNCH(ctx, k_cross_flags, dim3((nPti + FB - 1) / FB), dim3(FB), 0, ctx->stream, i_to_j, j_to_i, nPti, cflags);
        PCR_TRY(pcr_dev_flag_scan(ctx, cflags, nullptr, nPti, cpos, ncross_dev));
        PCR_LAUNCH(ctx, k_cross_emit, dim3((nPti + FB - 1) / FB), dim3(FB), 0, ctx->stream, i_to_j, cflags, cpos, nPti, cross);
        int64_t ncross = 0;
        PCR_TRY(pcr_read_count(ctx, ncross_dev, &ncross));
        // ---- tuple test
        int32_t *corr = nullptr; int64_t ncorr = 0;
        if (opt->tuple_test && ncross > 0 && opt->maximum_tuple_count > 0) {
            const long long trials = 100ll * ncross;
            uint8_t *tflags = arena<uint8_t>(ctx, trials);
            int *tpos = arena<int>(ctx, trials), *ntup_dev = arena<int>(ctx, 1);
            corr = arena<int32_t>(ctx, (size_t)opt->maximum_tuple_count * 6);
            if (!tflags || !tpos || !ntup_dev || !corr) return PCR_ENOMEM;
            TupleArgs ta;
            ta.pi = swapped ? P[1] : P[0]; ta.pj = swapped ? P[0] : P[1]; ta.cross = cross; ta.ncross = (int)ncross;
            ta.seed = opt->seed; ta.tuple_scale = opt->tuple_scale; ta.trials = trials;
            const unsigned gb = (unsigned)((trials + FB - 1) / FB);
            PCR_LAUNCH(ctx, k_tuple_flags, dim3(gb), dim3(FB), 0, ctx->stream, ta, tflags);
            PCR_TRY(pcr_dev_flag_scan(ctx, tflags, nullptr, (int)trials, tpos, ntup_dev));
            PCR_LAUNCH(ctx, k_tuple_emit, dim3(gb), dim3(FB), 0, ctx->stream, ta, tflags, tpos, opt->maximum_tuple_count, swapped, corr);
            int64_t nacc = 0;
            PCR_TRY(pcr_read_count(ctx, ntup_dev, &nacc));
            if (nacc > opt->maximum_tuple_count) nacc = opt->maximum_tuple_count;
            ncorr = nacc * 3;
        } else if (ncross > 0) {
            // tuple_test = false: every cross-checked pair (Open3D newer versions); (cloud0, cloud1) order
            corr = arena<int32_t>(ctx, (size_t)ncross * 2);
            if (!corr) return PCR_ENOMEM;
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(corr, cross, sizeof(int32_t) * 2 * (size_t)ncross, hipMemcpyDeviceToDevice, ctx->stream));
            ncorr = ncross;
            if (swapped) { ctx->err = "tuple_test=false with a larger target is not supported yet"; return PCR_EINVAL; }
        }
        // ---- OptimizePairwiseRegistration (moves cloud 1 = target onto cloud 0 = source)
        double trans[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (ncorr >= 10) {
            int nb = (int)((ncorr + FB - 1) / FB);
            if (nb > 256) nb = 256;
            FgrState *st = arena<FgrState>(ctx, 1);
            double *partials = arena<double>(ctx, (size_t)nb * FNVP);
            const int stride = (int)((ncorr + 63) / 64 * 64);
            double *pq = arena<double>(ctx, (size_t)stride * 6);
            if (!st || !partials || !pq) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fgr_gather_pairs, dim3((unsigned)((ncorr + FB - 1) / FB)), dim3(FB), 0, ctx->stream, P[0], P[1], corr, (int)ncorr, stride, pq);
            FgrOptArgs oa;
            oa.pq = pq; oa.stride = stride; oa.ncorr = (int)ncorr; oa.st = st; oa.partials = partials;
            oa.decrease_mu = opt->decrease_mu; oa.max_corr_dist = opt->maximum_correspondence_distance; oa.division_factor = opt->division_factor;
            PCR_LAUNCH(ctx, k_fgr_init, dim3(1), dim3(64), 0, ctx->stream, st, scale_start);
            if (getenv("PCR_DEBUG_FGR")) fprintf(stderr, "fgr: ncross %lld ncorr %lld iterations %d\n", (long long)ncross, (long long)ncorr, (int)opt->iteration_number);
            double *rows = arena<double>(ctx, (size_t)2 * FMG_MAX * FNVP);
            if (!rows) return PCR_ENOMEM;
            const int variant = fgr_opt_variant(ncorr);
            const bool multi = variant == 2;
            FgrState h;
            auto per_iteration = [&]() { for (int it = 0; it < opt->iteration_number; it++) PCR_LAUNCH(ctx, k_fgr_iter, dim3(nb), dim3(FB), 0, ctx->stream, oa); };
            static const int wide_env = getenv("PCR_FGR_MULTI_WGS") ? atoi(getenv("PCR_FGR_MULTI_WGS")) : 0;      // (diagnostics: 8 .. 32)
            const int multi_wgs = wide_env >= 1 && wide_env <= FMG_MAX ? wide_env : (ncorr >= FGR_MULTI_WIDE ? FMG_MAX : FMG);
            if (multi) PCR_LAUNCH(ctx, k_fgr_opt_multi, dim3(multi_wgs), dim3(FMB), 0, ctx->stream, oa, (int)opt->iteration_number, rows, fgr_opt_rule().multi_timeout);
            else if (variant == 0) PCR_LAUNCH(ctx, k_fgr_opt_single<256>, dim3(1), dim3(256), 0, ctx->stream, oa, (int)opt->iteration_number);
            else if (variant == 1) PCR_LAUNCH(ctx, k_fgr_opt_single<FSB>, dim3(1), dim3(FSB), 0, ctx->stream, oa, (int)opt->iteration_number);
            else per_iteration();
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (multi && h.failed) {                 // the workgroups were not co-resident in time: same result, one launch per iteration
                if (getenv("PCR_DEBUG_FGR")) fprintf(stderr, "fgr: multi-workgroup optimiser gave up waiting, falling back to one launch per iteration\n");
                PCR_LAUNCH(ctx, k_fgr_init, dim3(1), dim3(64), 0, ctx->stream, st, scale_start);
                per_iteration();
                PCR_HIP_CHECK(ctx, hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
                PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            }
            memcpy(trans, h.trans, sizeof trans);
            if (getenv("PCR_DEBUG_FGR") && h.t_dbg[0]) fprintf(stderr, "fgr single-workgroup optimiser, us per iteration: accumulate %.2f reduce %.2f solve+update %.2f\n",
                                                            h.t_dbg[0] * 0.01 / h.itr, h.t_dbg[1] * 0.01 / h.itr, h.t_dbg[2] * 0.01 / h.itr);
        }
        // ---- GetTransformationOriginalScale, then invert: source -> target
        double To[16] = {0};
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) To[r * 4 + c] = trans[r * 4 + c];
            To[r * 4 + 3] = -(trans[r * 4 + 0] * mean[1][0] + trans[r * 4 + 1] * mean[1][1] + trans[r * 4 + 2] * mean[1][2]) + trans[r * 4 + 3] * scale_global + mean[0][r];
        }
        To[15] = 1;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) Tsrc2tgt[r * 4 + c] = To[c * 4 + r];
            Tsrc2tgt[r * 4 + 3] = -(To[0 * 4 + r] * To[3] + To[1 * 4 + r] * To[7] + To[2 * 4 + r] * To[11]);
        }
        for (int k = 0; k < 16; k++) if (!std::isfinite(Tsrc2tgt[k])) { ctx->err = "non-finite FGR pose"; return PCR_ENUMERIC; }
    }
    return PCR_OK;
}

static int fgr_check_args(pcr_context *ctx, const pcr_fgr_option *opt, int64_t ns, int64_t nt) {
    if (!opt || ns < 0 || nt < 0) return PCR_EINVAL;
    if (ns > 0x7fffffff / 8 || nt > 0x7fffffff / 8) { ctx->err = "cloud too large"; return PCR_EINVAL; }
    if (!(opt->maximum_correspondence_distance > 0.0)) { ctx->err = "maximum_correspondence_distance <= 0"; return PCR_EINVAL; }
    return PCR_OK;
}

extern "C" int pcr_registration_fgr(pcr_context *ctx, const float *src_xyz, const float *src_feat, int64_t ns, const float *tgt_xyz,
                                    const float *tgt_feat, int64_t nt, const pcr_fgr_option *opt, pcr_result *result, int32_t *correspondences) {
    return pcr_api_call(ctx, [&]() -> int {
    if (!result) return PCR_EINVAL;
    PCR_TRY(fgr_check_args(ctx, opt, ns, nt));
    pcr_fgr_option opt_local = *opt;
    if (opt_local.maximum_tuple_count < 0) opt_local.maximum_tuple_count = (int32_t)((double)((ns + nt) / 2) * 0.2);     // the per-pair rule (pcr_hip.h)
    const pcr_fgr_option *opt = &opt_local;
    if ((ns > 0 && (!src_xyz || !src_feat)) || (nt > 0 && (!tgt_xyz || !tgt_feat))) return PCR_EINVAL;
    double Tsrc2tgt[16];
    PCR_TRY(pcr_arena_reserve(ctx, fgr_scratch_bytes(ns, nt, opt)));
    PCR_TRY(fgr_pose(ctx, src_xyz, src_feat, ns, tgt_xyz, tgt_feat, nt, opt, Tsrc2tgt));
    // ---- EvaluateRegistration(source, target, maximum_correspondence_distance, T)
    return pcr_evaluate_registration_impl(ctx, src_xyz, ns, tgt_xyz, nt, opt->maximum_correspondence_distance, Tsrc2tgt, result, correspondences);
    });
}

// ---- registro_FGR in one call (ALL_FUNCTIONS.py:178-203): every cloud is sorted and indexed once; its tree serves the hybrid
// normals, the FPFH neighbour lists and (target) the final evaluate_registration
static int fgr_tail(pcr_context *ctx, DevCloud *c, uint32_t **perm, float **feat, const float *src_xyz, const float *tgt_xyz, int64_t ns, int64_t nt,
                    const pcr_fgr_params *p, pcr_result *result, int32_t *correspondences);
int pcr_registro_fgr_impl(pcr_context *ctx, const float *src_xyz, const float *src_prior, int64_t ns, const float *tgt_xyz, const float *tgt_prior, int64_t nt,
                          const pcr_fgr_params *p_in, float *src_normals_out, float *tgt_normals_out, pcr_result *result, int32_t *correspondences) {
    if (!p_in || !result) return PCR_EINVAL;
    pcr_fgr_params p_local = *p_in;
    // maximum_tuple_count < 0: the reference's rule for THIS pair, int(0.2 * n_pontos) with n_pontos = int((len(source) + len(target)) / 2)
    // (ALL_FUNCTIONS.py:179,196) -- so that one plan serves pairs of different sizes (every NCLT scan has its own point count)
    if (p_local.option.maximum_tuple_count < 0) p_local.option.maximum_tuple_count = (int32_t)((double)((ns + nt) / 2) * 0.2);
    const pcr_fgr_params *p = &p_local;
    // the group forms of a lockstep plan concern its GICP stage: registro_FGR is the same arithmetic in every plan
    struct FormsGuard { pcr_context *c; bool was; ~FormsGuard() { c->group_forms = was; } } forms_guard{ctx, ctx->group_forms};
    ctx->group_forms = false;
    PCR_TRY(fgr_check_args(ctx, &p->option, ns, nt));
    if ((ns > 0 && !src_xyz) || (nt > 0 && !tgt_xyz)) return PCR_EINVAL;
    if (p->normal_max_nn < 1 || !(p->normal_radius > 0.0)) { ctx->err = "estimate_normals: radius <= 0 or max_nn < 1"; return PCR_EINVAL; }
    const int fk = p->feature_max_nn > 0 && p->feature_max_nn <= 200 ? p->feature_max_nn : 1;
    const size_t per_cloud = fpfh_scratch_bytes(ns > nt ? ns : nt, fk);
    const size_t fgr_b = fgr_scratch_bytes(ns, nt, &p->option);
    // the target cloud's chain (import, normals, FPFH) runs on the side lane in a block of its own while the source's runs on the
    // context's stream: their k-NN kernels fill the chip either way, the sorts, tree builds and tails overlap
    const size_t lane_block = nt > 0 ? pcr_scratch_bytes_for(nt) + (size_t)(nt + 2) * (33 * 4 + 16 + 8) + fpfh_scratch_bytes(nt, fk) + (1u << 20) : 0;
    const size_t own_fpfh = fpfh_scratch_bytes(ns, fk);
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(ns) + (size_t)(ns + 2) * (33 * 4 + 16 + 8) + lane_block + (own_fpfh > fgr_b ? own_fpfh : fgr_b) + (1u << 20)));
    DevCloud c[2]; uint32_t *perm[2] = {nullptr, nullptr}; float *feat[2] = {nullptr, nullptr};
    const float *xyz[2] = {src_xyz, tgt_xyz}, *prior[2] = {src_prior, tgt_prior}; float *nout[2] = {src_normals_out, tgt_normals_out};
    const int64_t n[2] = {ns, nt};
    auto cloud_chain = [&](int k) -> int {
        PCR_TRY(pcr_import_cloud(ctx, xyz[k], prior[k], n[k], &c[k], &perm[k], true));
        float4 *nrm_new = arena<float4>(ctx, n[k] > 0 ? n[k] : 1);
        feat[k] = arena<float>(ctx, (size_t)(n[k] > 0 ? n[k] : 1) * 33);
        if (!nrm_new || !feat[k]) return PCR_ENOMEM;
        if (n[k] == 0) return PCR_OK;
        PCR_TRY(pcr_dev_normals(ctx, &c[k], PCR_SEARCH_HYBRID, p->normal_max_nn, p->normal_radius, prior[k] ? c[k].nrm : nullptr, nrm_new, nullptr));
        c[k].nrm = nrm_new;
        if (nout[k]) PCR_TRY(pcr_dev_scatter_rows_f4_to_f3(ctx, nrm_new, perm[k], c[k].n, c[k].cap, nout[k]));
        return fpfh_of_cloud(ctx, c[k], perm[k], n[k], PCR_SEARCH_HYBRID, p->feature_max_nn, p->feature_radius, feat[k]);
    };
    static const bool two_lanes = !(getenv("PCR_FGR_LANES") && atoi(getenv("PCR_FGR_LANES")) < 2);
    char *block = (two_lanes && nt > 0) ? (char *)pcr_arena_alloc(ctx, lane_block) : nullptr;
    if (block) {
        PCR_TRY(pcr_ensure_lanes(ctx, 1));
        struct LaneGuard { hipStream_t a; ~LaneGuard() { (void)hipStreamSynchronize(a); } } guard{ctx->side_stream};   // no lane work over a recycled arena
        PCR_HIP_CHECK(ctx, hipEventRecord(ctx->side_ev[0], ctx->stream));       // the inputs are ready once the caller's stream gets here
        PCR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[0], 0));
        {
            SideLane lane(ctx, block, lane_block, ctx->side_stream);
            PCR_TRY(cloud_chain(1));
            PCR_HIP_CHECK(ctx, hipEventRecord(ctx->side_ev[1], ctx->stream));
        }
        PCR_TRY(cloud_chain(0));
        PCR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_ev[1], 0));
        return fgr_tail(ctx, c, perm, feat, src_xyz, tgt_xyz, ns, nt, p, result, correspondences);
    }
    for (int k = 0; k < 2; k++) PCR_TRY(cloud_chain(k));
    return fgr_tail(ctx, c, perm, feat, src_xyz, tgt_xyz, ns, nt, p, result, correspondences);
}
static int fgr_tail(pcr_context *ctx, DevCloud *c, uint32_t **perm, float **feat, const float *src_xyz, const float *tgt_xyz, int64_t ns, int64_t nt,
                    const pcr_fgr_params *p, pcr_result *result, int32_t *correspondences) {
    double T[16];
    PCR_TRY(fgr_pose(ctx, src_xyz, feat[0], ns, tgt_xyz, feat[1], nt, &p->option, T));
    int32_t *match = arena<int32_t>(ctx, ns > 0 ? ns : 1);
    if (!match) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_evaluate(ctx, &c[0], &c[1], p->option.maximum_correspondence_distance, T, result, match, nullptr));
    for (int k = 0; k < 16; k++) result->transformation[k] = T[k];
    if (correspondences) {
        int64_t nc = 0;
        PCR_TRY(pcr_dev_compact_matches(ctx, match, c[0].n, c[0].cap, perm[0], perm[1], correspondences, &nc));
    }
    return PCR_OK;
}

extern "C" int pcr_registro_fgr(pcr_context *ctx, const float *src_xyz, const float *src_prior, int64_t ns, const float *tgt_xyz, const float *tgt_prior, int64_t nt,
                                const pcr_fgr_params *p, float *src_normals_out, float *tgt_normals_out, pcr_result *result, int32_t *correspondences) {
    return pcr_api_call(ctx, [&]() -> int { return pcr_registro_fgr_impl(ctx, src_xyz, src_prior, ns, tgt_xyz, tgt_prior, nt, p, src_normals_out, tgt_normals_out, result, correspondences); });
}

// ============================================================================================ registro_FGR of a GROUP of pairs in lockstep
// The per-pair loop of 1_FGR_pairwise_registration_in_NCLT_dataset.py:134-147 (ALL_FUNCTIONS.py:349-357 in full_registration) for G pairs
// through the SAME launches: registro_FGR on an NCLT-size pair is a chain of ~125 small dependent kernels and 8 host waits, and the device
// retires ~90 k such kernels per second however many pairs are in flight -- so G pairs share every launch (blockIdx.y / .z = cloud, pair
// or (pair, direction)) and every host wait (6 per GROUP).  Per pair the arithmetic is that of pcr_registro_fgr_impl -- same kernels
// bodies, same grids per problem where a grid decides a summation order, same optimiser variant by correspondence count -- so poses,
// normals and correspondence sets are the same bits (tests/test_gpu_fgr.py).  Taken for pairs whose mutual feature search runs
// without tile pruning (under 5e9 row pairs: NCLT-size clouds); returns 1 = declined (the caller runs the pairs one by one).
struct Sum3Desc { const float *xyz; int n, nb; double *part, *mean3, *out, *pmax; };
__global__ void __launch_bounds__(FB) k_sum3_g(const Sum3Desc *d) {
    const Sum3Desc a = d[blockIdx.y];
    if ((int)blockIdx.x >= a.nb) return;
    double s[3] = {0, 0, 0};
    for (int i = blockIdx.x * FB + threadIdx.x; i < a.n; i += a.nb * FB) { s[0] += a.xyz[i * 3]; s[1] += a.xyz[i * 3 + 1]; s[2] += a.xyz[i * 3 + 2]; }
    __shared__ double sh[FB / PCR_WAVE][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) { const double v = pcr_wave_sum(s[k]); if (lane == 0) sh[w][k] = v; }
    __syncthreads();
    if (threadIdx.x < 3) { double v = 0; for (int k = 0; k < FB / PCR_WAVE; k++) v += sh[k][threadIdx.x]; a.part[blockIdx.x * 3 + threadIdx.x] = v; }
}
__global__ void k_sum3_final_g(const Sum3Desc *d) {
    const Sum3Desc a = d[blockIdx.x];
    if (threadIdx.x < 3) { double v = 0; for (int k = 0; k < a.nb; k++) v += a.part[k * 3 + threadIdx.x]; a.mean3[threadIdx.x] = a.n > 0 ? v / (double)a.n : 0.0; }
}
__global__ void __launch_bounds__(FB) k_center_g(const Sum3Desc *d) {
    const Sum3Desc a = d[blockIdx.y];
    if ((int)blockIdx.x >= a.nb) return;
    double mx = 0;
    for (int i = blockIdx.x * FB + threadIdx.x; i < a.n; i += a.nb * FB) {
        const double x = (double)a.xyz[i * 3] - a.mean3[0], y = (double)a.xyz[i * 3 + 1] - a.mean3[1], z = (double)a.xyz[i * 3 + 2] - a.mean3[2];
        a.out[(size_t)i * 3] = x; a.out[(size_t)i * 3 + 1] = y; a.out[(size_t)i * 3 + 2] = z;
        mx = fmax(mx, sqrt(x * x + y * y + z * z));
    }
    __shared__ double sh[FB];
    sh[threadIdx.x] = mx;
    __syncthreads();
    for (int o = FB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) a.pmax[blockIdx.x] = sh[0];
}
struct ScaleDesc { double *p; long long n3; double inv; };
__global__ void __launch_bounds__(FB) k_scale_g(const ScaleDesc *d) {
    const ScaleDesc a = d[blockIdx.y];
    const long long i = (long long)blockIdx.x * FB + threadIdx.x;
    if (i < a.n3) a.p[i] = a.p[i] / a.inv;
}
struct CrossDesc { const int32_t *i_to_j, *j_to_i; int n_i; uint8_t *flags; const int *pos; int32_t *cross; };
__global__ void __launch_bounds__(FB) k_cross_flags_g(const CrossDesc *d) {
    const CrossDesc a = d[blockIdx.y];
    const int i = blockIdx.x * FB + threadIdx.x;
    if (i >= a.n_i) return;
    const int j = a.i_to_j[i];
    a.flags[i] = (j >= 0 && a.j_to_i[j] == i) ? 1 : 0;
}
__global__ void __launch_bounds__(FB) k_cross_emit_g(const CrossDesc *d) {
    const CrossDesc a = d[blockIdx.y];
    const int i = blockIdx.x * FB + threadIdx.x;
    if (i >= a.n_i || !a.flags[i]) return;
    a.cross[2 * (size_t)a.pos[i]] = i; a.cross[2 * (size_t)a.pos[i] + 1] = a.i_to_j[i];
}
struct TupleDesc { TupleArgs t; uint8_t *flags; const int *pos; int max_tuples, swapped; int32_t *corr; };
__global__ void __launch_bounds__(FB) k_tuple_flags_g(const TupleDesc *d) {
    const TupleDesc &a = d[blockIdx.y];
    const long long t = (long long)blockIdx.x * FB + threadIdx.x;
    if (t >= a.t.trials) return;
    int r[3];
    a.flags[t] = tuple_ok(a.t, t, r) ? 1 : 0;
}
__global__ void __launch_bounds__(FB) k_tuple_emit_g(const TupleDesc *d) {
    const TupleDesc &a = d[blockIdx.y];
    const long long t = (long long)blockIdx.x * FB + threadIdx.x;
    if (t >= a.t.trials || !a.flags[t]) return;
    const int rank = a.pos[t];
    if (rank >= a.max_tuples) return;
    int r[3];
    (void)tuple_ok(a.t, t, r);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int ii = a.t.cross[2 * r[k]], jj = a.t.cross[2 * r[k] + 1];
        a.corr[((size_t)rank * 3 + k) * 2] = a.swapped ? jj : ii;
        a.corr[((size_t)rank * 3 + k) * 2 + 1] = a.swapped ? ii : jj;
    }
}
struct OptPrepDesc { const double *p0, *q0; const int32_t *corr; int ncorr, stride; double *pq; FgrState *st; double par; };
__global__ void __launch_bounds__(FB) k_fgr_prepare_g(const OptPrepDesc *d) {       // k_fgr_gather_pairs + k_fgr_init of every pair
    const OptPrepDesc a = d[blockIdx.y];
    const int c = blockIdx.x * FB + threadIdx.x;
    if (c == 0) {
        for (int k = 0; k < 16; k++) a.st->trans[k] = (k % 5 == 0) ? 1.0 : 0.0;
        a.st->par = a.par; a.st->ticket = 0; a.st->itr = 0;
        for (int k = 0; k < 4; k++) a.st->t_dbg[k] = 0;
        a.st->arrive = 0; a.st->failed = 0;
    }
    if (c >= a.ncorr) return;
    const double *p = a.p0 + (size_t)a.corr[2 * c] * 3, *q = a.q0 + (size_t)a.corr[2 * c + 1] * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) { a.pq[(size_t)k * a.stride + c] = p[k]; a.pq[(size_t)(3 + k) * a.stride + c] = q[k]; }
}

size_t pcr_registro_fgr_group_bytes(const pcr_fgr_group_pair *q, int G) {
    size_t total = 64u << 20;
    for (int g = 0; g < G; g++) {
        const int fk = q[g].p.feature_max_nn > 0 && q[g].p.feature_max_nn <= 200 ? q[g].p.feature_max_nn : 1;
        for (int64_t n : {q[g].ns, q[g].nt}) total += pcr_scratch_bytes_for(n) + (size_t)(n + 2) * (33 * 4 + 16 + 8 + 48) + fpfh_scratch_bytes(n, fk);
        total += fgr_scratch_bytes(q[g].ns, q[g].nt, &q[g].p.option);
    }
    return total;
}
int pcr_registro_fgr_group(pcr_context *ctx, pcr_fgr_group_pair *q, int G) {
    if (G < 1 || G > 64) return 1;
    for (int g = 0; g < G; g++) {
        pcr_fgr_group_pair &a = q[g];
        if (!a.result) return PCR_EINVAL;
        if (a.p.option.maximum_tuple_count < 0) a.p.option.maximum_tuple_count = (int32_t)((double)((a.ns + a.nt) / 2) * 0.2);       // the per-pair rule (pcr_hip.h)
        if (fgr_check_args(ctx, &a.p.option, a.ns, a.nt) != PCR_OK) return 1;               // an argument error lands on its pair in the one-pair path
        if (!pcr_fgr_group_takes(a.ns, a.nt) || !a.src_xyz || !a.tgt_xyz) return 1;       // tiny clouds, or the pruned feature search: one pair at a time
        if (a.p.normal_max_nn < 1 || a.p.normal_max_nn > 32 || !(a.p.normal_radius > 0.0) || a.p.feature_max_nn < 1 || a.p.feature_max_nn > 200 || !(a.p.feature_radius > 0.0)) return 1;
        if (!a.p.option.tuple_test || a.p.option.maximum_tuple_count < 1) return 1;
        const pcr_fgr_params &p0 = q[0].p;
        if (a.p.normal_max_nn != p0.normal_max_nn || a.p.normal_radius != p0.normal_radius || a.p.feature_max_nn != p0.feature_max_nn || a.p.feature_radius != p0.feature_radius ||
            a.p.option.maximum_correspondence_distance != p0.option.maximum_correspondence_distance || a.p.option.iteration_number != p0.option.iteration_number) return 1;
        a.status = PCR_OK;
    }
    const pcr_fgr_params &p0 = q[0].p;
    struct Flags { pcr_context *c; bool forms, octet; ~Flags() { c->group_forms = forms; c->octet_only = octet; } } flags_guard{ctx, ctx->group_forms, ctx->octet_only};
    ctx->group_forms = false; ctx->octet_only = true;                  // the one-cloud searches of registro_FGR run the octet kernel
    PCR_TRY(pcr_arena_reserve(ctx, pcr_registro_fgr_group_bytes(q, G)));
    const int C = 2 * G;                                                  // cloud c = 2 g + which (0 source, 1 target)
    std::vector<DevCloud> c((size_t)C); std::vector<uint32_t *> perm((size_t)C, nullptr); std::vector<float *> feat((size_t)C, nullptr);
    std::vector<const float *> xyz((size_t)C), prior((size_t)C); std::vector<int64_t> n((size_t)C); std::vector<float *> nout((size_t)C);
    bool any_prior = false;
    for (int g = 0; g < G; g++) {
        xyz[2 * g] = q[g].src_xyz; xyz[2 * g + 1] = q[g].tgt_xyz; prior[2 * g] = q[g].src_prior; prior[2 * g + 1] = q[g].tgt_prior;
        n[2 * g] = q[g].ns; n[2 * g + 1] = q[g].nt; nout[2 * g] = q[g].src_normals_out; nout[2 * g + 1] = q[g].tgt_normals_out;
        any_prior = any_prior || q[g].src_prior || q[g].tgt_prior;
    }
    // ---- import (bounds read-back: wait 1), hybrid normals, FPFH neighbour lists, SPFH, FPFH: 2 G clouds per launch
    for (int k = 0; k < C; k++) PCR_TRY(pcr_alloc_cloud(ctx, &c[k], (int)n[k], true, true));
    PCR_TRY(pcr_import_clouds_batch(ctx, C, xyz.data(), any_prior ? prior.data() : nullptr, n.data(), c.data(), perm.data()));
    {
        std::vector<DevCloud *> cp((size_t)C); std::vector<const float4 *> pr((size_t)C); std::vector<float4 *> nn((size_t)C);
        for (int k = 0; k < C; k++) {
            cp[k] = &c[k]; pr[k] = prior[k] ? c[k].nrm : nullptr;
            nn[k] = arena<float4>(ctx, n[k]); feat[k] = arena<float>(ctx, (size_t)n[k] * 33);
            if (!nn[k] || !feat[k]) return PCR_ENOMEM;
        }
        PCR_TRY(pcr_dev_normals_batch(ctx, cp.data(), C, PCR_SEARCH_HYBRID, p0.normal_max_nn, p0.normal_radius, pr.data(), nn.data(), nullptr));
        for (int k = 0; k < C; k++) {
            c[k].nrm = nn[k];
            if (nout[k]) PCR_TRY(pcr_dev_scatter_rows_f4_to_f3(ctx, nn[k], perm[k], c[k].n, c[k].cap, nout[k]));
        }
    }
    std::vector<FpfhArgs> fa((size_t)C);
    {
        std::vector<const DevCloud *> cp((size_t)C); std::vector<int32_t *> nbr((size_t)C), ncnt((size_t)C);
        int64_t nmax = 0;
        const int spfh_mode = pcr_options().spfh_float64.load(std::memory_order_relaxed);        // (as fpfh_of_cloud; read ONCE: one form for the launch, by the largest cloud)
        int64_t n_all_max = 0; for (int kk = 0; kk < C; kk++) n_all_max = n[kk] > n_all_max ? n[kk] : n_all_max;
        for (int k = 0; k < C; k++) {
            cp[k] = &c[k];
            nbr[k] = arena<int32_t>(ctx, (size_t)n[k] * p0.feature_max_nn); ncnt[k] = arena<int32_t>(ctx, n[k]);
            SpfhRow *spfh = arena<SpfhRow>(ctx, (size_t)n[k]);
            if (!nbr[k] || !ncnt[k] || !spfh) return PCR_ENOMEM;
            FpfhArgs &a = fa[k];
            a.pts = c[k].pts; a.nrm = c[k].nrm; a.n_ptr = c[k].n; a.nbr = nbr[k]; a.k = p0.feature_max_nn; a.r2 = p0.feature_radius * p0.feature_radius; a.cnt = ncnt[k];
            a.spfh = spfh; a.perm = perm[k]; a.feat = feat[k]; a.float64_only = (spfh_mode == 3 || spfh_mode == 4) ? 0 : (spfh_mode == 0 ? (n_all_max >= SPFH_SPLIT_MIN_POINTS ? 0 : 1) : 1); a.verify = nullptr;
            a.slowq = nullptr; a.slow_count = nullptr; a.slow_cap = 0; a.only_if_over = nullptr; a.over_cap = 0;
            if (!a.float64_only) {
                a.slow_cap = spfh_mode == 3 ? 16 : (int)std::min<int64_t>(8 * n[k], 1 << 28);
                a.slowq = arena<uint2>(ctx, (size_t)a.slow_cap); a.slow_count = arena<int>(ctx, 1);
                if (!a.slowq || !a.slow_count) return PCR_ENOMEM;
                PCR_HIP_CHECK(ctx, hipMemsetAsync(a.slow_count, 0, sizeof(int), ctx->stream));
            }
            nmax = n[k] > nmax ? n[k] : nmax;
        }
        PCR_TRY(pcr_dev_radius_lists_batch(ctx, cp.data(), C, p0.feature_max_nn, p0.feature_radius, nbr.data(), ncnt.data()));
        const FpfhArgs *dfa = pcr_desc_upload(ctx, fa.data(), C);
        if (!dfa) return PCR_ENOMEM;
        const dim3 grid((unsigned)(((size_t)nmax * OCT + FB - 1) / FB), C);
        if (!fa[0].float64_only) {      // float pass, its queue in float64, and (only for a queue that overflowed) every row again in float64
            PCR_LAUNCH(ctx, k_spfh_fast_g, grid, dim3(FB), 0, ctx->stream, dfa);
            PCR_LAUNCH(ctx, k_spfh_slow_g, dim3(std::min<unsigned>(grid.x, 256u), C), dim3(FB), 0, ctx->stream, dfa);
            for (int k = 0; k < C; k++) { fa[k].only_if_over = fa[k].slow_count; fa[k].over_cap = fa[k].slow_cap; fa[k].float64_only = 1; }
            dfa = pcr_desc_upload(ctx, fa.data(), C);
            if (!dfa) return PCR_ENOMEM;
        }
        PCR_LAUNCH(ctx, k_spfh_g, fa[0].only_if_over ? dim3(std::min<unsigned>(grid.x, 64u), C) : grid, dim3(FB), 0, ctx->stream, dfa);
        PCR_LAUNCH(ctx, k_fpfh_g, grid, dim3(FB), 0, ctx->stream, dfa);
    }
    // ---- NormalizePointCloud of the 2 G clouds: means and largest norms come back with the feature search's own wait (wait 2)
    std::vector<double *> P((size_t)C);
    std::vector<Sum3Desc> sd((size_t)C);
    int max_nb = 1;
    constexpr int NROW = 3 + 256;                                         // per cloud: mean (3) + per-block largest norms (<= 256), one contiguous read-back
    double *norm_all = arena<double>(ctx, (size_t)C * NROW);
    if (!norm_all) return PCR_ENOMEM;
    for (int k = 0; k < C; k++) {
        const int nb = (int)((n[k] + FB - 1) / FB < 256 ? (n[k] + FB - 1) / FB : 256);
        P[k] = arena<double>(ctx, (size_t)n[k] * 3);
        double *part = arena<double>(ctx, (size_t)nb * 3);
        if (!P[k] || !part) return PCR_ENOMEM;
        sd[k] = Sum3Desc{xyz[k], (int)n[k], nb, part, norm_all + (size_t)k * NROW, P[k], norm_all + (size_t)k * NROW + 3};
        max_nb = nb > max_nb ? nb : max_nb;
    }
    std::vector<double> hnorm((size_t)C * NROW);
    {
        const Sum3Desc *dsd = pcr_desc_upload(ctx, sd.data(), C);
        if (!dsd) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_sum3_g, dim3(max_nb, C), dim3(FB), 0, ctx->stream, dsd);
        PCR_LAUNCH(ctx, k_sum3_final_g, dim3(C), dim3(64), 0, ctx->stream, dsd);
        PCR_LAUNCH(ctx, k_center_g, dim3(max_nb, C), dim3(FB), 0, ctx->stream, dsd);
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(hnorm.data(), norm_all, sizeof(double) * hnorm.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    // ---- AdvancedMatching: mutual nearest feature rows of all pairs, both directions (i = the larger cloud of a pair)
    std::vector<int> swapped((size_t)G), nPti((size_t)G), nPtj((size_t)G);
    std::vector<int32_t *> j_to_i((size_t)G), i_to_j((size_t)G);
    std::vector<const float *> fi((size_t)G), fj((size_t)G);
    for (int g = 0; g < G; g++) {
        swapped[g] = q[g].nt > q[g].ns ? 1 : 0;
        fi[g] = swapped[g] ? feat[2 * g + 1] : feat[2 * g]; fj[g] = swapped[g] ? feat[2 * g] : feat[2 * g + 1];
        nPti[g] = (int)(swapped[g] ? q[g].nt : q[g].ns); nPtj[g] = (int)(swapped[g] ? q[g].ns : q[g].nt);
        j_to_i[g] = arena<int32_t>(ctx, nPtj[g]); i_to_j[g] = arena<int32_t>(ctx, nPti[g]);
        if (!j_to_i[g] || !i_to_j[g]) return PCR_ENOMEM;
    }
    std::vector<const int *> overflow_dev((size_t)G, nullptr);
    {
        const int rc = pcr_feature_nn_mutual_batch(ctx, G, fi.data(), nPti.data(), fj.data(), nPtj.data(), j_to_i.data(), i_to_j.data(), overflow_dev.data(), 1);
        if (rc == PCR_ECAPACITY) { PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); return 1; }
        if (rc != PCR_OK) return rc;
    }
    std::vector<double> scale_global((size_t)G), scale_start((size_t)G);
    std::vector<std::array<double, 3>> mean((size_t)C);
    for (int g = 0; g < G; g++) {                                       // (the feature search has waited for the stream: the norms are here)
        double mx[2];
        for (int w = 0; w < 2; w++) {
            const double *h = &hnorm[(size_t)(2 * g + w) * NROW];
            for (int d = 0; d < 3; d++) mean[2 * g + w][d] = h[d];
            double m = 0; for (int k = 0; k < sd[2 * g + w].nb; k++) m = h[3 + k] > m ? h[3 + k] : m;
            mx[w] = m;
        }
        const double scale = mx[0] > mx[1] ? mx[0] : mx[1];
        scale_global[g] = q[g].p.option.use_absolute_scale ? 1.0 : scale; scale_start[g] = q[g].p.option.use_absolute_scale ? scale : 1.0;
    }
    {
        std::vector<ScaleDesc> sc; long long mx3 = 0;
        for (int g = 0; g < G; g++) if (scale_global[g] != 1.0)
            for (int w = 0; w < 2; w++) { sc.push_back(ScaleDesc{P[2 * g + w], (long long)n[2 * g + w] * 3, scale_global[g]}); mx3 = (long long)n[2 * g + w] * 3 > mx3 ? (long long)n[2 * g + w] * 3 : mx3; }
        if (!sc.empty()) {
            const ScaleDesc *dsc = pcr_desc_upload(ctx, sc.data(), (int)sc.size());
            if (!dsc) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_scale_g, dim3((unsigned)((mx3 + FB - 1) / FB), (unsigned)sc.size()), dim3(FB), 0, ctx->stream, dsc);
        }
    }
    // ---- cross check (wait 3: the counts, with the overflow flags of the feature search)
    std::vector<uint8_t *> cflags((size_t)G); std::vector<int *> cpos((size_t)G), ncross_dev((size_t)G); std::vector<int32_t *> cross((size_t)G);
    std::vector<CrossDesc> cd((size_t)G); std::vector<int> cap_i((size_t)G);
    int max_i = 1;
    int *counts_all = arena<int>(ctx, (size_t)2 * G);                    // [g]: cross-checked pairs, [G + g]: accepted tuples
    if (!counts_all) return PCR_ENOMEM;
    for (int g = 0; g < G; g++) {
        cflags[g] = arena<uint8_t>(ctx, nPti[g]); cpos[g] = arena<int>(ctx, nPti[g]); ncross_dev[g] = counts_all + g; cross[g] = arena<int32_t>(ctx, (size_t)nPti[g] * 2);
        if (!cflags[g] || !cpos[g] || !cross[g]) return PCR_ENOMEM;
        cd[g] = CrossDesc{i_to_j[g], j_to_i[g], nPti[g], cflags[g], cpos[g], cross[g]};
        cap_i[g] = nPti[g]; max_i = nPti[g] > max_i ? nPti[g] : max_i;
    }
    std::vector<int> ncross((size_t)G, 0), overflow((size_t)G, 0);
    {
        const CrossDesc *dcd = pcr_desc_upload(ctx, cd.data(), G);
        if (!dcd) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_cross_flags_g, dim3((max_i + FB - 1) / FB, G), dim3(FB), 0, ctx->stream, dcd);
        PCR_TRY(pcr_dev_flag_scan_batch(ctx, G, cflags.data(), nullptr, cap_i.data(), cpos.data(), ncross_dev.data()));
        PCR_LAUNCH(ctx, k_cross_emit_g, dim3((max_i + FB - 1) / FB, G), dim3(FB), 0, ctx->stream, dcd);
        std::vector<int> hflags((size_t)2 * G, 0);                        // (the feature search keeps its per-pair flag words in one array, two per pair)
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(ncross.data(), counts_all, sizeof(int) * (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(hflags.data(), overflow_dev[0], sizeof(int) * (size_t)2 * G, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (int g = 0; g < G; g++) { overflow[g] = hflags[2 * g]; if (overflow[g]) { q[g].status = 1; pcr_counters().fgr_group_pool_overflows++; } }        // record pool exhausted: that pair alone takes the float64 path
    }
    // ---- tuple test (wait 4: the accepted counts)
    std::vector<int32_t *> corr((size_t)G, nullptr); std::vector<int64_t> ncorr((size_t)G, 0);
    {
        std::vector<TupleDesc> td; std::vector<int> owner; std::vector<uint8_t *> tfl; std::vector<int *> tps, ntd; std::vector<int> tcap;
        long long max_trials = 0;
        for (int g = 0; g < G; g++) {
            if (q[g].status != PCR_OK || ncross[g] <= 0) continue;
            const pcr_fgr_option &opt = q[g].p.option;
            const long long trials = 100ll * ncross[g];
            uint8_t *tflags = arena<uint8_t>(ctx, trials);
            int *tpos = arena<int>(ctx, trials), *ntup_dev = counts_all + G + g;
            corr[g] = arena<int32_t>(ctx, (size_t)opt.maximum_tuple_count * 6);
            if (!tflags || !tpos || !corr[g]) return PCR_ENOMEM;
            TupleDesc t;
            t.t.pi = swapped[g] ? P[2 * g + 1] : P[2 * g]; t.t.pj = swapped[g] ? P[2 * g] : P[2 * g + 1]; t.t.cross = cross[g]; t.t.ncross = ncross[g];
            t.t.seed = opt.seed; t.t.tuple_scale = opt.tuple_scale; t.t.trials = trials;
            t.flags = tflags; t.pos = tpos; t.max_tuples = opt.maximum_tuple_count; t.swapped = swapped[g]; t.corr = corr[g];
            td.push_back(t); owner.push_back(g); tfl.push_back(tflags); tps.push_back(tpos); ntd.push_back(ntup_dev); tcap.push_back((int)trials);
            max_trials = trials > max_trials ? trials : max_trials;
        }
        if (!td.empty()) {
            const int m = (int)td.size();
            const TupleDesc *dtd = pcr_desc_upload(ctx, td.data(), m);
            if (!dtd) return PCR_ENOMEM;
            const dim3 grid((unsigned)((max_trials + FB - 1) / FB), m);
            PCR_LAUNCH(ctx, k_tuple_flags_g, grid, dim3(FB), 0, ctx->stream, dtd);
            PCR_TRY(pcr_dev_flag_scan_batch(ctx, m, tfl.data(), nullptr, tcap.data(), tps.data(), ntd.data()));
            PCR_LAUNCH(ctx, k_tuple_emit_g, grid, dim3(FB), 0, ctx->stream, dtd);
            std::vector<int> nacc((size_t)G, 0);
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(nacc.data(), counts_all + G, sizeof(int) * (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            for (int k = 0; k < m; k++) {
                const int g = owner[k];
                const int64_t a = nacc[g] > q[g].p.option.maximum_tuple_count ? q[g].p.option.maximum_tuple_count : nacc[g];
                ncorr[g] = a * 3;
            }
        }
    }
    // ---- OptimizePairwiseRegistration: the variant by correspondence count, as the one-pair path picks it (wait 5: the states)
    std::vector<FgrState> hst((size_t)G); std::vector<FgrState *> st((size_t)G, nullptr);
    FgrState *st_all = arena<FgrState>(ctx, G);
    if (!st_all) return PCR_ENOMEM;
    {
        std::vector<OptPrepDesc> pd; std::vector<FgrOptArgs> cls[3]; std::vector<int> who;
        int max_corr = 1;
        for (int g = 0; g < G; g++) {
            if (q[g].status != PCR_OK || ncorr[g] < 10) continue;
            const pcr_fgr_option &opt = q[g].p.option;
            int nb = (int)((ncorr[g] + FB - 1) / FB); if (nb > 256) nb = 256;
            st[g] = st_all + g;
            double *partials = arena<double>(ctx, (size_t)nb * FNVP);
            const int stride = (int)((ncorr[g] + 63) / 64 * 64);
            double *pq = arena<double>(ctx, (size_t)stride * 6);
            if (!st[g] || !partials || !pq) return PCR_ENOMEM;
            pd.push_back(OptPrepDesc{P[2 * g], P[2 * g + 1], corr[g], (int)ncorr[g], stride, pq, st[g], scale_start[g]});
            FgrOptArgs oa;
            oa.pq = pq; oa.stride = stride; oa.ncorr = (int)ncorr[g]; oa.st = st[g]; oa.partials = partials;
            oa.decrease_mu = opt.decrease_mu; oa.max_corr_dist = opt.maximum_correspondence_distance; oa.division_factor = opt.division_factor;
            int variant = fgr_opt_variant(ncorr[g]);
            // the group's multi-workgroup launch has FMG workgroups per pair; the one-pair path takes FMG_MAX from FGR_MULTI_WIDE correspondences (or what
            // PCR_FGR_MULTI_WGS says): other rows, another summation order.  Such a pair goes the one-pair way, so that "group = pair by pair" holds
            // for explicit tuple counts beyond the reference's 0.2 n as well (round-4 advisor finding)
            if (variant == 2 && (ncorr[g] >= FGR_MULTI_WIDE || getenv("PCR_FGR_MULTI_WGS"))) variant = 3;
            if (variant < 3) cls[variant].push_back(oa);
            else { q[g].status = 1; pd.pop_back(); st[g] = nullptr; continue; }              // one launch per iteration / wide form: the one-pair path
            who.push_back(g);
            max_corr = (int)ncorr[g] > max_corr ? (int)ncorr[g] : max_corr;
        }
        if (!pd.empty()) {
            const OptPrepDesc *dpd = pcr_desc_upload(ctx, pd.data(), (int)pd.size());
            if (!dpd) return PCR_ENOMEM;
            PCR_LAUNCH(ctx, k_fgr_prepare_g, dim3((max_corr + FB - 1) / FB, (unsigned)pd.size()), dim3(FB), 0, ctx->stream, dpd);
            const int iters = (int)p0.option.iteration_number;
            for (int v = 0; v < 3; v++) {
                if (cls[v].empty()) continue;
                const int m = (int)cls[v].size();
                const FgrOptArgs *doa = pcr_desc_upload(ctx, cls[v].data(), m);
                if (!doa) return PCR_ENOMEM;
                if (v == 0) PCR_LAUNCH(ctx, k_fgr_opt_single_g<256>, dim3(1, m), dim3(256), 0, ctx->stream, doa, iters);
                else if (v == 1) PCR_LAUNCH(ctx, k_fgr_opt_single_g<FSB>, dim3(1, m), dim3(FSB), 0, ctx->stream, doa, iters);
                else {
                    double *rows = arena<double>(ctx, (size_t)m * 2 * FMG_MAX * FNVP);
                    if (!rows) return PCR_ENOMEM;
                    // FMG x m co-resident 512-thread workgroups (at most 8 x 64 of the chip's 1024 slots of that size)
                    PCR_LAUNCH(ctx, k_fgr_opt_multi_g, dim3(FMG, m), dim3(FMB), 0, ctx->stream, doa, iters, rows, fgr_opt_rule().multi_timeout);
                }
            }
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(hst.data(), st_all, sizeof(FgrState) * (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            for (int g : who) if (hst[g].failed) { q[g].status = 1; pcr_counters().fgr_group_barrier_timeouts++; }     // barrier timeout: the one-pair path reruns it launch by launch (counted: pcr_counter)
        }
    }
    // ---- GetTransformationOriginalScale + inverse per pair (host), then evaluate_registration of all pairs (wait 6)
    std::vector<double> T((size_t)G * 16);
    for (int g = 0; g < G; g++) {
        double trans[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (st[g] && q[g].status == PCR_OK) memcpy(trans, hst[g].trans, sizeof trans);
        const std::array<double, 3> &m0 = mean[2 * g], &m1 = mean[2 * g + 1];
        double To[16] = {0};
        for (int r = 0; r < 3; r++) {
            for (int cc = 0; cc < 3; cc++) To[r * 4 + cc] = trans[r * 4 + cc];
            To[r * 4 + 3] = -(trans[r * 4 + 0] * m1[0] + trans[r * 4 + 1] * m1[1] + trans[r * 4 + 2] * m1[2]) + trans[r * 4 + 3] * scale_global[g] + m0[r];
        }
        To[15] = 1;
        double *Ts = &T[16 * g];
        for (int k = 0; k < 16; k++) Ts[k] = (k % 5 == 0) ? 1.0 : 0.0;
        for (int r = 0; r < 3; r++) {
            for (int cc = 0; cc < 3; cc++) Ts[r * 4 + cc] = To[cc * 4 + r];
            Ts[r * 4 + 3] = -(To[0 * 4 + r] * To[3] + To[1 * 4 + r] * To[7] + To[2 * 4 + r] * To[11]);
        }
        for (int k = 0; k < 16; k++) if (!std::isfinite(Ts[k])) { q[g].status = 1; for (int j = 0; j < 16; j++) Ts[j] = (j % 5 == 0) ? 1.0 : 0.0; break; }
    }
    {
        std::vector<const DevCloud *> ss((size_t)G), tt((size_t)G); std::vector<int32_t *> match((size_t)G); std::vector<pcr_result> res((size_t)G);
        for (int g = 0; g < G; g++) {
            ss[g] = &c[2 * g]; tt[g] = &c[2 * g + 1];
            match[g] = arena<int32_t>(ctx, q[g].ns);
            if (!match[g]) return PCR_ENOMEM;
        }
        PCR_TRY(pcr_dev_evaluate_group(ctx, G, ss.data(), tt.data(), p0.option.maximum_correspondence_distance, T.data(), res.data(), match.data()));
        std::vector<const int32_t *> mm; std::vector<const int *> nn; std::vector<int> cc; std::vector<int32_t *> oo; std::vector<const uint32_t *> sp, tp;
        for (int g = 0; g < G; g++) {
            if (q[g].status != PCR_OK) continue;
            *q[g].result = res[g];
            for (int k = 0; k < 16; k++) q[g].result->transformation[k] = T[16 * g + k];
            if (q[g].correspondences) { mm.push_back(match[g]); nn.push_back(c[2 * g].n); cc.push_back(c[2 * g].cap); oo.push_back((int32_t *)q[g].correspondences); sp.push_back(perm[2 * g]); tp.push_back(perm[2 * g + 1]); }
        }
        // the correspondence sets of the group in three batched launches (they were three per pair)
        if (!mm.empty()) PCR_TRY(pcr_dev_compact_matches_batch(ctx, (int)mm.size(), mm.data(), nn.data(), cc.data(), oo.data(), sp.data(), tp.data()));
    }
    return PCR_OK;
}
