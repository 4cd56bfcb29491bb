// pcr_cloud.hip -- per-cloud stages of the hot path on gfx950:
//   bounds, voxel-grid mean (K1), Morton ordering + implicit BVH (K2), exact k-NN (K3),
//   statistical outlier removal (K4), covariance + analytic 3x3 eigenvector normals (K5).
// Reference behaviour: Open3D PointCloud::{VoxelDownSample, RemoveStatisticalOutliers, EstimateNormals,
// EstimateCovariances} as called at ALL_FUNCTIONS.py:293-302 / 2_MGICP_refinement_in_NCLT_dataset.py:146-153
// (SURVEY.md A.1-A.4).  All kernels are count-driven by a DEVICE-side int so that no host round trip is
// needed between stages.
#include "pcr_device.h"

#define BS 256

// ================================================================================== bounds
__global__ void __launch_bounds__(BS) k_bounds_partial(const float *__restrict__ xyz, int64_t n, float *__restrict__ part) {
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int64_t i = blockIdx.x * (int64_t)BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
        mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
    }
    __shared__ float s[BS / PCR_WAVE][6];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = pcr_wave_min(mn[d]); mx[d] = pcr_wave_max(mx[d]); }
    if (lane == 0) { for (int d = 0; d < 3; d++) { s[w][d] = mn[d]; s[w][3 + d] = mx[d]; } }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[0][threadIdx.x];
        for (int k = 1; k < BS / PCR_WAVE; k++) v = threadIdx.x < 3 ? fminf(v, s[k][threadIdx.x]) : fmaxf(v, s[k][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ void k_bounds_final(const float *__restrict__ part, int nb, float *__restrict__ out6) {
    if (threadIdx.x < 6) {
        float v = part[threadIdx.x];
        for (int k = 1; k < nb; k++) v = threadIdx.x < 3 ? fminf(v, part[k * 6 + threadIdx.x]) : fmaxf(v, part[k * 6 + threadIdx.x]);
        out6[threadIdx.x] = v;
    }
}

int pcr_dev_bounds(pcr_context *ctx, const float *xyz, int64_t n, double *b6) {
    if (n <= 0) { for (int i = 0; i < 6; i++) b6[i] = 0; return PCR_OK; }
    ArenaMark mark(ctx);
    const int nb = (int)((n + BS - 1) / BS < 256 ? (n + BS - 1) / BS : 256);
    float *part = arena<float>(ctx, (size_t)nb * 6 + 6);
    if (!part) return PCR_ENOMEM;
    float *out6 = part + (size_t)nb * 6;
    hipLaunchKernelGGL(k_bounds_partial, dim3(nb), dim3(BS), 0, ctx->stream, xyz, n, part);
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(64), 0, ctx->stream, part, nb, out6);
    float h[6];
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(h, out6, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 6; i++) b6[i] = (double)h[i];
    return PCR_OK;
}

int pcr_read_count(pcr_context *ctx, const int *dev_n, int64_t *out) {
    int h = 0;
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(&h, dev_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = h;
    return PCR_OK;
}

// ========================================================================== flag scan (compaction)
// pos[i] = number of set flags before i; *total = number of set flags.  Three small hand-written kernels
// (tile count -> single-block scan of tile sums -> tile-local scan).  TILE = 4 elements per thread.
#define TILE (BS * 4)

__device__ static inline int block_exclusive_scan(int v, int *total) {   // BS threads, returns exclusive prefix
    __shared__ int wsum[BS / PCR_WAVE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < PCR_WAVE; o <<= 1) { int t = __shfl_up(inc, o, PCR_WAVE); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < BS / PCR_WAVE; k++) { int s = wsum[k]; if (k < w) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(BS) k_scan_tile_count(const uint8_t *__restrict__ flags, const int *__restrict__ n_ptr, int n_host, int *__restrict__ tile_sums) {
    const int n = n_ptr ? *n_ptr : n_host;
    const int base = blockIdx.x * TILE + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j < n) c += flags[base + j] ? 1 : 0;
    int tot; (void)block_exclusive_scan(c, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(BS) k_scan_tile_sums(int *__restrict__ tile_sums, int n_tiles, int *__restrict__ total_out) {
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b = 0; b < n_tiles; b += BS) {
        int i = b + threadIdx.x;
        int v = i < n_tiles ? tile_sums[i] : 0, tot;
        int ex = block_exclusive_scan(v, &tot);
        int c = carry;
        if (i < n_tiles) tile_sums[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ void __launch_bounds__(BS) k_scan_tile_apply(const uint8_t *__restrict__ flags, const int *__restrict__ n_ptr, int n_host, const int *__restrict__ tile_sums, int *__restrict__ pos) {
    const int n = n_ptr ? *n_ptr : n_host;
    const int base = blockIdx.x * TILE + threadIdx.x * 4;
    int f[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { f[j] = (base + j < n && flags[base + j]) ? 1 : 0; c += f[j]; }
    int tot; int ex = block_exclusive_scan(c, &tot) + tile_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++) { if (base + j < n) pos[base + j] = ex; ex += f[j]; }
}

int pcr_dev_flag_scan(pcr_context *ctx, const uint8_t *flags, const int *n_ptr, int n_cap, int *pos, int *total_dev) {
    const int n_tiles = (n_cap + TILE - 1) / TILE;
    int *tile_sums = arena<int>(ctx, (size_t)n_tiles + 1);
    if (!tile_sums) return PCR_ENOMEM;
    hipLaunchKernelGGL(k_scan_tile_count, dim3(n_tiles), dim3(BS), 0, ctx->stream, flags, n_ptr, n_cap, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(BS), 0, ctx->stream, tile_sums, n_tiles, total_dev);
    hipLaunchKernelGGL(k_scan_tile_apply, dim3(n_tiles), dim3(BS), 0, ctx->stream, flags, n_ptr, n_cap, tile_sums, pos);
    return PCR_OK;
}

// ================================================================================== voxel (K1)
__global__ void __launch_bounds__(BS) k_voxel_keys(const float *__restrict__ xyz, int n, double ox, double oy, double oz, double voxel,
                                                   uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    // identical float64 expression to Open3D / the oracle: floor((p - origin) / voxel)
    const double x = (double)xyz[i * 3], y = (double)xyz[i * 3 + 1], z = (double)xyz[i * 3 + 2];
    const uint32_t ix = (uint32_t)(int)floor((x - ox) / voxel);
    const uint32_t iy = (uint32_t)(int)floor((y - oy) / voxel);
    const uint32_t iz = (uint32_t)(int)floor((z - oz) / voxel);
    keys[i] = pcr_morton3(ix, iy, iz);
    vals[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(BS) k_head_flags(const uint64_t *__restrict__ keys, int n, uint8_t *__restrict__ flags) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ void __launch_bounds__(BS) k_voxel_mean(const float *__restrict__ xyz, const float *__restrict__ nrm_in, const uint64_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ vals, const uint8_t *__restrict__ flags, const int *__restrict__ pos, int n,
                                                   float4 *__restrict__ out_pts, float4 *__restrict__ out_nrm) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const uint64_t key = keys[i];
    double sx = 0, sy = 0, sz = 0, nx = 0, ny = 0, nz = 0;
    int j = i;
    do {   // members in input order (stable sort) => same float64 sum as the oracle
        const uint32_t v = vals[j];
        sx += (double)xyz[v * 3]; sy += (double)xyz[v * 3 + 1]; sz += (double)xyz[v * 3 + 2];
        if (nrm_in) { nx += (double)nrm_in[v * 3]; ny += (double)nrm_in[v * 3 + 1]; nz += (double)nrm_in[v * 3 + 2]; }
        j++;
    } while (j < n && keys[j] == key);
    const double c = (double)(j - i);
    const int o = pos[i];
    out_pts[o] = make_float4((float)(sx / c), (float)(sy / c), (float)(sz / c), 0.0f);
    if (nrm_in && out_nrm) out_nrm[o] = make_float4((float)(nx / c), (float)(ny / c), (float)(nz / c), 0.0f);
}

static int bits_for(uint32_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

int pcr_dev_voxel(pcr_context *ctx, const float *xyz, const float *nrm_in, int64_t n, const double *b6, double voxel, DevCloud *out) {
    if (!(voxel > 0.0)) { ctx->err = "voxel_size <= 0"; return PCR_EINVAL; }
    if (n > 0x7fffffff / 4) { ctx->err = "cloud too large"; return PCR_EINVAL; }
    if (n == 0) { PCR_HIP_CHECK(ctx, hipMemsetAsync(out->n, 0, sizeof(int), ctx->stream)); return PCR_OK; }
    const double ox = b6[0] - voxel * 0.5, oy = b6[1] - voxel * 0.5, oz = b6[2] - voxel * 0.5;
    uint32_t mx = 0;
    for (int d = 0; d < 3; d++) {
        double e = floor((b6[3 + d] - (b6[d] - voxel * 0.5)) / voxel);
        if (!(e < 2097152.0)) { ctx->err = "voxel_size is too small"; return PCR_EINVAL; }
        if ((uint32_t)e > mx) mx = (uint32_t)e;
    }
    const int end_bit = 3 * bits_for(mx);
    ArenaMark mark(ctx);
    const int ni = (int)n;
    uint64_t *k0 = arena<uint64_t>(ctx, n), *k1 = arena<uint64_t>(ctx, n);
    uint32_t *v0 = arena<uint32_t>(ctx, n), *v1 = arena<uint32_t>(ctx, n);
    uint8_t *flags = arena<uint8_t>(ctx, n);
    int *pos = arena<int>(ctx, n);
    const size_t tb = pcr_sort_temp_bytes(n);
    void *temp = pcr_arena_alloc(ctx, tb);
    if (!k0 || !k1 || !v0 || !v1 || !flags || !pos || !temp) return PCR_ENOMEM;
    const int nb = (ni + BS - 1) / BS;
    hipLaunchKernelGGL(k_voxel_keys, dim3(nb), dim3(BS), 0, ctx->stream, xyz, ni, ox, oy, oz, voxel, k0, v0);
    PCR_TRY(pcr_sort_pairs(ctx, temp, tb, k0, k1, v0, v1, n, end_bit));
    hipLaunchKernelGGL(k_head_flags, dim3(nb), dim3(BS), 0, ctx->stream, k1, ni, flags);
    PCR_TRY(pcr_dev_flag_scan(ctx, flags, nullptr, ni, pos, out->n));
    hipLaunchKernelGGL(k_voxel_mean, dim3(nb), dim3(BS), 0, ctx->stream, xyz, nrm_in, k1, v1, flags, pos, ni, out->pts, out->nrm);
    return PCR_OK;
}

// ===================================================== Morton ordering of a raw cloud (K2, part 1)
__global__ void __launch_bounds__(BS) k_raw_keys(const float *__restrict__ xyz, int n, float ox, float oy, float oz, float sx, float sy, float sz,
                                                 uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const uint32_t ix = (uint32_t)fminf((xyz[i * 3] - ox) * sx, 65535.0f);
    const uint32_t iy = (uint32_t)fminf((xyz[i * 3 + 1] - oy) * sy, 65535.0f);
    const uint32_t iz = (uint32_t)fminf((xyz[i * 3 + 2] - oz) * sz, 65535.0f);
    keys[i] = pcr_morton3(ix, iy, iz);
    vals[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(BS) k_gather_f3_to_f4(const float *__restrict__ src, const uint32_t *__restrict__ perm, int n, float4 *__restrict__ dst) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = perm ? perm[i] : (uint32_t)i;
    dst[i] = make_float4(src[v * 3], src[v * 3 + 1], src[v * 3 + 2], 0.0f);
}
__global__ void __launch_bounds__(BS) k_scatter_f4_to_f3(const float4 *__restrict__ src, const uint32_t *__restrict__ perm, const int *__restrict__ n_ptr, int n_host,
                                                         float *__restrict__ dst) {
    const int n = n_ptr ? *n_ptr : n_host;
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = perm ? perm[i] : (uint32_t)i;
    const float4 p = src[i];
    dst[v * 3] = p.x; dst[v * 3 + 1] = p.y; dst[v * 3 + 2] = p.z;
}
__global__ void k_set_int(int *p, int v) { *p = v; }

int pcr_dev_gather_f3_to_f4(pcr_context *ctx, const float *src, const uint32_t *perm, int64_t n, float4 *dst) {
    if (n <= 0) return PCR_OK;
    hipLaunchKernelGGL(k_gather_f3_to_f4, dim3((unsigned)((n + BS - 1) / BS)), dim3(BS), 0, ctx->stream, src, perm, (int)n, dst);
    return PCR_OK;
}
int pcr_dev_unpack_f3_to_f4(pcr_context *ctx, const float *src, int64_t n, float4 *dst) { return pcr_dev_gather_f3_to_f4(ctx, src, nullptr, n, dst); }
int pcr_dev_scatter_rows_f4_to_f3(pcr_context *ctx, const float4 *src, const uint32_t *perm, const int *n, int cap, float *dst) {
    if (cap <= 0) return PCR_OK;
    hipLaunchKernelGGL(k_scatter_f4_to_f3, dim3((cap + BS - 1) / BS), dim3(BS), 0, ctx->stream, src, perm, n, cap, dst);
    return PCR_OK;
}
int pcr_dev_pack_f4_to_f3(pcr_context *ctx, const float4 *src, const int *n, int cap, float *dst) { return pcr_dev_scatter_rows_f4_to_f3(ctx, src, nullptr, n, cap, dst); }

int pcr_dev_sort_cloud(pcr_context *ctx, const float *xyz, int64_t n, const double *b6, DevCloud *out, uint32_t *perm) {
    if (n > 0x7fffffff / 4) { ctx->err = "cloud too large"; return PCR_EINVAL; }
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, ctx->stream, out->n, (int)n);
    if (n == 0) return PCR_OK;
    ArenaMark mark(ctx);
    uint64_t *k0 = arena<uint64_t>(ctx, n), *k1 = arena<uint64_t>(ctx, n);
    uint32_t *v0 = arena<uint32_t>(ctx, n);
    const size_t tb = pcr_sort_temp_bytes(n);
    void *temp = pcr_arena_alloc(ctx, tb);
    if (!k0 || !k1 || !v0 || !temp) return PCR_ENOMEM;
    float s[3];
    for (int d = 0; d < 3; d++) { double e = b6[3 + d] - b6[d]; s[d] = e > 0 ? (float)(65535.0 / e) : 0.0f; }
    const int nb = (int)((n + BS - 1) / BS);
    hipLaunchKernelGGL(k_raw_keys, dim3(nb), dim3(BS), 0, ctx->stream, xyz, (int)n, (float)b6[0], (float)b6[1], (float)b6[2], s[0], s[1], s[2], k0, v0);
    PCR_TRY(pcr_sort_pairs(ctx, temp, tb, k0, k1, v0, perm, n, 48));
    hipLaunchKernelGGL(k_gather_f3_to_f4, dim3(nb), dim3(BS), 0, ctx->stream, xyz, perm, (int)n, out->pts);
    return PCR_OK;
}

// ====================================================================== implicit BVH build (K2, part 2)
__global__ void __launch_bounds__(BS) k_bvh_leaves(const float4 *__restrict__ pts, const int *__restrict__ n_ptr, float4 *__restrict__ boxes) {
    __shared__ BvhMeta m;
    const int n = *n_ptr;
    if (threadIdx.x == 0) pcr_bvh_meta(n, m);
    __syncthreads();
    const int leaf = blockIdx.x * BS + threadIdx.x;
    float4 lo = make_float4(3.4e38f, 3.4e38f, 3.4e38f, 0), hi = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, 0);
    if (leaf < m.cnt[0]) {
        const int b = leaf * PCR_LEAF;
#pragma unroll
        for (int j = 0; j < PCR_LEAF; j++) {
            if (b + j < n) {
                const float4 p = pts[b + j];
                lo.x = fminf(lo.x, p.x); lo.y = fminf(lo.y, p.y); lo.z = fminf(lo.z, p.z);
                hi.x = fmaxf(hi.x, p.x); hi.y = fmaxf(hi.y, p.y); hi.z = fmaxf(hi.z, p.z);
            }
        }
        boxes[2 * (size_t)leaf] = lo; boxes[2 * (size_t)leaf + 1] = hi;
    }
    // level 1 = 8 consecutive leaves = 8 consecutive lanes
#pragma unroll
    for (int o = 1; o < PCR_FANOUT; o <<= 1) {
        lo.x = fminf(lo.x, __shfl_xor(lo.x, o, PCR_WAVE)); lo.y = fminf(lo.y, __shfl_xor(lo.y, o, PCR_WAVE)); lo.z = fminf(lo.z, __shfl_xor(lo.z, o, PCR_WAVE));
        hi.x = fmaxf(hi.x, __shfl_xor(hi.x, o, PCR_WAVE)); hi.y = fmaxf(hi.y, __shfl_xor(hi.y, o, PCR_WAVE)); hi.z = fmaxf(hi.z, __shfl_xor(hi.z, o, PCR_WAVE));
    }
    if (m.n_levels > 1 && (threadIdx.x & 7) == 0) {
        const int p = leaf >> 3;
        if (p < m.cnt[1]) { boxes[2 * (size_t)(m.off[1] + p)] = lo; boxes[2 * (size_t)(m.off[1] + p) + 1] = hi; }
    }
}
__global__ void __launch_bounds__(BS) k_bvh_level(const int *__restrict__ n_ptr, float4 *__restrict__ boxes, int level) {
    __shared__ BvhMeta m;
    if (threadIdx.x == 0) pcr_bvh_meta(*n_ptr, m);
    __syncthreads();
    if (level >= m.n_levels) return;
    const int node = blockIdx.x * BS + threadIdx.x;
    if (node >= m.cnt[level]) return;
    float4 lo = make_float4(3.4e38f, 3.4e38f, 3.4e38f, 0), hi = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, 0);
    const int first = node * PCR_FANOUT, cc = m.cnt[level - 1];
    for (int c = 0; c < PCR_FANOUT && first + c < cc; c++) {
        const float4 a = boxes[2 * (size_t)(m.off[level - 1] + first + c)], b = boxes[2 * (size_t)(m.off[level - 1] + first + c) + 1];
        lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z);
        hi.x = fmaxf(hi.x, b.x); hi.y = fmaxf(hi.y, b.y); hi.z = fmaxf(hi.z, b.z);
    }
    boxes[2 * (size_t)(m.off[level] + node)] = lo; boxes[2 * (size_t)(m.off[level] + node) + 1] = hi;
}

int pcr_dev_build_bvh(pcr_context *ctx, DevCloud *c) {
    if (c->cap <= 0) return PCR_OK;
    BvhMeta m; pcr_bvh_meta(c->cap, m);
    if (m.n_levels > 7) { ctx->err = "cloud too large for the 7-level BVH"; return PCR_EINVAL; }
    hipLaunchKernelGGL(k_bvh_leaves, dim3((m.cnt[0] + BS - 1) / BS), dim3(BS), 0, ctx->stream, c->pts, c->n, c->boxes);
    for (int l = 2; l < m.n_levels; l++)
        hipLaunchKernelGGL(k_bvh_level, dim3((m.cnt[l] + BS - 1) / BS), dim3(BS), 0, ctx->stream, c->n, c->boxes, l);
    return PCR_OK;
}

// ====================================================================================== k-NN (K3)
// One query per thread; the running k-best lives in LDS ([slot][thread], conflict-free), candidates come from
// the stack-free BVH walk seeded with the query's Morton neighbours.
#define KNN_BS 128

struct KnnVisitor {
    const float4 *__restrict__ pts;
    float *sd; int *si;        // LDS columns of this thread (stride KNN_BS)
    int n, k, count, wslot, skip_lo, skip_hi;
    float qx, qy, qz, worst, r2cap;
    __device__ float bound() const { return count < k ? r2cap : worst; }
    __device__ void rescan() {
        float w = -1.0f; int ws = 0;
        for (int s = 0; s < k; s++) { float v = sd[s * KNN_BS]; if (v > w) { w = v; ws = s; } }
        worst = w; wslot = ws;
    }
    __device__ void point(int idx, float d2) {
        if (count < k) {
            if (d2 < r2cap) { sd[count * KNN_BS] = d2; si[count * KNN_BS] = idx; count++; if (count == k) rescan(); }
        } else if (d2 < worst) {
            sd[wslot * KNN_BS] = d2; si[wslot * KNN_BS] = idx; rescan();
        }
    }
    __device__ void scan(int leaf) {
        const int b = leaf * PCR_LEAF;
#pragma unroll
        for (int j = 0; j < PCR_LEAF; j++) {
            const int idx = b + j;
            if (idx < n) { const float4 p = pts[idx]; point(idx, pcr_d2(p.x - qx, p.y - qy, p.z - qz)); }
        }
    }
    __device__ void leaf(int l) { if (l < skip_lo || l > skip_hi) scan(l); }
};

// ---- analytic symmetric 3x3 eigenvector of the smallest eigenvalue (Eberly's non-iterative solver)
__device__ static inline void d_cross(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ static inline double d_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// A = [a00 a01 a02; a01 a11 a12; a02 a12 a22] passed as 6 values
__device__ static void d_eigvec0(const double *A, double ev, double *out) {
    double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    d_cross(r0, r1, c01); d_cross(r0, r2, c02); d_cross(r1, r2, c12);
    double d0 = d_dot(c01, c01), d1 = d_dot(c02, c02), d2 = d_dot(c12, c12);
    double dmax = d0; int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) { imax = 2; }
    double s;
    if (imax == 0) { s = sqrt(d0); out[0] = c01[0] / s; out[1] = c01[1] / s; out[2] = c01[2] / s; }
    else if (imax == 1) { s = sqrt(d1); out[0] = c02[0] / s; out[1] = c02[1] / s; out[2] = c02[2] / s; }
    else { s = sqrt(d2); out[0] = c12[0] / s; out[1] = c12[1] / s; out[2] = c12[2] / s; }
}
__device__ static void d_eigvec1(const double *A, const double *e0, double ev1, double *out) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) { double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]); U[0] = -e0[2] * inv; U[1] = 0; U[2] = e0[0] * inv; }
    else { double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]); U[0] = 0; U[1] = e0[2] * inv; U[2] = -e0[1] * inv; }
    d_cross(e0, U, V);
    double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2], A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
    double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2], A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
    double m00 = d_dot(U, AU) - ev1, m01 = d_dot(U, AV), m11 = d_dot(V, AV) - ev1;
    double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
            for (int k = 0; k < 3; k++) out[k] = m01 * U[k] - m00 * V[k];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    } else {
        if (fmax(a11, a01) > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
            for (int k = 0; k < 3; k++) out[k] = m11 * U[k] - m01 * V[k];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    }
}
__device__ static void d_fast_eigen3x3(const double *C6, double *nv) {
    double A[6];
    double mc = C6[0];
    for (int k = 1; k < 6; k++) mc = fmax(mc, C6[k]);
    if (mc == 0.0) { nv[0] = nv[1] = nv[2] = 0; return; }
    for (int k = 0; k < 6; k++) A[k] = C6[k] / mc;
    const double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (norm > 0) {
        const double q = (A[0] + A[3] + A[5]) / 3.0;
        const double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
        const double c00 = b11 * b22 - A[4] * A[4], c01 = A[1] * b22 - A[4] * A[2], c02 = A[1] * A[4] - b11 * A[2];
        const double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double hd = det * 0.5; hd = fmin(fmax(hd, -1.0), 1.0);
        const double angle = acos(hd) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        const double beta2 = cos(angle) * 2.0, beta0 = cos(angle + two_thirds_pi) * 2.0, beta1 = -(beta0 + beta2);
        const double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
        double e0[3], e1[3], e2[3];
        if (hd >= 0) {
            d_eigvec0(A, ev2, e2);
            if (ev2 < ev0 && ev2 < ev1) { nv[0] = e2[0]; nv[1] = e2[1]; nv[2] = e2[2]; return; }
            d_eigvec1(A, e2, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) { nv[0] = e1[0]; nv[1] = e1[1]; nv[2] = e1[2]; return; }
            d_cross(e1, e2, e0);
            nv[0] = e0[0]; nv[1] = e0[1]; nv[2] = e0[2];
        } else {
            d_eigvec0(A, ev0, e0);
            if (ev0 < ev1 && ev0 < ev2) { nv[0] = e0[0]; nv[1] = e0[1]; nv[2] = e0[2]; return; }
            d_eigvec1(A, e0, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) { nv[0] = e1[0]; nv[1] = e1[1]; nv[2] = e1[2]; return; }
            d_cross(e0, e1, e2);
            nv[0] = e2[0]; nv[1] = e2[1]; nv[2] = e2[2];
        }
    } else {
        // diagonal matrix: axis of the smallest diagonal entry, (0,0,1) on ties
        if (A[0] < A[3] && A[0] < A[5]) { nv[0] = 1; nv[1] = 0; nv[2] = 0; }
        else if (A[3] < A[0] && A[3] < A[5]) { nv[0] = 0; nv[1] = 1; nv[2] = 0; }
        else { nv[0] = 0; nv[1] = 0; nv[2] = 1; }
    }
}

enum { KNN_MODE_SOR = 0, KNN_MODE_NORMALS = 1, KNN_MODE_DEBUG = 2 };

struct KnnArgs {
    const float4 *pts; const float4 *boxes; const int *n_ptr;
    int k; float r2cap_f; double r2cap;
    double *avg;                       // SOR
    const float4 *prior; float4 *normals; float *cov6;   // normals
    int32_t *dbg_idx; float *dbg_d2; int32_t *dbg_cnt;   // debug
};

template <int MODE>
__global__ void __launch_bounds__(KNN_BS) k_knn(KnnArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ BvhMeta m;
    const int n = *a.n_ptr;
    if (threadIdx.x == 0) pcr_bvh_meta(n, m);
    __syncthreads();
    const int i = blockIdx.x * KNN_BS + threadIdx.x;
    if (i >= n) return;
    float *sd = (float *)smem + threadIdx.x;
    int *si = (int *)(smem + sizeof(float) * (size_t)a.k * KNN_BS) + threadIdx.x;
    const float4 q = a.pts[i];
    KnnVisitor v;
    v.pts = a.pts; v.sd = sd; v.si = si; v.n = n; v.k = a.k; v.count = 0; v.wslot = 0; v.worst = 3.4e38f; v.r2cap = a.r2cap_f;
    v.qx = q.x; v.qy = q.y; v.qz = q.z;
    // seed with the Morton neighbourhood: enough leaves around the query's own leaf to hold k points
    const int own = i / PCR_LEAF, span = (a.k + PCR_LEAF - 1) / PCR_LEAF / 2 + 1;
    int lo = own - span, hi = own + span;
    if (lo < 0) lo = 0;
    if (hi > m.cnt[0] - 1) hi = m.cnt[0] - 1;
    v.skip_lo = 1; v.skip_hi = 0;
    for (int l = lo; l <= hi; l++) v.scan(l);
    v.skip_lo = lo; v.skip_hi = hi;
    pcr_bvh_traverse(a.boxes, m, q.x, q.y, q.z, v);

    // ---- epilogue in float64 on the selected neighbours (inputs are exact float32 -> same values as the oracle)
    const double qx = q.x, qy = q.y, qz = q.z;
    if (MODE == KNN_MODE_SOR) {
        double s = 0; int c = 0;
        for (int t = 0; t < v.count; t++) {
            const float4 p = a.pts[si[t * KNN_BS]];
            const double dx = (double)p.x - qx, dy = (double)p.y - qy, dz = (double)p.z - qz;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < a.r2cap) { s += sqrt(d2); c++; }
        }
        a.avg[i] = c > 0 ? s / (double)c : -1.0;
    } else if (MODE == KNN_MODE_NORMALS) {
        double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; int c = 0;
        for (int t = 0; t < v.count; t++) {
            const float4 p = a.pts[si[t * KNN_BS]];
            const double x = p.x, y = p.y, z = p.z;
            const double dx = x - qx, dy = y - qy, dz = z - qz;
            if (dx * dx + dy * dy + dz * dz < a.r2cap) {
                cu[0] += x; cu[1] += y; cu[2] += z;
                cu[3] += x * x; cu[4] += x * y; cu[5] += x * z; cu[6] += y * y; cu[7] += y * z; cu[8] += z * z;
                c++;
            }
        }
        double C6[6];
        if (c >= 3) {
            const double inv = 1.0 / (double)c;   // cumulants /= n, as Open3D
            for (int t = 0; t < 9; t++) cu[t] = cu[t] / (double)c;
            (void)inv;
            C6[0] = cu[3] - cu[0] * cu[0]; C6[1] = cu[4] - cu[0] * cu[1]; C6[2] = cu[5] - cu[0] * cu[2];
            C6[3] = cu[6] - cu[1] * cu[1]; C6[4] = cu[7] - cu[1] * cu[2]; C6[5] = cu[8] - cu[2] * cu[2];
        } else { C6[0] = 1; C6[1] = 0; C6[2] = 0; C6[3] = 1; C6[4] = 0; C6[5] = 1; }
        if (a.cov6) { for (int t = 0; t < 6; t++) a.cov6[(size_t)i * 6 + t] = (float)C6[t]; }
        if (a.normals) {
            double nv[3];
            d_fast_eigen3x3(C6, nv);
            const double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
            double px = 0, py = 0, pz = 0;
            if (a.prior) { const float4 pr = a.prior[i]; px = pr.x; py = pr.y; pz = pr.z; }
            if (nn == 0.0 || !(nn == nn)) { if (a.prior) { nv[0] = px; nv[1] = py; nv[2] = pz; } else { nv[0] = 0; nv[1] = 0; nv[2] = 1; } }
            if (a.prior && nv[0] * px + nv[1] * py + nv[2] * pz < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
            a.normals[i] = make_float4((float)nv[0], (float)nv[1], (float)nv[2], 0.0f);
        }
    } else {
        // debug: selection-sort the k-best ascending (d2, idx) and emit
        for (int t = 0; t < v.count; t++) {
            int best = t; float bd = sd[t * KNN_BS]; int bi = si[t * KNN_BS];
            for (int u = t + 1; u < v.count; u++) {
                float d = sd[u * KNN_BS]; int ix = si[u * KNN_BS];
                if (d < bd || (d == bd && ix < bi)) { best = u; bd = d; bi = ix; }
            }
            if (best != t) { sd[best * KNN_BS] = sd[t * KNN_BS]; si[best * KNN_BS] = si[t * KNN_BS]; sd[t * KNN_BS] = bd; si[t * KNN_BS] = bi; }
            a.dbg_idx[(size_t)i * a.k + t] = bi; a.dbg_d2[(size_t)i * a.k + t] = bd;
        }
        for (int t = v.count; t < a.k; t++) { a.dbg_idx[(size_t)i * a.k + t] = -1; a.dbg_d2[(size_t)i * a.k + t] = __builtin_inff(); }
        if (a.dbg_cnt) a.dbg_cnt[i] = v.count;
    }
}

template <int MODE>
static int launch_knn(pcr_context *ctx, const DevCloud *c, KnnArgs a) {
    if (c->cap <= 0) return PCR_OK;
    const size_t lds = (size_t)a.k * KNN_BS * 8;
    if (a.k < 1 || lds > 150 * 1024) { ctx->err = "k out of range for the per-thread k-NN kernel (1..150)"; return PCR_EINVAL; }
    static bool attr_set[3] = {false, false, false};
    if (!attr_set[MODE]) {
        PCR_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_knn<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set[MODE] = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_knn<MODE>), dim3((c->cap + KNN_BS - 1) / KNN_BS), dim3(KNN_BS), lds, ctx->stream, a);
    return PCR_OK;
}

static void knn_radius(KnnArgs &a, int search_kind, double radius) {
    if (search_kind == PCR_SEARCH_HYBRID && radius > 0) {
        a.r2cap = radius * radius;
        a.r2cap_f = (float)(a.r2cap * (1.0 + 1e-6));     // float32 walk slightly wide, exact float64 test in the epilogue
    } else { a.r2cap = 1e300; a.r2cap_f = 3.4e38f; }
}

int pcr_dev_knn_debug(pcr_context *ctx, const DevCloud *c, int k, double radius, int32_t *idx, float *d2, int32_t *counts) {
    KnnArgs a = {};
    a.pts = c->pts; a.boxes = c->boxes; a.n_ptr = c->n; a.k = k;
    knn_radius(a, radius > 0 ? PCR_SEARCH_HYBRID : PCR_SEARCH_KNN, radius);
    if (radius > 0) a.r2cap_f = (float)(radius * radius);
    a.dbg_idx = idx; a.dbg_d2 = d2; a.dbg_cnt = counts;
    return launch_knn<KNN_MODE_DEBUG>(ctx, c, a);
}

// ============================================================================ SOR (K4)
// mean / Bessel std of the per-point mean neighbour distance, one block, fixed summation tree
__global__ void __launch_bounds__(1024) k_sor_stats(const double *__restrict__ avg, const int *__restrict__ n_ptr, double std_ratio, double *__restrict__ out3) {
    __shared__ double s[1024];
    __shared__ long long sc[1024];
    __shared__ double mean_s;
    const int n = *n_ptr, t = threadIdx.x;
    double a = 0; long long c = 0;
    for (int i = t; i < n; i += 1024) { double v = avg[i]; if (v > 0) { a += v; c++; } }
    s[t] = a; sc[t] = c;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) { s[t] += s[t + o]; sc[t] += sc[t + o]; } __syncthreads(); }
    const long long valid = sc[0];
    if (t == 0) mean_s = valid > 0 ? s[0] / (double)valid : 0.0;
    __syncthreads();
    const double mean = mean_s;
    a = 0;
    for (int i = t; i < n; i += 1024) { double v = avg[i]; if (v > 0) a += (v - mean) * (v - mean); }
    __syncthreads();
    s[t] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) s[t] += s[t + o]; __syncthreads(); }
    if (t == 0) {
        const double sd = sqrt(s[0] / (double)(valid - 1));
        out3[0] = mean; out3[1] = sd; out3[2] = valid > 0 ? mean + std_ratio * sd : -1.0;
    }
}
__global__ void __launch_bounds__(BS) k_sor_flags(const double *__restrict__ avg, const int *__restrict__ n_ptr, const double *__restrict__ stats3, uint8_t *__restrict__ flags) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= *n_ptr) return;
    const double v = avg[i];
    flags[i] = (v > 0 && v < stats3[2]) ? 1 : 0;
}
__global__ void __launch_bounds__(BS) k_compact_cloud(const float4 *__restrict__ pts, const float4 *__restrict__ nrm, const uint8_t *__restrict__ flags, const int *__restrict__ pos,
                                                      const int *__restrict__ n_ptr, float4 *__restrict__ out_pts, float4 *__restrict__ out_nrm) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= *n_ptr || !flags[i]) return;
    const int o = pos[i];
    out_pts[o] = pts[i];
    if (nrm && out_nrm) out_nrm[o] = nrm[i];
}

int pcr_dev_sor(pcr_context *ctx, const DevCloud *in, int nb_neighbors, double std_ratio, DevCloud *out, uint8_t *keep_sorted, double *avg_sorted) {
    if (nb_neighbors < 1 || !(std_ratio > 0.0)) { ctx->err = "nb_neighbors < 1 or std_ratio <= 0"; return PCR_EINVAL; }
    if (in->cap <= 0) { PCR_HIP_CHECK(ctx, hipMemsetAsync(out->n, 0, sizeof(int), ctx->stream)); return PCR_OK; }
    ArenaMark mark(ctx);
    double *avg = avg_sorted ? avg_sorted : arena<double>(ctx, in->cap);
    double *stats3 = arena<double>(ctx, 4);
    uint8_t *flags = keep_sorted ? keep_sorted : arena<uint8_t>(ctx, in->cap);
    int *pos = arena<int>(ctx, in->cap);
    if (!avg || !stats3 || !flags || !pos) return PCR_ENOMEM;
    KnnArgs a = {};
    a.pts = in->pts; a.boxes = in->boxes; a.n_ptr = in->n; a.k = nb_neighbors; a.avg = avg;
    knn_radius(a, PCR_SEARCH_KNN, 0);
    PCR_TRY(launch_knn<KNN_MODE_SOR>(ctx, in, a));
    hipLaunchKernelGGL(k_sor_stats, dim3(1), dim3(1024), 0, ctx->stream, avg, in->n, std_ratio, stats3);
    const int nb = (in->cap + BS - 1) / BS;
    hipLaunchKernelGGL(k_sor_flags, dim3(nb), dim3(BS), 0, ctx->stream, avg, in->n, stats3, flags);
    PCR_TRY(pcr_dev_flag_scan(ctx, flags, in->n, in->cap, pos, out->n));
    hipLaunchKernelGGL(k_compact_cloud, dim3(nb), dim3(BS), 0, ctx->stream, in->pts, in->nrm, flags, pos, in->n, out->pts, out->nrm);
    return PCR_OK;
}

// ================================================================== covariances / normals (K5)
int pcr_dev_normals(pcr_context *ctx, DevCloud *c, int search_kind, int knn, double radius, const float4 *prior, float4 *normals_out, float *cov6_out) {
    if (search_kind == PCR_SEARCH_RADIUS) { ctx->err = "pure radius search not implemented on device yet"; return PCR_EINVAL; }
    if (knn < 1) { ctx->err = "knn < 1"; return PCR_EINVAL; }
    if (search_kind == PCR_SEARCH_HYBRID && !(radius > 0)) { ctx->err = "radius <= 0"; return PCR_EINVAL; }
    KnnArgs a = {};
    a.pts = c->pts; a.boxes = c->boxes; a.n_ptr = c->n; a.k = knn; a.prior = prior; a.normals = normals_out; a.cov6 = cov6_out;
    knn_radius(a, search_kind, radius);
    return launch_knn<KNN_MODE_NORMALS>(ctx, c, a);
}

size_t pcr_scratch_bytes_for(int64_t n) {
    // voxel/sort temporaries (2x u64 keys, 2x u32 vals, flags, pos, sort temp) + clouds + boxes, with slack
    return (size_t)(n > 0 ? n : 1) * 160 + pcr_sort_temp_bytes((size_t)(n > 0 ? n : 1)) + (4u << 20);
}
