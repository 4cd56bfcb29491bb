// pcr_cloud.hip -- per-cloud stages of the hot path on gfx950:
//   bounds, voxel-grid mean (K1), Morton ordering + implicit BVH (K2), exact k-NN (K3),
//   statistical outlier removal (K4), covariance + analytic 3x3 eigenvector normals (K5).
// Reference behaviour: Open3D PointCloud::{VoxelDownSample, RemoveStatisticalOutliers, EstimateNormals,
// EstimateCovariances} as called at ALL_FUNCTIONS.py:293-302 / 2_MGICP_refinement_in_NCLT_dataset.py:146-153
// (SURVEY.md A.1-A.4).  All kernels are count-driven by a DEVICE-side int so that no host round trip is
// needed between stages.
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#include "pcr_octree.h"

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
    PCR_LAUNCH(ctx, k_bounds_partial, dim3(nb), dim3(BS), 0, ctx->stream, xyz, n, part);
    PCR_LAUNCH(ctx, k_bounds_final, dim3(1), dim3(64), 0, ctx->stream, part, nb, out6);
    float h[6];
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(h, out6, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 6; i++) b6[i] = (double)h[i];
    return PCR_OK;
}

// bounds of `count` clouds: two launches (blockIdx.y = cloud), one copy, one synchronisation
struct BoundsArgs { const float *xyz; int64_t n; float *part; float *out6; int nb; };
__global__ void __launch_bounds__(BS) k_bounds_partial_g(const BoundsArgs *a_) {
    const BoundsArgs &a = a_[blockIdx.y];
    if ((int)blockIdx.x >= a.nb) return;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int64_t i = blockIdx.x * (int64_t)BS + threadIdx.x; i < a.n; i += (int64_t)a.nb * BS) {
        float x = a.xyz[i * 3], y = a.xyz[i * 3 + 1], z = a.xyz[i * 3 + 2];
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
        a.part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ void k_bounds_final_g(const BoundsArgs *a_) {
    const BoundsArgs &a = a_[blockIdx.x];
    if (threadIdx.x < 6) {
        float v = a.part[threadIdx.x];
        for (int k = 1; k < a.nb; k++) v = threadIdx.x < 3 ? fminf(v, a.part[k * 6 + threadIdx.x]) : fmaxf(v, a.part[k * 6 + threadIdx.x]);
        a.out6[threadIdx.x] = v;
    }
}
int pcr_dev_bounds_batch(pcr_context *ctx, int count, const float *const *xyz, const int64_t *n, double *b6 /* count x 6, host */) {
    if (count < 1) return PCR_OK;
    ArenaMark mark(ctx);
    std::vector<BoundsArgs> a((size_t)count);
    float *out = arena<float>(ctx, (size_t)count * 6);
    if (!out) return PCR_ENOMEM;
    int max_nb = 1;
    for (int c = 0; c < count; c++) {
        if (n[c] <= 0) { ctx->err = "empty cloud in a group"; return PCR_EINVAL; }
        const int nb = (int)((n[c] + BS - 1) / BS < 256 ? (n[c] + BS - 1) / BS : 256);
        a[c].xyz = xyz[c]; a[c].n = n[c]; a[c].nb = nb; a[c].part = arena<float>(ctx, (size_t)nb * 6); a[c].out6 = out + 6 * c;
        if (!a[c].part) return PCR_ENOMEM;
        max_nb = nb > max_nb ? nb : max_nb;
    }
    const BoundsArgs *d = pcr_desc_upload(ctx, a.data(), count);
    if (!d) return PCR_ENOMEM;
    PCR_LAUNCH(ctx, k_bounds_partial_g, dim3(max_nb, count), dim3(BS), 0, ctx->stream, d);
    PCR_LAUNCH(ctx, k_bounds_final_g, dim3(count), dim3(64), 0, ctx->stream, d);
    std::vector<float> h((size_t)count * 6);
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), out, sizeof(float) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < h.size(); i++) b6[i] = (double)h[i];
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

// Optional flag producers fused into the counting pass (each saved a launch of its own): head flags of sorted keys
// (voxel grid), or the SOR keep test on the mean neighbour distances.  The flags are stored for the later kernels.
struct FlagSrc { const uint64_t *keys; const double *avg; const double *stats3; };
#define PCR_MAX_BATCH 8          // problems whose argument structs travel in the kernel arguments (blockIdx.y picks the problem): the scales of
                                 //   a multiscale registration.  Larger batches (the clouds and scales of a GROUP of pairs) read their argument structs
                                 //   from device memory: every batched kernel has a by-value and a by-pointer entry point (pcr_batch_launch)
#define PCR_MAX_GROUP_BATCH 256
struct ScanArgs { uint8_t *flags; const int *n_ptr; int n_host; int *tile_cnt; int *pos; int *total; FlagSrc src; int n_tiles; };
struct ScanBatch { ScanArgs a[PCR_MAX_BATCH]; };
__device__ static inline void d_scan_tile_count(const ScanArgs &a) {
    if ((int)blockIdx.x >= a.n_tiles) return;
    const int n = a.n_ptr ? *a.n_ptr : a.n_host;
    const int base = blockIdx.x * TILE + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = base + j;
        if (i < n) {
            int f;
            if (a.src.keys) { f = (i == 0 || a.src.keys[i] != a.src.keys[i - 1]) ? 1 : 0; a.flags[i] = (uint8_t)f; }
            else if (a.src.avg) { const double v = a.src.avg[i]; f = (v > 0 && v < a.src.stats3[2]) ? 1 : 0; a.flags[i] = (uint8_t)f; }
            else f = a.flags[i] ? 1 : 0;
            c += f;
        }
    }
    int tot; (void)block_exclusive_scan(c, &tot);
    if (threadIdx.x == 0) a.tile_cnt[blockIdx.x] = tot;
}
// exclusive positions: every tile first adds up the counts of the tiles before it (a few hundred ints, one read per lane)
// instead of waiting for a separate single-workgroup scan of the tile counts -- one launch less per scan
__device__ static inline void d_scan_tile_apply(const ScanArgs &a) {
    __shared__ int wsum[BS / 64];
    if ((int)blockIdx.x >= a.n_tiles) return;
    const int n = a.n_ptr ? *a.n_ptr : a.n_host;
    int before = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += BS) before += a.tile_cnt[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = before;
    __syncthreads();
    int offset = 0;
#pragma unroll
    for (int w = 0; w < BS / 64; w++) offset += wsum[w];
    const int base = blockIdx.x * TILE + threadIdx.x * 4;
    int f[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { f[j] = (base + j < n && a.flags[base + j]) ? 1 : 0; c += f[j]; }
    int tot; int ex = block_exclusive_scan(c, &tot) + offset;
#pragma unroll
    for (int j = 0; j < 4; j++) { if (base + j < n) a.pos[base + j] = ex; ex += f[j]; }
    if ((int)blockIdx.x == a.n_tiles - 1 && threadIdx.x == 0) *a.total = offset + tot;
}
__global__ void __launch_bounds__(BS) k_scan_tile_count(ScanArgs a) { d_scan_tile_count(a); }
__global__ void __launch_bounds__(BS) k_scan_tile_apply(ScanArgs a) { d_scan_tile_apply(a); }
__global__ void __launch_bounds__(BS) k_scan_tile_count_batch(ScanBatch b) { d_scan_tile_count(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_scan_tile_apply_batch(ScanBatch b) { d_scan_tile_apply(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_scan_tile_count_batchp(const ScanArgs *a) { d_scan_tile_count(a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_scan_tile_apply_batchp(const ScanArgs *a) { d_scan_tile_apply(a[blockIdx.y]); }

// `count` problems in one launch: argument structs by value up to PCR_MAX_BATCH, else through the context's descriptor buffer
// (pinned staging -> device, one small asynchronous copy on the launch stream)
template <class B, class A, class KV, class KP>
static int pcr_batch_launch(pcr_context *ctx, const char *file, int line, KV by_value, KP by_pointer, const A *args, int count, dim3 grid, dim3 block) {
    if (count <= PCR_MAX_BATCH) {
        B b; std::memset(&b, 0, sizeof b);
        for (int k = 0; k < count; k++) b.a[k] = args[k];
        pcr_launch(ctx, file, line, by_value, grid, block, 0, ctx->stream, b);
    } else {
        const A *dev = pcr_desc_upload(ctx, args, count);
        if (!dev) return PCR_ENOMEM;
        pcr_launch(ctx, file, line, by_pointer, grid, block, 0, ctx->stream, dev);
    }
    return PCR_OK;
}
#define PCR_BATCH_LAUNCH(ctx, B, kv, kp, args, count, grid, block) pcr_batch_launch<B>(ctx, __FILE__, __LINE__, kv, kp, args, count, grid, block)

static int scan_args(pcr_context *ctx, ScanArgs *a, uint8_t *flags, const int *n_ptr, int n_cap, int *pos, int *total_dev, FlagSrc src) {
    a->n_tiles = (n_cap + TILE - 1) / TILE;
    a->tile_cnt = arena<int>(ctx, (size_t)a->n_tiles + 1);
    if (!a->tile_cnt) return PCR_ENOMEM;
    a->flags = flags; a->n_ptr = n_ptr; a->n_host = n_cap; a->pos = pos; a->total = total_dev; a->src = src;
    return PCR_OK;
}
static int flag_scan(pcr_context *ctx, uint8_t *flags, const int *n_ptr, int n_cap, int *pos, int *total_dev, FlagSrc src) {
    ScanArgs a;
    PCR_TRY(scan_args(ctx, &a, flags, n_ptr, n_cap, pos, total_dev, src));
    PCR_LAUNCH(ctx, k_scan_tile_count, dim3(a.n_tiles), dim3(BS), 0, ctx->stream, a);
    PCR_LAUNCH(ctx, k_scan_tile_apply, dim3(a.n_tiles), dim3(BS), 0, ctx->stream, a);
    return PCR_OK;
}
static int flag_scan_batch(pcr_context *ctx, const ScanArgs *a, int count) {
    int mt = 0;
    for (int k = 0; k < count; k++) mt = a[k].n_tiles > mt ? a[k].n_tiles : mt;
    if (mt == 0) return PCR_OK;
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, ScanBatch, k_scan_tile_count_batch, k_scan_tile_count_batchp, a, count, dim3(mt, count), dim3(BS)));
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, ScanBatch, k_scan_tile_apply_batch, k_scan_tile_apply_batchp, a, count, dim3(mt, count), dim3(BS)));
    return PCR_OK;
}
int pcr_dev_flag_scan(pcr_context *ctx, const uint8_t *flags, const int *n_ptr, int n_cap, int *pos, int *total_dev) {
    return flag_scan(ctx, const_cast<uint8_t *>(flags), n_ptr, n_cap, pos, total_dev, FlagSrc{nullptr, nullptr, nullptr});
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
__global__ void __launch_bounds__(BS) k_voxel_mean(const float *__restrict__ xyz, const float *__restrict__ nrm_in, const uint64_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ vals, const uint8_t *__restrict__ flags, const int *__restrict__ pos, int n,
                                                   float4 *__restrict__ out_pts, float4 *__restrict__ out_nrm, uint64_t *__restrict__ out_keys) {
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
    out_keys[o] = key;
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
    out->key_org[0] = (float)ox; out->key_org[1] = (float)oy; out->key_org[2] = (float)oz;
    out->key_unit[0] = out->key_unit[1] = out->key_unit[2] = (float)voxel; out->voxel_lattice = true;
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
    PCR_LAUNCH(ctx, k_voxel_keys, dim3(nb), dim3(BS), 0, ctx->stream, xyz, ni, ox, oy, oz, voxel, k0, v0);
    PCR_TRY(pcr_sort_pairs(ctx, temp, tb, k0, k1, v0, v1, n, end_bit));
    PCR_TRY(flag_scan(ctx, flags, nullptr, ni, pos, out->n, FlagSrc{k1, nullptr, nullptr}));      // head flags produced inside the scan
    PCR_LAUNCH(ctx, k_voxel_mean, dim3(nb), dim3(BS), 0, ctx->stream, xyz, nrm_in, k1, v1, flags, pos, ni, out->pts, out->nrm, out->keys);
    return PCR_OK;
}

// ---- the voxel stage of ALL scales of a multiscale registration in ONE keys / sort / scan / mean pass: element e = s * n + i is
// point i on the grid of scale s, the scale index sits above the Morton bits of the key.  Every scale owns exactly n consecutive
// elements of the sorted order, so its first voxel is pos[s * n] and no count has to come back to the host.  Same member order
// inside a voxel as the one-scale pass (stable sort, original index order) => bit-identical means; 13 launches instead of 13 S.
#define VOX_MAX_SCALES 8
struct VoxelGrids { double ox[VOX_MAX_SCALES], oy[VOX_MAX_SCALES], oz[VOX_MAX_SCALES], voxel[VOX_MAX_SCALES]; };
struct VoxelOuts { float4 *pts[VOX_MAX_SCALES]; float4 *nrm[VOX_MAX_SCALES]; uint64_t *keys[VOX_MAX_SCALES]; int *n[VOX_MAX_SCALES]; };
struct VoxArgs {
    const float *xyz, *nrm_in; int n, n_scales, shift; VoxelGrids g;
    uint64_t *keys_raw; uint32_t *vals_raw;                    // written by the key kernel
    const uint64_t *keys; const uint32_t *vals;                // sorted
    const uint8_t *flags; const int *pos; const int *total; VoxelOuts o;
};
__device__ static inline void d_voxel_keys_multi(const VoxArgs &a) {
    const int e = blockIdx.x * BS + threadIdx.x;
    const int n = a.n;
    if (e >= n * a.n_scales) return;
    const int s = e / n, i = e - s * n;
    const double x = (double)a.xyz[i * 3], y = (double)a.xyz[i * 3 + 1], z = (double)a.xyz[i * 3 + 2];
    double ox = a.g.ox[0], oy = a.g.oy[0], oz = a.g.oz[0], voxel = a.g.voxel[0];
#pragma unroll
    for (int k = 1; k < VOX_MAX_SCALES; k++) if (s == k) { ox = a.g.ox[k]; oy = a.g.oy[k]; oz = a.g.oz[k]; voxel = a.g.voxel[k]; }
    const uint32_t ix = (uint32_t)(int)floor((x - ox) / voxel);        // identical float64 expression to the one-scale kernel
    const uint32_t iy = (uint32_t)(int)floor((y - oy) / voxel);
    const uint32_t iz = (uint32_t)(int)floor((z - oz) / voxel);
    a.keys_raw[e] = ((uint64_t)s << a.shift) | pcr_morton3(ix, iy, iz);
    a.vals_raw[e] = (uint32_t)i;
}
__device__ static inline void d_voxel_mean_multi(const VoxArgs &a) {
    const int p = blockIdx.x * BS + threadIdx.x;
    const int n = a.n, ne = n * a.n_scales;
    if (p >= ne) return;
    const int s = p / n;
    const int base = a.pos[s * n];                     // element s * n starts scale s: always a voxel head
    float4 *out_pts = a.o.pts[0], *out_nrm = a.o.nrm[0]; uint64_t *out_keys = a.o.keys[0]; int *out_n = a.o.n[0];
#pragma unroll
    for (int k = 1; k < VOX_MAX_SCALES; k++) if (s == k) { out_pts = a.o.pts[k]; out_nrm = a.o.nrm[k]; out_keys = a.o.keys[k]; out_n = a.o.n[k]; }
    if (p == s * n) *out_n = (s + 1 < a.n_scales ? a.pos[(s + 1) * n] : *a.total) - base;
    if (!a.flags[p]) return;
    const uint64_t key = a.keys[p];
    double sx = 0, sy = 0, sz = 0, nx = 0, ny = 0, nz = 0;
    int j = p;
    do {   // members in input order (stable sort) => same float64 sum as the oracle
        const uint32_t v = a.vals[j];
        sx += (double)a.xyz[v * 3]; sy += (double)a.xyz[v * 3 + 1]; sz += (double)a.xyz[v * 3 + 2];
        if (a.nrm_in) { nx += (double)a.nrm_in[v * 3]; ny += (double)a.nrm_in[v * 3 + 1]; nz += (double)a.nrm_in[v * 3 + 2]; }
        j++;
    } while (j < ne && a.keys[j] == key);
    const double c = (double)(j - p);
    const int idx = a.pos[p] - base;
    out_pts[idx] = make_float4((float)(sx / c), (float)(sy / c), (float)(sz / c), 0.0f);
    out_keys[idx] = key & ((1ull << a.shift) - 1ull);
    if (a.nrm_in && out_nrm) out_nrm[idx] = make_float4((float)(nx / c), (float)(ny / c), (float)(nz / c), 0.0f);
}
__global__ void __launch_bounds__(BS) k_voxel_keys_multi(VoxArgs a) { d_voxel_keys_multi(a); }
__global__ void __launch_bounds__(BS) k_voxel_mean_multi(VoxArgs a) { d_voxel_mean_multi(a); }
__global__ void __launch_bounds__(BS) k_voxel_keys_multi_g(const VoxArgs *a) { d_voxel_keys_multi(a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_voxel_mean_multi_g(const VoxArgs *a) { d_voxel_mean_multi(a[blockIdx.y]); }

// the voxel stage of `count` clouds x n_scales grids in ONE keys / sort / scan / mean pass each (blockIdx.y = cloud): count == 1 is
// the per-cloud call.  outs: count x n_scales clouds (pts / keys / n / optional nrm allocated by the caller, cap >= n).  *done = false
// (and nothing enqueued) when the scales of some cloud cannot share one sort key or the scratch does not fit.
static int voxel_multi_batch(pcr_context *ctx, int count, const float *const *xyz, const float *const *nrm_in, const int64_t *n, const double *b6 /* count x 6 */, const double *voxels,
                             int n_scales, DevCloud *outs /* count x n_scales */, bool *done) {
    *done = false;
    if (n_scales < 2 || n_scales > VOX_MAX_SCALES || count < 1) return PCR_OK;
    std::vector<VoxArgs> va((size_t)count); std::vector<ScanArgs> sa((size_t)count);
    std::vector<void *> temps((size_t)count); std::vector<const uint64_t *> kin((size_t)count); std::vector<uint64_t *> kout((size_t)count);
    std::vector<const uint32_t *> vin((size_t)count); std::vector<uint32_t *> vout((size_t)count); std::vector<size_t> ne((size_t)count);
    int end_bit = 1; size_t max_ne = 0;
    for (int c = 0; c < count; c++) {
        if (n[c] <= 0 || n[c] * n_scales > 0x7fffffff / 4) return PCR_OK;
        VoxArgs &a = va[c]; std::memset(&a, 0, sizeof a);
        int shift = 0;
        for (int s = 0; s < n_scales; s++) {
            const double voxel = voxels[s];
            if (!(voxel > 0.0)) { ctx->err = "voxel_size <= 0"; return PCR_EINVAL; }
            a.g.ox[s] = b6[6 * c] - voxel * 0.5; a.g.oy[s] = b6[6 * c + 1] - voxel * 0.5; a.g.oz[s] = b6[6 * c + 2] - voxel * 0.5; a.g.voxel[s] = voxel;
            uint32_t mx = 0;
            for (int d = 0; d < 3; d++) {
                double e = floor((b6[6 * c + 3 + d] - (b6[6 * c + d] - voxel * 0.5)) / voxel);
                if (!(e < 2097152.0)) { ctx->err = "voxel_size is too small"; return PCR_EINVAL; }
                if ((uint32_t)e > mx) mx = (uint32_t)e;
            }
            if (3 * bits_for(mx) > shift) shift = 3 * bits_for(mx);
            DevCloud &o = outs[c * n_scales + s];
            o.key_org[0] = (float)a.g.ox[s]; o.key_org[1] = (float)a.g.oy[s]; o.key_org[2] = (float)a.g.oz[s];
            o.key_unit[0] = o.key_unit[1] = o.key_unit[2] = (float)voxel; o.voxel_lattice = true;
            a.o.pts[s] = o.pts; a.o.nrm[s] = o.nrm; a.o.keys[s] = o.keys; a.o.n[s] = o.n;
        }
        if (shift < 1) shift = 1;
        const int eb = shift + bits_for((uint32_t)(n_scales - 1));
        if (eb > 64) return PCR_OK;                    // the scale index does not fit above the Morton bits
        end_bit = eb > end_bit ? eb : end_bit;
        a.xyz = xyz[c]; a.nrm_in = nrm_in ? nrm_in[c] : nullptr; a.n = (int)n[c]; a.n_scales = n_scales; a.shift = shift;
        ne[c] = (size_t)n[c] * n_scales; max_ne = ne[c] > max_ne ? ne[c] : max_ne;
    }
    for (int c = 0; c < count; c++) {
        VoxArgs &a = va[c];
        uint64_t *k0 = arena<uint64_t>(ctx, ne[c]), *k1 = arena<uint64_t>(ctx, ne[c]);
        uint32_t *v0 = arena<uint32_t>(ctx, ne[c]), *v1 = arena<uint32_t>(ctx, ne[c]);
        uint8_t *flags = arena<uint8_t>(ctx, ne[c]);
        int *pos = arena<int>(ctx, ne[c]), *total = arena<int>(ctx, 1);
        const size_t tb = pcr_sort_temp_bytes(ne[c]);
        void *temp = pcr_arena_alloc(ctx, tb);
        if (!k0 || !k1 || !v0 || !v1 || !flags || !pos || !total || !temp) return PCR_OK;      // not enough scratch in this block: one by one
        a.keys_raw = k0; a.vals_raw = v0; a.keys = k1; a.vals = v1; a.flags = flags; a.pos = pos; a.total = total;
        temps[c] = temp; kin[c] = k0; kout[c] = k1; vin[c] = v0; vout[c] = v1;
        PCR_TRY(scan_args(ctx, &sa[c], flags, nullptr, (int)ne[c], pos, total, FlagSrc{k1, nullptr, nullptr}));
    }
    const int nb = (int)((max_ne + BS - 1) / BS);
    if (count == 1) {
        PCR_LAUNCH(ctx, k_voxel_keys_multi, dim3(nb), dim3(BS), 0, ctx->stream, va[0]);
        PCR_TRY(pcr_sort_pairs(ctx, temps[0], pcr_sort_temp_bytes(ne[0]), kin[0], kout[0], vin[0], vout[0], ne[0], end_bit));
        PCR_TRY(flag_scan_batch(ctx, sa.data(), 1));
        PCR_LAUNCH(ctx, k_voxel_mean_multi, dim3(nb), dim3(BS), 0, ctx->stream, va[0]);
    } else {
        const VoxArgs *dv = pcr_desc_upload(ctx, va.data(), count);
        if (!dv) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_voxel_keys_multi_g, dim3(nb, count), dim3(BS), 0, ctx->stream, dv);
        PCR_TRY(pcr_sort_pairs_batch(ctx, count, temps.data(), kin.data(), kout.data(), vin.data(), vout.data(), ne.data(), end_bit));
        PCR_TRY(flag_scan_batch(ctx, sa.data(), count));
        PCR_LAUNCH(ctx, k_voxel_mean_multi_g, dim3(nb, count), dim3(BS), 0, ctx->stream, dv);
    }
    *done = true;
    return PCR_OK;
}
int pcr_dev_voxel_multi(pcr_context *ctx, const float *xyz, const float *nrm_in, int64_t n, const double *b6, const double *voxels, int n_scales, DevCloud *outs, bool *done) {
    ArenaMark mark(ctx);
    return voxel_multi_batch(ctx, 1, &xyz, nrm_in ? &nrm_in : nullptr, &n, b6, voxels, n_scales, outs, done);
}
// the clouds of a group of pairs: scratch stays allocated above the caller's mark until the caller releases it
int pcr_dev_voxel_multi_batch(pcr_context *ctx, int count, const float *const *xyz, const float *const *nrm_in, const int64_t *n, const double *b6, const double *voxels,
                              int n_scales, DevCloud *outs, bool *done) {
    return voxel_multi_batch(ctx, count, xyz, nrm_in, n, b6, voxels, n_scales, outs, done);
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
    PCR_LAUNCH(ctx, k_gather_f3_to_f4, dim3((unsigned)((n + BS - 1) / BS)), dim3(BS), 0, ctx->stream, src, perm, (int)n, dst);
    return PCR_OK;
}
int pcr_dev_unpack_f3_to_f4(pcr_context *ctx, const float *src, int64_t n, float4 *dst) { return pcr_dev_gather_f3_to_f4(ctx, src, nullptr, n, dst); }
int pcr_dev_scatter_rows_f4_to_f3(pcr_context *ctx, const float4 *src, const uint32_t *perm, const int *n, int cap, float *dst) {
    if (cap <= 0) return PCR_OK;
    PCR_LAUNCH(ctx, k_scatter_f4_to_f3, dim3((cap + BS - 1) / BS), dim3(BS), 0, ctx->stream, src, perm, n, cap, dst);
    return PCR_OK;
}
int pcr_dev_pack_f4_to_f3(pcr_context *ctx, const float4 *src, const int *n, int cap, float *dst) { return pcr_dev_scatter_rows_f4_to_f3(ctx, src, nullptr, n, cap, dst); }

int pcr_dev_sort_cloud(pcr_context *ctx, const float *xyz, int64_t n, const double *b6, DevCloud *out, uint32_t *perm) {
    if (n > 0x7fffffff / 4) { ctx->err = "cloud too large"; return PCR_EINVAL; }
    PCR_LAUNCH(ctx, k_set_int, dim3(1), dim3(1), 0, ctx->stream, out->n, (int)n);
    if (n == 0) return PCR_OK;
    ArenaMark mark(ctx);
    uint64_t *k0 = arena<uint64_t>(ctx, n), *k1 = out->keys;
    uint32_t *v0 = arena<uint32_t>(ctx, n);
    const size_t tb = pcr_sort_temp_bytes(n);
    void *temp = pcr_arena_alloc(ctx, tb);
    if (!k0 || !k1 || !v0 || !temp) return PCR_ENOMEM;
    // ONE lattice unit for the three axes (the largest extent over 2^16 cells): cubic Morton cells.  A unit per axis (round 1-2) made
    // the cells of an NCLT-shaped cloud (235 x 235 x 12 m) 20 x flatter than wide, and the curve then cuts a ground patch into contour
    // strips: 64 consecutive points had 408 distinct 30-NN neighbours instead of 209 (tools/knnw_stats.py).
    float s[3];
    double emax = 0.0;
    for (int d = 0; d < 3; d++) emax = b6[3 + d] - b6[d] > emax ? b6[3 + d] - b6[d] : emax;
    for (int d = 0; d < 3; d++) {
        const double e = emax;
        s[d] = e > 0 ? (float)(65535.0 / e) : 0.0f;
        out->key_org[d] = (float)b6[d]; out->key_unit[d] = e > 0 ? (float)(e / 65535.0) : 1.0f;
    }
    const int nb = (int)((n + BS - 1) / BS);
    PCR_LAUNCH(ctx, k_raw_keys, dim3(nb), dim3(BS), 0, ctx->stream, xyz, (int)n, (float)b6[0], (float)b6[1], (float)b6[2], s[0], s[1], s[2], k0, v0);
    PCR_TRY(pcr_sort_pairs(ctx, temp, tb, k0, k1, v0, perm, n, 48));
    PCR_LAUNCH(ctx, k_gather_f3_to_f4, dim3(nb), dim3(BS), 0, ctx->stream, xyz, perm, (int)n, out->pts);
    return PCR_OK;
}

// ---- the same for `count` clouds in ONE pass (lockstep FGR groups): bounds (one read-back), keys, the batched radix sort, gathers and
// all octrees in one batch of the six build kernels.  Per cloud the arithmetic of pcr_dev_sort_cloud: same lattice, same keys, same
// stable order.  cs[c] must be allocated (pcr_alloc_cloud with a tree); perms[c]: sorted -> caller index.
struct RawKeyArgs { const float *xyz; const float *nrm; int n; float ox, oy, oz, s; uint64_t *keys; uint32_t *vals; const uint32_t *perm; float4 *pts; float4 *nrm_out; int *n_dev; };
__global__ void __launch_bounds__(BS) k_raw_keys_g(const RawKeyArgs *a_) {
    const RawKeyArgs &a = a_[blockIdx.y];
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i == 0) *a.n_dev = a.n;
    if (i >= a.n) return;
    const uint32_t ix = (uint32_t)fminf((a.xyz[i * 3] - a.ox) * a.s, 65535.0f);
    const uint32_t iy = (uint32_t)fminf((a.xyz[i * 3 + 1] - a.oy) * a.s, 65535.0f);
    const uint32_t iz = (uint32_t)fminf((a.xyz[i * 3 + 2] - a.oz) * a.s, 65535.0f);
    a.keys[i] = pcr_morton3(ix, iy, iz);
    a.vals[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(BS) k_gather_sorted_g(const RawKeyArgs *a_) {
    const RawKeyArgs &a = a_[blockIdx.y];
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t v = a.perm[i];
    a.pts[i] = make_float4(a.xyz[v * 3], a.xyz[v * 3 + 1], a.xyz[v * 3 + 2], 0.0f);
    if (a.nrm) a.nrm_out[i] = make_float4(a.nrm[v * 3], a.nrm[v * 3 + 1], a.nrm[v * 3 + 2], 0.0f);
}
int pcr_import_clouds_batch(pcr_context *ctx, int count, const float *const *xyz, const float *const *nrm, const int64_t *n, DevCloud *cs, uint32_t **perms) {
    if (count < 1) return PCR_OK;
    std::vector<double> b6((size_t)count * 6);
    PCR_TRY(pcr_dev_bounds_batch(ctx, count, xyz, n, b6.data()));
    std::vector<RawKeyArgs> a((size_t)count);
    std::vector<void *> temps((size_t)count); std::vector<const uint64_t *> kin((size_t)count); std::vector<uint64_t *> kout((size_t)count);
    std::vector<const uint32_t *> vin((size_t)count); std::vector<uint32_t *> vout((size_t)count); std::vector<size_t> nn((size_t)count);
    int max_nb = 1;
    for (int c = 0; c < count; c++) {
        if (n[c] > 0x7fffffff / 4) { ctx->err = "cloud too large"; return PCR_EINVAL; }
        DevCloud *out = &cs[c];
        perms[c] = arena<uint32_t>(ctx, n[c]);
        uint64_t *k0 = arena<uint64_t>(ctx, n[c]); uint32_t *v0 = arena<uint32_t>(ctx, n[c]);
        temps[c] = pcr_arena_alloc(ctx, pcr_sort_temp_bytes(n[c]));
        if (!perms[c] || !k0 || !v0 || !temps[c]) return PCR_ENOMEM;
        double emax = 0.0;
        for (int d = 0; d < 3; d++) emax = b6[6 * c + 3 + d] - b6[6 * c + d] > emax ? b6[6 * c + 3 + d] - b6[6 * c + d] : emax;
        for (int d = 0; d < 3; d++) { out->key_org[d] = (float)b6[6 * c + d]; out->key_unit[d] = emax > 0 ? (float)(emax / 65535.0) : 1.0f; }
        RawKeyArgs &r = a[c];
        r.xyz = xyz[c]; r.nrm = nrm ? nrm[c] : nullptr; r.n = (int)n[c]; r.ox = (float)b6[6 * c]; r.oy = (float)b6[6 * c + 1]; r.oz = (float)b6[6 * c + 2];
        r.s = emax > 0 ? (float)(65535.0 / emax) : 0.0f; r.keys = k0; r.vals = v0; r.perm = perms[c]; r.pts = out->pts; r.nrm_out = out->nrm; r.n_dev = out->n;
        if (r.nrm && !r.nrm_out) { ctx->err = "import batch: cloud allocated without normals"; return PCR_EINVAL; }
        kin[c] = k0; kout[c] = out->keys; vin[c] = v0; vout[c] = perms[c]; nn[c] = (size_t)n[c];
        const int nb = (int)((n[c] + BS - 1) / BS);
        max_nb = nb > max_nb ? nb : max_nb;
    }
    const RawKeyArgs *d = pcr_desc_upload(ctx, a.data(), count);
    if (!d) return PCR_ENOMEM;
    PCR_LAUNCH(ctx, k_raw_keys_g, dim3(max_nb, count), dim3(BS), 0, ctx->stream, d);
    PCR_TRY(pcr_sort_pairs_batch(ctx, count, temps.data(), kin.data(), kout.data(), vin.data(), vout.data(), nn.data(), 48));
    PCR_LAUNCH(ctx, k_gather_sorted_g, dim3(max_nb, count), dim3(BS), 0, ctx->stream, d);
    std::vector<DevCloud *> cp((size_t)count);
    for (int c = 0; c < count; c++) cp[c] = &cs[c];
    return pcr_dev_build_bvh_batch(ctx, cp.data(), count);
}

// ====================================================================== linear octree build (K2, part 2)
// A: level of the highest Morton bit in which consecutive keys differ (-1: identical keys) + histogram
#define OCT_ROW 24      // ints per tile row: cumulative node-start counts of the 22 key levels (+ padding)
__device__ static inline void d_oct_lstar(const uint64_t *__restrict__ keys, const int *__restrict__ n_ptr, signed char *__restrict__ ls, int *__restrict__ rows) {
    __shared__ int h[OCT_KEY_LEVELS];
    if (threadIdx.x < OCT_KEY_LEVELS) h[threadIdx.x] = 0;
    __syncthreads();
    const int n = *n_ptr;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = blockIdx.x * TILE + j * BS + threadIdx.x;
        if (i < n) {
            int v;
            if (i == 0) v = OCT_KEY_LEVELS - 1;
            else {
                const uint64_t x = keys[i] ^ keys[i - 1];
                v = x ? (63 - __builtin_clzll(x)) / 3 : -1;
            }
            ls[i] = (signed char)v;
            if (v >= 0) atomicAdd(&h[v], 1);
        }
    }
    __syncthreads();
    // row[l] = number of elements of this tile that start a node of level l (= differ from their predecessor at level >= l):
    // no global histogram, no memset, and the same rows are the per-tile counts the node numbering needs
    if (threadIdx.x < OCT_ROW) {
        int c = 0;
        for (int l = threadIdx.x; l < OCT_KEY_LEVELS; l++) c += h[l];
        rows[blockIdx.x * OCT_ROW + threadIdx.x] = threadIdx.x < OCT_KEY_LEVELS ? c : 0;
    }
}
// B: choose the leaf level (first level with <= n/leaf_div cells; levels are 8x apart, so 6 means 6-48 points per leaf), the root level, offsets
struct OctGeom { float org[3]; float unit[3]; int leaf_div; };
__device__ static inline void d_oct_meta(const int *__restrict__ n_ptr, const int *__restrict__ rows, int n_tiles, int node_cap, OctMeta *__restrict__ meta, int *__restrict__ child, OctGeom g) {
    __shared__ int part[8][32];
    __shared__ int tot[OCT_ROW];
    {
        const int l = threadIdx.x & 31, tl = threadIdx.x >> 5;          // 8 tile lanes x 32 columns
        int c = 0;
        if (l < OCT_ROW) for (int t = tl; t < n_tiles; t += 8) c += rows[t * OCT_ROW + l];
        part[tl][l] = c;
    }
    __syncthreads();
    if (threadIdx.x < OCT_ROW) { int c = 0; for (int k = 0; k < 8; k++) c += part[k][threadIdx.x]; tot[threadIdx.x] = c; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int n = *n_ptr;
    OctMeta m;
    for (int d = 0; d < 3; d++) { m.org[d] = g.org[d]; m.unit[d] = g.unit[d]; }
    m.n = n; m.l0 = 0; m.nl = 0; m.total = 0;
    for (int k = 0; k < OCT_MAXL; k++) { m.cnt[k] = 0; m.off[k] = 0; }
    if (n > 0) {
        int cnt[OCT_KEY_LEVELS + 1];
        cnt[OCT_KEY_LEVELS] = 0;
        for (int l = OCT_KEY_LEVELS - 1; l >= 0; l--) cnt[l] = tot[l];
        const int want = n / g.leaf_div > 1 ? n / g.leaf_div : 1;
        int l0 = 0;
        while (l0 < OCT_KEY_LEVELS - 1 && cnt[l0] > want) l0++;
        int top = l0;
        while (top < OCT_KEY_LEVELS - 1 && cnt[top] > 1) top++;
        for (;;) {      // respect the level and node budgets by coarsening the leaves
            if (top - l0 + 1 > OCT_MAXL) { l0 = top - OCT_MAXL + 1; }
            long long tot = 0;
            for (int l = l0; l <= top; l++) tot += cnt[l] + 1;
            if (tot <= node_cap || l0 == top) break;
            l0++;
        }
        m.l0 = l0; m.nl = top - l0 + 1;
        int off = 0;
        for (int li = 0; li < m.nl; li++) { m.cnt[li] = cnt[l0 + li]; m.off[li] = off; off += m.cnt[li] + 1; }
        m.total = off;
        for (int li = 0; li < m.nl; li++) child[m.off[li] + m.cnt[li]] = li == 0 ? n : m.cnt[li - 1];
    }
    *meta = m;
}
// E: node ids by ballot ranking; write child links and the leaf of every point
__device__ static inline void d_oct_apply(const signed char *__restrict__ ls, const OctMeta *__restrict__ meta, const int *__restrict__ rows,
                                                  int *__restrict__ child, int *__restrict__ leaf_of) {
    __shared__ int wtot[4][BS / PCR_WAVE][OCT_MAXL];
    __shared__ int part[BS / OCT_MAXL][OCT_MAXL];
    __shared__ int pre_s[OCT_MAXL];
    const int n = meta->n, l0 = meta->l0, nl = meta->nl;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // exclusive prefix over the tiles before this one, per stored level (was a separate single-workgroup scan launch)
    {
        const int li = threadIdx.x & (OCT_MAXL - 1), tl = threadIdx.x / OCT_MAXL;
        int c = 0;
        if (li < nl) for (int t = tl; t < (int)blockIdx.x; t += BS / OCT_MAXL) c += rows[t * OCT_ROW + l0 + li];
        part[tl][li] = c;
        __syncthreads();
        if (threadIdx.x < OCT_MAXL) { int sum = 0; for (int k = 0; k < BS / OCT_MAXL; k++) sum += part[k][threadIdx.x]; pre_s[threadIdx.x] = sum; }
        __syncthreads();
    }
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int e = blockIdx.x * TILE + j * BS + threadIdx.x; v[j] = e < n ? (int)ls[e] : -2; }
    for (int li = 0; li < nl; li++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long b = __ballot(v[j] >= l0 + li);
            if (lane == 0) wtot[j][w][li] = __popcll(b);
        }
    __syncthreads();
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int e = blockIdx.x * TILE + j * BS + threadIdx.x;
        int below = -1;
        for (int li = 0; li < nl; li++) {
            const bool f = v[j] >= l0 + li;
            const unsigned long long b = __ballot(f);
            int pre = pre_s[li];
            for (int jj = 0; jj < 4; jj++)
                for (int ww = 0; ww < BS / PCR_WAVE; ww++)
                    if (jj < j || (jj == j && ww < w)) pre += wtot[jj][ww][li];
            const int id = pre + __popcll(b & lt);      // exclusive rank of this element among the starts of level li
            if (li == 0 && e < n) leaf_of[e] = id + (f ? 1 : 0) - 1;
            if (f) child[meta->off[li] + id] = li == 0 ? e : below;
            below = id;
        }
    }
}
// F/G: tight boxes, bottom-up.  ONE OCTET (8 lanes) PER NODE: lane c reads point c, c + 8, ... of a leaf (coalesced 128-byte rows) or
// child c of an inner node (one 256-byte read per octet); min / max over the octet by DPP.  (A thread per node walked its points or
// children serially and, at level 1, wrote the fat-leaf record of up to ~400 points in one loop: 95 + 78 us per build in flight.)
__device__ static inline float pcr_octet_minf(float v) {
    v = fminf(v, pcr_dpp_f<PCR_DPP_XOR1>(v)); v = fminf(v, pcr_dpp_f<PCR_DPP_XOR2>(v)); v = fminf(v, pcr_dpp_f<PCR_DPP_HMIRROR>(v));
    return v;
}
__device__ static inline float pcr_octet_maxf(float v) {
    v = fmaxf(v, pcr_dpp_f<PCR_DPP_XOR1>(v)); v = fmaxf(v, pcr_dpp_f<PCR_DPP_XOR2>(v)); v = fmaxf(v, pcr_dpp_f<PCR_DPP_HMIRROR>(v));
    return v;
}
__device__ static inline void d_oct_leaf_boxes(const float4 *__restrict__ pts, const OctMeta *__restrict__ meta, const int *__restrict__ child, float4 *__restrict__ boxes, int4 *__restrict__ up,
                                                       int4 *__restrict__ pinfo) {
    const int ol = threadIdx.x & 7;
    const int j = blockIdx.x * (BS / OCT) + (threadIdx.x >> 3);
    const bool live = meta->nl >= 1 && j < meta->cnt[0];
    if (__ballot(live) == 0ull) return;
    const int a = live ? child[j] : 0, b = live ? child[j + 1] : 0;
    const bool one_level = meta->nl == 1;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = a + ol; i < b; i += OCT) {
        const float4 p = pts[i];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
        if (one_level) pinfo[i] = make_int4(j, a, b - a, 0);     // one-level tree: the leaf is the start node
    }
#pragma unroll
    for (int d = 0; d < 3; d++) { lo[d] = pcr_octet_minf(lo[d]); hi[d] = pcr_octet_maxf(hi[d]); }
    if (live && ol == 0) {
        boxes[2 * (size_t)j] = make_float4(lo[0], lo[1], lo[2], __int_as_float(a));
        boxes[2 * (size_t)j + 1] = make_float4(hi[0], hi[1], hi[2], __int_as_float(b - a));
        if (one_level) up[j] = make_int4(0, 0, 1, 0);
    }
}
// node j of level li (li >= 1), all 64 lanes of the wavefront call it with their octet's j (live: octet-uniform)
__device__ static inline void oct_node_box(const OctMeta &m, const int *__restrict__ child, float4 *__restrict__ boxes, int4 *__restrict__ up, int li, int j, bool live, int ol,
                                           int4 *__restrict__ pinfo = nullptr, int2 *__restrict__ l1 = nullptr) {
    const int a = live ? child[m.off[li] + j] : 0, b = live ? child[m.off[li] + j + 1] : 0;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    int pf = 0x7fffffff, pl = 0;                                   // first point / one past the last point of the node (level 1 only)
    const int c = a + ol;
    if (c < b) {                                                   // <= 8 children: one per lane
        const float4 x = boxes[2 * (size_t)(m.off[li - 1] + c)], y = boxes[2 * (size_t)(m.off[li - 1] + c) + 1];
        lo[0] = x.x; lo[1] = x.y; lo[2] = x.z; hi[0] = y.x; hi[1] = y.y; hi[2] = y.z;
        up[m.off[li - 1] + c] = make_int4(j, a, b - a, 0);          // child -> (parent, first sibling, sibling count)
        if (li == 1 && pinfo) { pf = __float_as_int(x.w); pl = pf + __float_as_int(y.w); }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) { lo[d] = pcr_octet_minf(lo[d]); hi[d] = pcr_octet_maxf(hi[d]); }
    if (li == 1 && pinfo) {       // point -> (level-1 node, its first point, its point count, 1): the octet writes the node's points together
        pf = min(pf, pcr_dpp_i<PCR_DPP_XOR1>(pf)); pf = min(pf, pcr_dpp_i<PCR_DPP_XOR2>(pf)); pf = min(pf, pcr_dpp_i<PCR_DPP_HMIRROR>(pf));
        pl = max(pl, pcr_dpp_i<PCR_DPP_XOR1>(pl)); pl = max(pl, pcr_dpp_i<PCR_DPP_XOR2>(pl)); pl = max(pl, pcr_dpp_i<PCR_DPP_HMIRROR>(pl));
        if (live) for (int i = pf + ol; i < pl; i += OCT) pinfo[i] = make_int4(j, pf, pl - pf, 1);
        if (live && ol == 0 && l1) l1[j] = make_int2(pf, pl - pf);
    }
    if (live && ol == 0) {
        if (li == m.nl - 1) up[m.off[li] + j] = make_int4(0, 0, 1, 0);
        boxes[2 * (size_t)(m.off[li] + j)] = make_float4(lo[0], lo[1], lo[2], __int_as_float(a));
        boxes[2 * (size_t)(m.off[li] + j) + 1] = make_float4(hi[0], hi[1], hi[2], __int_as_float(b - a));
    }
}
__device__ static inline void d_oct_level_boxes(const OctMeta *__restrict__ meta, const int *__restrict__ child, float4 *__restrict__ boxes, int4 *__restrict__ up, int li,
                                                        int4 *__restrict__ pinfo, int2 *__restrict__ l1) {
    const int j = blockIdx.x * (BS / OCT) + (threadIdx.x >> 3);
    const bool live = li < meta->nl && j < meta->cnt[li];
    if (__ballot(live) == 0ull) return;
    oct_node_box(*meta, child, boxes, up, li, j, live, threadIdx.x & 7, pinfo, l1);
}
// remaining (small) levels in ONE workgroup per tree, level by level.  OCT_UPPER_BS threads = that / 8 octets per round.  (Round 5: 256 threads instead of
// 1024.  A 1024-thread workgroup needs a CU with sixteen free wavefront slots at once; next to the other groups' 9-ms k-NN launches it waited for one:
// 0.9 ms on average and up to 11 ms for ~30 us of work, 5.3 % of the headline run's kernel time and every group's chain held up twice.)
#define OCT_UPPER_BS 256
__device__ static inline void d_oct_upper_boxes(const OctMeta *__restrict__ meta, const int *__restrict__ child, float4 *__restrict__ boxes, int4 *__restrict__ up, int first_li) {
    __shared__ OctMeta m;
    if (threadIdx.x == 0) m = *meta;
    __syncthreads();
    constexpr int OPR = OCT_UPPER_BS / OCT;
    for (int li = first_li; li < m.nl; li++) {
        const int rounds = (m.cnt[li] + OPR - 1) / OPR;
        for (int r = 0; r < rounds; r++) {
            const int j = r * OPR + (threadIdx.x >> 3);
            oct_node_box(m, child, boxes, up, li, j, j < m.cnt[li], threadIdx.x & 7);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- the six build kernels serve up to OCT_BATCH trees per launch: blockIdx.y picks the tree (the voxel clouds of all scales
// of a registration are built together: 6 launches instead of 6 per scale)
#define OCT_BATCH PCR_MAX_BATCH
struct OctBuildDesc {
    const uint64_t *keys; const int *n; signed char *ls; int *rows; OctMeta *meta; int *child; int *leaf_of;
    const float4 *pts; float4 *nodes; int4 *up; int4 *pinfo; int2 *l1; OctGeom g; int n_tiles, node_cap;
};
struct OctBuildBatch { OctBuildDesc a[OCT_BATCH]; };
__device__ static inline void dd_oct_lstar(const OctBuildDesc &d) { if ((int)blockIdx.x >= d.n_tiles) return; d_oct_lstar(d.keys, d.n, d.ls, d.rows); }
__device__ static inline void dd_oct_meta(const OctBuildDesc &d) { d_oct_meta(d.n, d.rows, d.n_tiles, d.node_cap, d.meta, d.child, d.g); }
__device__ static inline void dd_oct_apply(const OctBuildDesc &d) { if ((int)blockIdx.x >= d.n_tiles) return; d_oct_apply(d.ls, d.meta, d.rows, d.child, d.leaf_of); }
__device__ static inline void dd_oct_leaf(const OctBuildDesc &d) { d_oct_leaf_boxes(d.pts, d.meta, d.child, d.nodes, d.up, d.pinfo); }
__device__ static inline void dd_oct_level1(const OctBuildDesc &d) { d_oct_level_boxes(d.meta, d.child, d.nodes, d.up, 1, d.pinfo, d.l1); }
__device__ static inline void dd_oct_upper(const OctBuildDesc &d) { d_oct_upper_boxes(d.meta, d.child, d.nodes, d.up, 2); }
__global__ void __launch_bounds__(BS) k_oct_lstar(OctBuildBatch b) { dd_oct_lstar(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(256) k_oct_meta(OctBuildBatch b) { dd_oct_meta(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_oct_apply(OctBuildBatch b) { dd_oct_apply(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_oct_leaf_boxes(OctBuildBatch b) { dd_oct_leaf(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_oct_level_boxes(OctBuildBatch b) { dd_oct_level1(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(OCT_UPPER_BS) k_oct_upper_boxes(OctBuildBatch b) { dd_oct_upper(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_oct_lstar_p(const OctBuildDesc *a) { dd_oct_lstar(a[blockIdx.y]); }
__global__ void __launch_bounds__(256) k_oct_meta_p(const OctBuildDesc *a) { dd_oct_meta(a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_oct_apply_p(const OctBuildDesc *a) { dd_oct_apply(a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_oct_leaf_boxes_p(const OctBuildDesc *a) { dd_oct_leaf(a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_oct_level_boxes_p(const OctBuildDesc *a) { dd_oct_level1(a[blockIdx.y]); }
__global__ void __launch_bounds__(OCT_UPPER_BS) k_oct_upper_boxes_p(const OctBuildDesc *a) { dd_oct_upper(a[blockIdx.y]); }

int pcr_dev_build_bvh_batch(pcr_context *ctx, DevCloud *const *cs, int count) {
    if (count < 1) return PCR_OK;
    if (count > PCR_MAX_GROUP_BATCH) { ctx->err = "too many trees in one batch"; return PCR_EINVAL; }
    ArenaMark mark(ctx);
    std::vector<OctBuildDesc> b((size_t)count);
    std::memset(b.data(), 0, sizeof(OctBuildDesc) * (size_t)count);
    static const int div = getenv("PCR_OCT_DIV") ? atoi(getenv("PCR_OCT_DIV")) : 6;
    int m = 0, max_tiles = 0, max_nbl = 0;
    for (int k = 0; k < count; k++) {
        DevCloud *c = cs[k];
        if (c->cap <= 0) continue;
        OctBuildDesc &d = b[m++];
        d.n_tiles = (c->cap + TILE - 1) / TILE; d.node_cap = (int)oct_node_capacity(c->cap);
        d.ls = arena<signed char>(ctx, c->cap); d.rows = arena<int>(ctx, (size_t)d.n_tiles * OCT_ROW);
        if (!d.ls || !d.rows) return PCR_ENOMEM;
        d.keys = c->keys; d.n = c->n; d.meta = c->oct_meta; d.child = c->oct_child; d.leaf_of = c->leaf_of;
        d.pts = c->pts; d.nodes = c->oct_nodes; d.up = c->oct_up; d.pinfo = c->pinfo; d.l1 = c->oct_l1;
        for (int a = 0; a < 3; a++) { d.g.org[a] = c->key_org[a]; d.g.unit[a] = c->key_unit[a]; }
        d.g.leaf_div = div;
        const int nbl = (int)((((size_t)c->cap / 2 + 1) * OCT + BS - 1) / BS);       // <= n/2 leaves (or 1), one octet each
        if (d.n_tiles > max_tiles) max_tiles = d.n_tiles;
        if (nbl > max_nbl) max_nbl = nbl;
    }
    if (m == 0) return PCR_OK;
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, OctBuildBatch, k_oct_lstar, k_oct_lstar_p, b.data(), m, dim3(max_tiles, m), dim3(BS)));
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, OctBuildBatch, k_oct_meta, k_oct_meta_p, b.data(), m, dim3(1, m), dim3(256)));
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, OctBuildBatch, k_oct_apply, k_oct_apply_p, b.data(), m, dim3(max_tiles, m), dim3(BS)));
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, OctBuildBatch, k_oct_leaf_boxes, k_oct_leaf_boxes_p, b.data(), m, dim3(max_nbl, m), dim3(BS)));
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, OctBuildBatch, k_oct_level_boxes, k_oct_level_boxes_p, b.data(), m, dim3(max_nbl, m), dim3(BS)));
    // levels >= 2 (n/64 nodes and fewer) in ONE workgroup per tree, level by level: a launch less than one grid per level
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, OctBuildBatch, k_oct_upper_boxes, k_oct_upper_boxes_p, b.data(), m, dim3(1, m), dim3(OCT_UPPER_BS)));
    return PCR_OK;
}
int pcr_dev_build_bvh(pcr_context *ctx, DevCloud *c) { return pcr_dev_build_bvh_batch(ctx, &c, 1); }

// ====================================================================== cell hash of a voxel-lattice cloud (pcr_octree.h GridView)
// One thread per point: the first point of a level-L Morton cell counts the cell's points (<= 8^L: the lattice holds one point per
// voxel) and inserts (code -> first, count) by linear probing.  blockIdx.y picks the cloud.
struct GridBuildDesc { const uint64_t *keys; const int *n; GridEntry *tab; unsigned *dmask; unsigned max_slots; int shift; };
// the table of a cloud is as large as its point COUNT asks for (pcr_grid_slots), inside the allocation its capacity asked for: cleared here, mask left for the searches
__global__ void __launch_bounds__(BS) k_grid_clear(const GridBuildDesc *descs) {
    const GridBuildDesc d = descs[blockIdx.y];
    const int n = *d.n;
    unsigned slots = pcr_grid_slots((unsigned)(n > 0 ? n : 0)); slots = slots < d.max_slots ? slots : d.max_slots;
    if (blockIdx.x == 0 && threadIdx.x == 0) *d.dmask = slots - 1;
    const unsigned long long ff = 0xffffffffffffffffull;
    ulonglong2 *t2 = (ulonglong2 *)d.tab;
    for (unsigned i = blockIdx.x * BS + threadIdx.x; i < slots; i += gridDim.x * BS) t2[i] = make_ulonglong2(ff, ff);
}
__global__ void __launch_bounds__(BS) k_grid_build(const GridBuildDesc *descs) {
    const GridBuildDesc dd = descs[blockIdx.y];
    const int n = *dd.n, i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    unsigned slots = pcr_grid_slots((unsigned)n); slots = slots < dd.max_slots ? slots : dd.max_slots;
    struct { const uint64_t *keys; GridEntry *tab; unsigned mask; int shift; } d = {dd.keys, dd.tab, slots - 1, dd.shift};
    const unsigned long long code = d.keys[i] >> d.shift;
    if (i > 0 && (d.keys[i - 1] >> d.shift) == code) return;
    int j = i + 1;
    while (j < n && (d.keys[j] >> d.shift) == code) j++;
    unsigned h = pcr_grid_hash(code, d.mask);
    for (;;) {
        const unsigned long long old = atomicCAS(&d.tab[h].code, PCR_GRID_EMPTY, code);
        if (old == PCR_GRID_EMPTY) { d.tab[h].first = i; d.tab[h].count = j - i; return; }
        h = (h + 1) & d.mask;                              // (a cell has one first point: `old == code` cannot happen)
    }
}
// level whose cell edge covers twice the radius with 2 % to spare (rounding of the query's cell against the points' exact voxel indices), or -1:
// only for voxel-lattice clouds
int pcr_grid_level_for(const DevCloud *c, double search_radius) {
    if (!c->voxel_lattice || !(search_radius > 0.0) || !(c->key_unit[0] > 0.0f)) return -1;
    // grid_nn_query8: cell edge >= 2.04 r (the ball meets only the 2 x 2 x 2 block of cells on q's side), cells of at most 8 voxels
    // (<= 512 points: the packed range of a cell holds 10 bits of count and 22 of index)
    if (c->cap >= (1 << 22)) return -1;
    int L = 0;
    while (L <= 3 && (double)c->key_unit[0] * (double)(1 << L) * 0.49 < search_radius) L++;
    return L <= 3 ? L : -1;
}
int pcr_dev_build_grid_batch(pcr_context *ctx, const DevCloud *const *cs, const int *levels, int count, GridView *views) {
    std::vector<GridBuildDesc> d; int max_cap = 0;
    size_t total_slots = 0;
    auto slots_of = [](const DevCloud *c) { return pcr_grid_slots((unsigned)c->cap); };      // the allocation: what the capacity could need (load factor <= 0.5)
    for (int k = 0; k < count; k++) {
        std::memset(&views[k], 0, sizeof(GridView)); views[k].L = -1;
        if (levels[k] < 0 || cs[k]->cap <= 0 || !cs[k]->voxel_lattice) continue;
        total_slots += slots_of(cs[k]);
    }
    if (total_slots == 0) return PCR_OK;
    GridEntry *tab = arena<GridEntry>(ctx, total_slots);
    unsigned *dmasks = arena<unsigned>(ctx, (size_t)count);
    if (!tab || !dmasks) return PCR_ENOMEM;
    size_t off = 0;
    for (int k = 0; k < count; k++) {
        const DevCloud *c = cs[k];
        if (levels[k] < 0 || c->cap <= 0 || !c->voxel_lattice) continue;
        const unsigned slots = slots_of(c);
        GridView &g = views[k];
        g.tab = tab + off; g.dmask = dmasks + k; g.L = levels[k];
        for (int a = 0; a < 3; a++) { g.org[a] = c->key_org[a]; g.inv_unit[a] = 1.0f / c->key_unit[a]; g.cell[a] = c->key_unit[a] * (float)(1 << levels[k]); }
        d.push_back(GridBuildDesc{c->keys, c->n, tab + off, dmasks + k, slots, 3 * levels[k]});
        off += slots; max_cap = c->cap > max_cap ? c->cap : max_cap;
    }
    const GridBuildDesc *dd = pcr_desc_upload(ctx, d.data(), (int)d.size());
    if (!dd) return PCR_ENOMEM;
    {   // clear: a grid-stride launch over what the COUNT needs (a quarter of a 2M-capacity table for a 460k-point scale)
        const unsigned gx = (unsigned)((2 * (size_t)max_cap + BS - 1) / BS);
        PCR_LAUNCH(ctx, k_grid_clear, dim3(gx < 2048u ? gx : 2048u, (unsigned)d.size()), dim3(BS), 0, ctx->stream, dd);
    }
    PCR_LAUNCH(ctx, k_grid_build, dim3((unsigned)((max_cap + BS - 1) / BS), (unsigned)d.size()), dim3(BS), 0, ctx->stream, dd);
    return PCR_OK;
}

static inline OctView oct_view(const DevCloud *c) {
    OctView v; v.pts = c->pts; v.nodes = c->oct_nodes; v.up = c->oct_up; v.meta = c->oct_meta; v.leaf_of = c->leaf_of; v.keys = c->keys; v.pinfo = c->pinfo; v.l1rng = c->oct_l1;
    return v;
}

// ====================================================================================== k-NN (K3)
// "Octet" k-NN: 8 consecutive lanes cooperate on ONE query, a wavefront serves 8 Morton-consecutive queries.
//   * node visit : lane c tests child c's box (one coalesced 256-B read per octet), the 8 verdicts come back
//                  as one byte of a wave ballot -> the pending-children mask of the stack-free walk;
//   * leaf visit : lane c tests point c of the leaf (one coalesced 128-B read), survivors are inserted one per
//                  round into the octet's k-best, which is DISTRIBUTED over the 8 lanes (slot s lives in lane s%8,
//                  register s/8); the current worst is an 8-lane arg-max (3 xor-shuffles);
//   * no LDS, no per-lane divergence inside a query, ~50 VGPRs -> full occupancy.
// The walk is seeded with the query's Morton neighbours (own leaf +- span), which already contain most of the
// true neighbours, so the bound is tight before the first box test.
#define KNN_BS 256

template <int SLOTS>
struct OctetKnn {
    // the octet's k-best: slot s lives in lane s % 8, register s / 8.  Every lane keeps its SLOTS registers sorted in DESCENDING order
    // (disabled slots, -1, at the end), so its largest entry is sd[0], the octet's worst is an 8-lane maximum of sd[0], and an insertion is
    // "replace sd[0] of the owning lane and re-sort", which for one new value is a chain of v_med3: sorted {x, s1, .., s_{S-1}} has
    // max(x, s1), med3(x, s_j, s_{j+1}) ..., min(x, s_{S-1}).  15 VALU instructions for 4 slots (the replace-by-register-index form with
    // its local arg-max took 21), 99 instead of 147 for the 25 slots of the 200-NN lists.
    float sd[SLOTS]; int si[SLOTS];
    float worst; int wlane;      // octet-wide worst distance and the octet-lane that owns it
    int ol;                      // lane within the octet
    __device__ void init(int k, float cap, int ol_) {
        ol = ol_;
#pragma unroll
        for (int j = 0; j < SLOTS; j++) { sd[j] = (ol + OCT * j < k) ? cap : -1.0f; si[j] = -1; }
        refresh();
    }
    __device__ void refresh() {
        worst = pcr_octet_max(sd[0]);
        // owner = lowest lane of the octet whose largest entry is the octet maximum (DPP + one ballot, no LDS hop)
        const unsigned long long own = __ballot(sd[0] == worst);
        wlane = __builtin_ctz((uint32_t)(own >> ((threadIdx.x & 56))) & 0xffu);
    }
    // after slots were written directly: odd-even transposition sort of the lane's registers (descending), then the octet state
    __device__ void sort_lane() {
#pragma unroll
        for (int pass = 0; pass < SLOTS; pass++)
#pragma unroll
            for (int j = pass & 1; j + 1 < SLOTS; j += 2) {
                const bool sw = sd[j] < sd[j + 1];
                const float a = sd[j], b = sd[j + 1]; const int ia = si[j], ib = si[j + 1];
                sd[j] = sw ? b : a; sd[j + 1] = sw ? a : b; si[j] = sw ? ib : ia; si[j + 1] = sw ? ia : ib;
            }
        refresh();
    }
    // all 8 lanes call with the same candidate; branch-free (a candidate that no longer beats the bound changes nothing)
    __device__ void insert(float cd, int ci) {
        const bool own = cd < worst && ol == wlane;
        const float x = own ? cd : sd[0];
        const int ix = own ? ci : si[0];
        if (SLOTS == 1) { sd[0] = x; si[0] = ix; refresh(); return; }
        bool c[SLOTS];
#pragma unroll
        for (int j = 1; j < SLOTS; j++) c[j] = x >= sd[j];
        float nd[SLOTS]; int ni[SLOTS];
        // entries are non-negative floats or the -1 sentinel: they order like their bit patterns as signed integers (no NaN canonicalisation)
        nd[0] = __int_as_float(max(__float_as_int(x), __float_as_int(sd[1]))); ni[0] = c[1] ? ix : si[1];
#pragma unroll
        for (int j = 1; j + 1 < SLOTS; j++) { nd[j] = __builtin_amdgcn_fmed3f(x, sd[j], sd[j + 1]); ni[j] = c[j] ? si[j] : (c[j + 1] ? ix : si[j + 1]); }
        nd[SLOTS - 1] = __int_as_float(min(__float_as_int(x), __float_as_int(sd[SLOTS - 1]))); ni[SLOTS - 1] = c[SLOTS - 1] ? si[SLOTS - 1] : ix;
#pragma unroll
        for (int j = 0; j < SLOTS; j++) { sd[j] = nd[j]; si[j] = ni[j]; }
        refresh();
    }
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
    OctView t; const int *n_ptr;
    int k; float r2cap_f; double r2cap;
    double *avg;                       // SOR
    const float4 *prior; float4 *normals; float *cov6;   // normals
    int32_t *dbg_idx; float *dbg_d2; int32_t *dbg_cnt;   // debug (rows unsorted)
    int dbg_visits;                                      // debug: dbg_cnt <- traversal counters
    int32_t *list_idx; int list_pitch;                   // k-best lists: rows of list_pitch int32 per query (-1: empty slot).  SOR: optional output (32);
                                                         //   wavefront kernel (pcr_knn_wave.h): required, its working rows (scratch in the other modes)
    const uint8_t *todo;                                 // optional: only queries with todo[q] != 0 are processed
    unsigned long long *stamps;                          // diagnostics (PCR_KNN_STAMPS): 24 words per wavefront
    int seed_span;                                       // Morton-index half-width of the seed range (-1: k)
    int *zero_a, *zero_b;                                // optional counters of LATER kernels, zeroed here (saves two memset launches)
    uint8_t *hard; int wave_budget;                      // wavefront kernel: optional, queries it gave up after wave_budget batches (1) or served (0)
    int hard_piece;                                      // k_knn_list: 1 = the entries of hard_list are the first queries of 8-query pieces (k_radius_list), 0 = of 64-query wavefronts
    int *hard_list, *hard_count;                         // ... and (hard_list != nullptr, instead of `hard`) the first query of every wavefront it gave up, appended with one atomic
                                                         //   per wavefront; the octet kernel's list form (k_knn_list) serves exactly those
    const uint8_t *keep; const int *pos;                 // optional: search only among points with keep[i] != 0; results and `todo`
                                                         //   are indexed by pos[i] (the compacted order) -- the cleaned cloud needs no tree of its own
};

__device__ static inline double octet_sum(double v) { return pcr_octet_sum(v); }

struct KnnBatch { KnnArgs a[PCR_MAX_BATCH]; };
// One wavefront = 8 Morton-consecutive queries g0 .. g0 + 7 (one per octet): the lambda `item` below.  LIST = false: wavefront w of workgroup b
// serves queries (4 b + w) 8 ...; LIST = true: the queries of the wavefronts the one-query-per-lane kernel gave up (pcr_knn_wave.h: it appends
// the first query of every such wavefront of 64 to a.hard_list, one atomic per wavefront) and nothing else -- a fixed, small grid whose
// wavefronts stride over the 8-query pieces of the listed wavefronts (the full-range launch with a todo mask it replaces started ~100x the
// workgroups to find the same few).  (The body is a lambda INSIDE the kernel function so that `m` and `gstk` stay LDS objects to the
// compiler: passed by reference to a helper they became generic pointers -- 131 VGPRs and 3 wavefronts per SIMD instead of 7.)
template <int MODE, int SLOTS, bool LIST>
__device__ static inline void d_knn_(const KnnArgs &a) {
    constexpr int OPB = KNN_BS / OCT;
    __shared__ OctMeta m;
    __shared__ OctGroupStack gstk[KNN_BS / 64];
    int n_items = 0;
    if (LIST) {
        n_items = *a.hard_count * (a.hard_piece ? 1 : 8);    // written by the kernel before this one on the stream; entries = wavefronts of 64 queries, or (hard_piece) pieces of 8
        if ((int)blockIdx.x * (KNN_BS / 64) >= n_items) return;      // (uniform over the workgroup)
    }
    if (threadIdx.x == 0) m = *a.t.meta;
    if (!LIST && blockIdx.x == 0 && threadIdx.x == 0) { if (a.zero_a) *a.zero_a = 0; if (a.zero_b) *a.zero_b = 0; }
    __syncthreads();
    auto item = [&](const int g0, const size_t stamp_slot) {
    const int n = m.n;
    const int lane = threadIdx.x & 63, oct = lane >> 3, ol = lane & 7;
    const unsigned long long t_begin = wall_clock64();
    const unsigned long long c_begin = a.stamps ? __builtin_readcyclecounter() : 0ull;
    const int qi = g0 + oct;
    const int oq = (a.keep && qi < n) ? a.pos[qi] : qi;                 // output / todo index of this query
    const bool live = qi < n && (!a.keep || a.keep[qi]) && (!a.todo || a.todo[oq]);
    if (__ballot(live) == 0ull) return;                      // nothing to do for this wavefront
    const float4 q = a.t.pts[live ? qi : 0];
    OctetKnn<SLOTS> tk;
    tk.init(a.k, a.r2cap_f, ol);

    // the wavefront's 8 queries are Morton-consecutive points g0..g0+7: seed every query with the index range
    // [g0-k, g0+7+k] (it already holds most true neighbours), then ONE shared bottom-up walk completes all 8 exactly
    const int glast = g0 + OCT - 1 < n - 1 ? g0 + OCT - 1 : n - 1;
    const int span = a.seed_span >= 0 ? a.seed_span : a.k;
    const int half = (a.k < OCT * SLOTS ? a.k : OCT * SLOTS) / 2 + OCT;      // the seed range must hold the direct fill
    const int sp = span > half ? span : half;
    const int plo = g0 - sp < 0 ? 0 : g0 - sp, phi = glast + sp > n - 1 ? n - 1 : glast + sp;
    bool seeding = true;
    int st_scan = 0, st_rounds = 0;      // diagnostics: 8-point scan steps and insertion rounds of this wavefront

    // ---- scan `count` consecutive points from `first` for all 8 queries, 8 candidates at a time; survivors enter the k-best
    auto visit = [&](int first, int count) {
        for (int base = first; base < first + count; base += OCT) {
            float d2 = 0.0f; bool pass = false;
            const int idx = base + ol;
            if (live && idx < first + count && (seeding || idx < plo || idx > phi) && (!a.keep || a.keep[idx])) {
                const float4 p = a.t.pts[idx];
                d2 = pcr_d2(p.x - q.x, p.y - q.y, p.z - q.z);
                pass = d2 < tk.worst;
            }
            unsigned long long bal = __ballot(pass);
            uint32_t surv = (uint32_t)(bal >> (oct * 8)) & 0xffu;
            st_scan++;
            while (bal != 0ull) {                      // one candidate per octet and round, no divergence inside
                st_rounds++;
                const int sl = __builtin_ctz(surv | 0x100u) & 7;
                const float cd = __shfl(d2, sl, OCT);
                tk.insert(surv ? cd : __builtin_inff(), base + sl);
                surv &= surv - 1;
                bal = __ballot(surv != 0);
            }
        }
    };
    // direct fill: the min(k, 8*SLOTS) seeds CENTRED on the group go straight into the slots (slot j of lane l <- candidate
    // c0 + 8 j + l), no insertion rounds while the k-best is not yet full; the outer seeds then mostly fail the bound test
    const int filled = a.k < OCT * SLOTS ? a.k : OCT * SLOTS;
    int c0 = g0 + OCT / 2 - filled / 2;
    if (c0 + filled - 1 > phi) c0 = phi - filled + 1;
    if (c0 < plo) c0 = plo;
#pragma unroll
    for (int j = 0; j < SLOTS; j++) {
        const int idx = c0 + OCT * j + ol;
        if (live && OCT * j + ol < filled && idx <= phi && (!a.keep || a.keep[idx])) {
            const float4 p = a.t.pts[idx];
            const float d2 = pcr_d2(p.x - q.x, p.y - q.y, p.z - q.z);
            if (d2 < a.r2cap_f) { tk.sd[j] = d2; tk.si[j] = idx; }
        }
    }
    tk.sort_lane();
    if (c0 > plo) visit(plo, c0 - plo);
    if (c0 + filled <= phi) visit(c0 + filled, phi - (c0 + filled) + 1);
    seeding = false;

    const float seed_worst = tk.worst;
    int nvis = 0;
    const int first_live = g0 + (__builtin_ctzll(__ballot(live)) >> 3);
    oct_search_group(a.t, m, gstk[threadIdx.x >> 6], live, a.t.leaf_of[first_live], q.x, q.y, q.z, [&]() { return tk.worst; }, visit,
                     [&](int f, int c) { return f >= plo && f + c - 1 <= phi; }, ol,
                     ((MODE == KNN_MODE_DEBUG && a.dbg_visits) || a.stamps) ? &nvis : nullptr);
    if (a.stamps && ol == 0) {
        unsigned long long *w = a.stamps + 24 * stamp_slot;
        w[8 + oct] = ((unsigned long long)__float_as_uint(seed_worst) << 32) | __float_as_uint(tk.worst);
        w[16 + oct] = ((unsigned long long)__float_as_uint(q.x) << 32) | __float_as_uint(q.y);
    }
    if (a.stamps && lane == 0) {
        unsigned long long *w = a.stamps + 24 * stamp_slot;
        w[4] = ((unsigned long long)(unsigned)st_scan << 32) | (unsigned)st_rounds;
        w[5] = ((unsigned long long)__float_as_uint(q.x) << 32) | __float_as_uint(q.y);
        w[6] = ((unsigned long long)__float_as_uint(q.z) << 32) | __float_as_uint(tk.worst);
        w[7] = (unsigned long long)nvis;
        w[0] = t_begin; w[1] = wall_clock64(); w[2] = __builtin_readcyclecounter() - c_begin;
        w[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    if (!live) return;

    // ---- epilogue in float64 on the selected neighbours (inputs are exact float32 -> same values as the oracle)
    const double qx = q.x, qy = q.y, qz = q.z;
    if (MODE == KNN_MODE_SOR) {
        double s = 0, c = 0;
#pragma unroll
        for (int j = 0; j < SLOTS; j++) {
            if (tk.si[j] >= 0) {
                const float4 p = a.t.pts[tk.si[j]];
                const double dx = (double)p.x - qx, dy = (double)p.y - qy, dz = (double)p.z - qz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (d2 < a.r2cap) { s += sqrt(d2); c += 1.0; }
            }
        }
        s = octet_sum(s); c = octet_sum(c);
        if (ol == 0) a.avg[qi] = c > 0 ? s / c : -1.0;
        if (a.list_idx && SLOTS == 4) {
#pragma unroll
            for (int j = 0; j < SLOTS; j++) a.list_idx[(size_t)qi * 32 + ol + OCT * j] = tk.si[j];
        }
    } else if (MODE == KNN_MODE_NORMALS) {
        double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c = 0;
#pragma unroll
        for (int j = 0; j < SLOTS; j++) {
            if (tk.si[j] >= 0) {
                const float4 p = a.t.pts[tk.si[j]];
                const double x = p.x, y = p.y, z = p.z;
                const double dx = x - qx, dy = y - qy, dz = z - qz;
                if (dx * dx + dy * dy + dz * dz < a.r2cap) {
                    cu[0] += x; cu[1] += y; cu[2] += z;
                    cu[3] += x * x; cu[4] += x * y; cu[5] += x * z; cu[6] += y * y; cu[7] += y * z; cu[8] += z * z;
                    c += 1.0;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; t++) cu[t] = octet_sum(cu[t]);
        c = octet_sum(c);
        if (ol == 0) {
            double C6[6];
            if (c >= 3.0) {
                for (int t = 0; t < 9; t++) cu[t] = cu[t] / c;       // cumulants /= n, as Open3D
                C6[0] = cu[3] - cu[0] * cu[0]; C6[1] = cu[4] - cu[0] * cu[1]; C6[2] = cu[5] - cu[0] * cu[2];
                C6[3] = cu[6] - cu[1] * cu[1]; C6[4] = cu[7] - cu[1] * cu[2]; C6[5] = cu[8] - cu[2] * cu[2];
            } else { C6[0] = 1; C6[1] = 0; C6[2] = 0; C6[3] = 1; C6[4] = 0; C6[5] = 1; }
            if (a.cov6) { for (int t = 0; t < 6; t++) a.cov6[(size_t)oq * 6 + t] = (float)C6[t]; }
            if (a.normals) {
                double nv[3];
                d_fast_eigen3x3(C6, nv);
                const double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                double px = 0, py = 0, pz = 0;
                if (a.prior) { const float4 pr = a.prior[oq]; px = pr.x; py = pr.y; pz = pr.z; }
                if (nn == 0.0 || !(nn == nn)) { if (a.prior) { nv[0] = px; nv[1] = py; nv[2] = pz; } else { nv[0] = 0; nv[1] = 0; nv[2] = 1; } }
                if (a.prior && nv[0] * px + nv[1] * py + nv[2] * pz < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
                a.normals[oq] = make_float4((float)nv[0], (float)nv[1], (float)nv[2], 0.0f);
            }
        }
    } else {
        int c = 0;
#pragma unroll
        for (int j = 0; j < SLOTS; j++) {
            const int slot = ol + OCT * j;
            if (slot < a.k) {
                a.dbg_idx[(size_t)qi * a.k + slot] = tk.si[j];
                if (a.dbg_d2) a.dbg_d2[(size_t)qi * a.k + slot] = tk.si[j] >= 0 ? tk.sd[j] : __builtin_inff();
                c += tk.si[j] >= 0 ? 1 : 0;
            }
        }
#pragma unroll
        for (int o = 1; o < OCT; o <<= 1) c += __shfl_xor(c, o, OCT);
        if (ol == 0 && a.dbg_cnt) a.dbg_cnt[qi] = a.dbg_visits == 2 ? (int)(wall_clock64() - t_begin) : (a.dbg_visits ? nvis : c);
    }
    };
    const int wv = threadIdx.x >> 6;
    if (!LIST) item((int)blockIdx.x * OPB + wv * OCT, (size_t)blockIdx.x * (KNN_BS / 64) + wv);
    else for (int it = blockIdx.x * (KNN_BS / 64) + wv; it < n_items; it += gridDim.x * (KNN_BS / 64)) item(a.hard_piece ? a.hard_list[it] : a.hard_list[it >> 3] + (it & 7) * OCT, 0);
}
template <int MODE, int SLOTS> __device__ static inline void d_knn(const KnnArgs &a) { d_knn_<MODE, SLOTS, false>(a); }
template <int MODE, int SLOTS> __device__ static inline void d_knn_list(const KnnArgs &a) { d_knn_<MODE, SLOTS, true>(a); }
template <int MODE, int SLOTS> __global__ void __launch_bounds__(KNN_BS) k_knn(KnnArgs a) { d_knn<MODE, SLOTS>(a); }
template <int MODE, int SLOTS> __global__ void __launch_bounds__(KNN_BS) k_knn_list(KnnArgs a) { d_knn_list<MODE, SLOTS>(a); }
template <int MODE, int SLOTS> __global__ void __launch_bounds__(KNN_BS) k_knn_list_batch(KnnBatch b) { d_knn_list<MODE, SLOTS>(b.a[blockIdx.y]); }
template <int MODE, int SLOTS> __global__ void __launch_bounds__(KNN_BS) k_knn_list_batchp(const KnnArgs *a) { d_knn_list<MODE, SLOTS>(a[blockIdx.y]); }
template <int MODE, int SLOTS> __global__ void __launch_bounds__(KNN_BS) k_knn_batch(KnnBatch b) { d_knn<MODE, SLOTS>(b.a[blockIdx.y]); }
template <int MODE, int SLOTS> __global__ void __launch_bounds__(KNN_BS) k_knn_batchp(const KnnArgs *a) { d_knn<MODE, SLOTS>(a[blockIdx.y]); }

#include "pcr_knn_wave.h"

// Two exact k-NN kernels: the OCTET kernel (8 lanes per query) and the WAVEFRONT kernel (one query per lane, pcr_knn_wave.h, k <= 64, no
// sparse `todo` searches).  The wavefront kernel needs a third fewer VALU instructions (440 against 660 per query at k = 30); ALONE it
// takes the same time or longer -- both run at the chip's VALU issue rate, the wavefront kernel at 4 wavefronts per SIMD (128 VGPRs)
// with a long tail of wavefronts whose 64 queries lie far apart (handed to the octet kernel after a budget), the octet kernel at 7 with
// no tail to speak of (tools/knn_sat.py, 1.4 M queries, k = 30, outlier-filter mode: 1.8 + 0.1 ms against 2.0 ms; one pair at a time
// 182 against 200 pairs/s) -- but in the batched searches of a lockstep group (2 G clouds x S scales per launch, other groups' kernels
// alongside) the tail hides and the instruction count decides: 200k-point pairs 569 -> 602 pairs/s, 100k 913 -> 977, 20k 2736 -> 2880.
// So: the wavefront kernel for batches of at least KNN_WAVE_MIN_BATCH searches or KNN_WAVE_MIN_POINTS queries, the octet kernel otherwise.  PCR_KNN_WAVE=0 / 1 forces
// one of them (per call: tests/test_gpu_stages.py runs every search through both and compares them -- two independent exact searches).
#define KNN_WAVE_MIN_BATCH 6
#define KNN_WAVE_MIN_POINTS 1500000       // ... or of that much query CAPACITY in all (the scales of config 5, 2M each: 21.7 -> 22.6 pairs/s; a lone 200k-point pair, 3 x 200k in a batch, is faster with the octet kernel: 208 against 183 pairs/s)
static bool knn_wave_enabled(const pcr_context *ctx, int batch, long long points) {
    const int forced = pcr_options().knn_wave.load(std::memory_order_relaxed);
    return forced >= 0 ? forced != 0 : (!ctx->octet_only && (ctx->group_forms || batch >= KNN_WAVE_MIN_BATCH || points >= KNN_WAVE_MIN_POINTS));
}
static bool knn_wave_fits(const KnnArgs &a) { return a.k >= 1 && a.k <= 64 && !a.todo && !a.stamps && !a.dbg_visits; }
// the wavefront kernel appends every query's k-best to a row in global memory: the caller's list (SOR) or scratch from the arena
template <int MODE>
static int knn_wave_rows(pcr_context *ctx, KnnArgs &a, int cap) {
    if (MODE == KNN_MODE_DEBUG || (a.list_idx && a.list_pitch >= a.k)) return PCR_OK;
    a.list_pitch = a.k;
    a.list_idx = arena<int32_t>(ctx, (size_t)(cap > 0 ? cap : 1) * (size_t)a.k);
    return a.list_idx ? PCR_OK : PCR_ENOMEM;
}
template <int MODE> static int launch_knn_wave_only(pcr_context *ctx, KnnArgs *a, int count, int mc, int kmax);
static int knn_wave_budget() { return pcr_options().knnw_budget.load(std::memory_order_relaxed); }      // batches of 64 candidates in pass 1 (mean 17, p99 43 at k = 30)
template <int MODE> static int launch_knn_octet_batch(pcr_context *ctx, KnnArgs *a, const int *caps, int count);
// grid of the list form: enough wavefronts for the usual ~1 % of hard wavefronts at once, striding when there are more
static int knn_list_grid(int cap) { const int g = cap / (64 * 16); return g < 8 ? 8 : (g > 128 ? 128 : g); }
template <int MODE>
static int launch_knn_wave_batch(pcr_context *ctx, KnnArgs *a, const int *caps, int count, int mc, int kmax) {
    for (int k = 0; k < count; k++) PCR_TRY(knn_wave_rows<MODE>(ctx, a[k], caps[k]));
    // the queries of the wavefronts that give up (budget, log) go to the octet kernel in a second launch, through a list (k_knn_list)
    const int budget = knn_wave_budget();
    const bool handover = MODE != KNN_MODE_DEBUG && kmax <= 64 && budget > 0;
    if (handover) {
        int *counts = arena<int>(ctx, count);
        if (!counts) return PCR_ENOMEM;
        PCR_HIP_CHECK(ctx, hipMemsetAsync(counts, 0, sizeof(int) * (size_t)count, ctx->stream));
        for (int k = 0; k < count; k++) {
            a[k].hard = nullptr; a[k].wave_budget = budget; a[k].hard_count = counts + k;
            a[k].hard_list = arena<int>(ctx, (size_t)(caps[k] > 0 ? caps[k] : 1) / 64 + 1);
            if (!a[k].hard_list) return PCR_ENOMEM;
        }
    }
    PCR_TRY(launch_knn_wave_only<MODE>(ctx, a, count, mc, kmax));
    if (!handover) return PCR_OK;
    std::vector<KnnArgs> o(a, a + count);
    for (int k = 0; k < count; k++) { o[k].todo = nullptr; o[k].zero_a = nullptr; o[k].zero_b = nullptr; o[k].seed_span = -1; }
    const dim3 grid(knn_list_grid(mc), count), block(KNN_BS);
    if (kmax <= 32) return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_list_batch<MODE, 4>), (k_knn_list_batchp<MODE, 4>), o.data(), count, grid, block);
    return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_list_batch<MODE, 8>), (k_knn_list_batchp<MODE, 8>), o.data(), count, grid, block);
}
template <int MODE>
static int launch_knn_wave_only(pcr_context *ctx, KnnArgs *a, int count, int mc, int kmax) {
    const dim3 grid((unsigned)(((size_t)mc + KW_BS - 1) / KW_BS), count), block(KW_BS);
    if (kmax <= 20) return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_wave_batch<MODE, 20>), (k_knn_wave_batchp<MODE, 20>), a, count, grid, block);
    if (kmax <= 30) return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_wave_batch<MODE, 30>), (k_knn_wave_batchp<MODE, 30>), a, count, grid, block);
    return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_wave_batch<MODE, 64>), (k_knn_wave_batchp<MODE, 64>), a, count, grid, block);
}
template <int MODE>
static int launch_knn_wave(pcr_context *ctx, int cap, KnnArgs a) {
    PCR_TRY(knn_wave_rows<MODE>(ctx, a, cap));
    const dim3 grid((unsigned)(((size_t)cap + KW_BS - 1) / KW_BS)), block(KW_BS);
    if (const char *path = MODE == KNN_MODE_DEBUG ? getenv("PCR_KNNW_STATS") : nullptr) {      // diagnostics: per-wavefront counters of the two passes, appended to a file
        const size_t words = (size_t)grid.x * (KW_BS / 64) * 24;
        unsigned long long *dev = nullptr;
        if (hipMalloc(&dev, words * 8) != hipSuccess) return PCR_ENOMEM;
        (void)hipMemsetAsync(dev, 0, words * 8, ctx->stream);
        a.stamps = dev;
        if (a.k <= 20) PCR_LAUNCH(ctx, (k_knn_wave_stats<MODE, 20>), grid, block, 0, ctx->stream, a);
        else if (a.k <= 30) PCR_LAUNCH(ctx, (k_knn_wave_stats<MODE, 30>), grid, block, 0, ctx->stream, a);
        else PCR_LAUNCH(ctx, (k_knn_wave_stats<MODE, 64>), grid, block, 0, ctx->stream, a);
        (void)hipStreamSynchronize(ctx->stream);
        std::vector<unsigned long long> h(words);
        (void)hipMemcpy(h.data(), dev, words * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(path, "ab")) { const unsigned long long hdr[4] = {0x4b57535441ull, (unsigned long long)a.k, (unsigned long long)cap, words / 24}; fwrite(hdr, 8, 4, f); fwrite(h.data(), 8, words, f); fclose(f); }
        (void)hipFree(dev);
        return PCR_OK;
    }
    const int budget = knn_wave_budget();
    const bool handover = MODE != KNN_MODE_DEBUG && budget > 0;
    if (handover) {
        a.hard = nullptr; a.wave_budget = budget;
        a.hard_count = arena<int>(ctx, 1); a.hard_list = arena<int>(ctx, (size_t)(cap > 0 ? cap : 1) / 64 + 1);
        if (!a.hard_count || !a.hard_list) return PCR_ENOMEM;
        PCR_HIP_CHECK(ctx, hipMemsetAsync(a.hard_count, 0, sizeof(int), ctx->stream));
    }
    if (a.k <= 20) PCR_LAUNCH(ctx, (k_knn_wave<MODE, 20>), grid, block, 0, ctx->stream, a);
    else if (a.k <= 30) PCR_LAUNCH(ctx, (k_knn_wave<MODE, 30>), grid, block, 0, ctx->stream, a);
    else PCR_LAUNCH(ctx, (k_knn_wave<MODE, 64>), grid, block, 0, ctx->stream, a);
    if (handover) {                                  // the queries of the wavefronts that gave up: the octet kernel's list form, 8 per wavefront
        KnnArgs o = a;
        o.todo = nullptr; o.zero_a = nullptr; o.zero_b = nullptr; o.seed_span = -1;
        const dim3 og(knn_list_grid(cap));
        if (o.k <= 32) PCR_LAUNCH(ctx, (k_knn_list<MODE, 4>), og, dim3(KNN_BS), 0, ctx->stream, o);
        else PCR_LAUNCH(ctx, (k_knn_list<MODE, 8>), og, dim3(KNN_BS), 0, ctx->stream, o);
    }
    return PCR_OK;
}

// `count` searches in ONE launch (blockIdx.y picks the problem; k <= 32): the SOR / normals searches of all scales of a cloud
template <int MODE> static int launch_knn_cap(pcr_context *ctx, int cap, KnnArgs a);
template <int MODE>
static int launch_knn_batch(pcr_context *ctx, KnnArgs *a, const int *caps, int count) {
    if (getenv("PCR_KNN_STAMPS")) {                          // diagnostics: one stamped launch per problem
        for (int k = 0; k < count; k++) PCR_TRY(launch_knn_cap<MODE>(ctx, caps[k], a[k]));
        return PCR_OK;
    }
    long long total_pts = 0;
    for (int k = 0; k < count; k++) total_pts += caps[k];
    if (knn_wave_enabled(ctx, count, total_pts)) {
        bool fits = true; int kmax = 0, mcw = 0;
        for (int k = 0; k < count; k++) { fits = fits && knn_wave_fits(a[k]); kmax = a[k].k > kmax ? a[k].k : kmax; mcw = caps[k] > mcw ? caps[k] : mcw; }
        if (fits && mcw > 0) {
            for (int k = 0; k < count; k++) a[k].seed_span = -1;
            return launch_knn_wave_batch<MODE>(ctx, a, caps, count, mcw, kmax);
        }
    }
    return launch_knn_octet_batch<MODE>(ctx, a, caps, count);
}
template <int MODE>
static int launch_knn_octet_batch(pcr_context *ctx, KnnArgs *a, const int *caps, int count) {
    int mc = 0;
    int kmax = 0;
    for (int k = 0; k < count; k++) {
        if (a[k].k < 1 || a[k].k > 200) { ctx->err = "batched k-NN: k must be in 1..200"; return PCR_EINVAL; }
        a[k].seed_span = -1;
        mc = caps[k] > mc ? caps[k] : mc; kmax = a[k].k > kmax ? a[k].k : kmax;
    }
    if (mc <= 0) return PCR_OK;
    // register slots per lane by the largest k of the batch, exactly as the one-cloud launch picks them (launch_knn_cap): same arithmetic
    const dim3 grid((unsigned)(((size_t)mc * OCT + KNN_BS - 1) / KNN_BS), count), block(KNN_BS);
    if (kmax <= 32) return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_batch<MODE, 4>), (k_knn_batchp<MODE, 4>), a, count, grid, block);
    if (kmax <= 64) return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_batch<MODE, 8>), (k_knn_batchp<MODE, 8>), a, count, grid, block);
    return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_batch<MODE, 25>), (k_knn_batchp<MODE, 25>), a, count, grid, block);
}

template <int MODE>
static int launch_knn(pcr_context *ctx, const DevCloud *c, KnnArgs a) { return launch_knn_cap<MODE>(ctx, c->cap, a); }
template <int MODE>
static int launch_knn_cap(pcr_context *ctx, int cap, KnnArgs a) {
    if (cap <= 0) return PCR_OK;
    if (a.k < 1 || a.k > 200) { ctx->err = "k out of range for the octet k-NN kernel (1..200)"; return PCR_EINVAL; }
    const dim3 grid((unsigned)(((size_t)cap * OCT + KNN_BS - 1) / KNN_BS)), block(KNN_BS);
    const char *stamp_path = getenv("PCR_KNN_STAMPS");      // diagnostics only: per-wavefront begin/end clocks and hardware ids
    const size_t stamp_words = (size_t)grid.x * (KNN_BS / 64) * 24;
    if (stamp_path) {
        if (hipMalloc(&a.stamps, stamp_words * 8) != hipSuccess) return PCR_ENOMEM;
        (void)hipMemsetAsync(a.stamps, 0, stamp_words * 8, ctx->stream);
    }
    struct StampDump {
        pcr_context *ctx; const char *path; unsigned long long *dev; size_t words; int mode, k;
        ~StampDump() {
            if (!path) return;
            (void)hipStreamSynchronize(ctx->stream);
            unsigned long long *h = (unsigned long long *)malloc(words * 8);
            (void)hipMemcpy(h, dev, words * 8, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(path, "ab")) {
                const unsigned long long hdr[4] = {0x5354414d50ull, (unsigned long long)mode, (unsigned long long)k, words / 24};
                fwrite(hdr, 8, 4, f); fwrite(h, 8, words, f); fclose(f);
            }
            free(h); (void)hipFree(dev);
        }
    } dump{ctx, stamp_path, a.stamps, stamp_words, MODE, a.k};
    a.seed_span = -1;
    if (knn_wave_enabled(ctx, 1, cap) && knn_wave_fits(a)) { a.seed_span = -1; return launch_knn_wave<MODE>(ctx, cap, a); }
    if (a.k <= 32) PCR_LAUNCH(ctx, k_knn<MODE, 4>, grid, block, 0, ctx->stream, a);
    else if (a.k <= 64) PCR_LAUNCH(ctx, k_knn<MODE, 8>, grid, block, 0, ctx->stream, a);
    else PCR_LAUNCH(ctx, k_knn<MODE, 25>, grid, block, 0, ctx->stream, a);
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
    a.t = oct_view(c); a.n_ptr = c->n; a.k = k;
    knn_radius(a, radius > 0 ? PCR_SEARCH_HYBRID : PCR_SEARCH_KNN, radius);
    if (radius > 0) a.r2cap_f = (float)(radius * radius);
    a.dbg_idx = idx; a.dbg_d2 = d2; a.dbg_cnt = counts; a.dbg_visits = pcr_options().debug_visits.load(std::memory_order_relaxed);
    return launch_knn<KNN_MODE_DEBUG>(ctx, c, a);
}

// ============================================================== pure radius neighbourhoods (KDTreeSearchParamRadius)
// All points with d^2 < r^2 contribute to the moments directly (no k-best): same shared walk, fixed bound.
struct RadArgs {
    OctView t; const int *n_ptr; float r2f; double r2;
    const float4 *prior; float4 *normals; float *cov6;
};
__global__ void __launch_bounds__(KNN_BS) k_radius_moments(RadArgs a) {
    constexpr int OPB = KNN_BS / OCT;
    __shared__ OctMeta m;
    __shared__ OctGroupStack gstk[KNN_BS / 64];
    if (threadIdx.x == 0) m = *a.t.meta;
    __syncthreads();
    const int n = m.n;
    const int lane = threadIdx.x & 63, ol = lane & 7, ob = threadIdx.x >> 3;
    const int qi = blockIdx.x * OPB + ob;
    const bool live = qi < n;
    if (__ballot(live) == 0ull) return;
    const float4 q = a.t.pts[live ? qi : 0];
    const double qx = q.x, qy = q.y, qz = q.z;
    double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c = 0;
    auto visit = [&](int first, int count) {
        for (int base = first; base < first + count; base += OCT) {
            const int idx = base + ol;
            if (live && idx < first + count) {
                const float4 p = a.t.pts[idx];
                if (pcr_d2(p.x - q.x, p.y - q.y, p.z - q.z) < a.r2f) {
                    const double x = p.x, y = p.y, z = p.z;
                    const double dx = x - qx, dy = y - qy, dz = z - qz;
                    if (dx * dx + dy * dy + dz * dz < a.r2) {
                        cu[0] += x; cu[1] += y; cu[2] += z;
                        cu[3] += x * x; cu[4] += x * y; cu[5] += x * z; cu[6] += y * y; cu[7] += y * z; cu[8] += z * z;
                        c += 1.0;
                    }
                }
            }
        }
    };
    const int g0 = blockIdx.x * OPB + (threadIdx.x >> 6) * OCT;
    oct_search_group(a.t, m, gstk[threadIdx.x >> 6], live, a.t.leaf_of[g0], q.x, q.y, q.z, [&]() { return a.r2f; }, visit,
                     [](int, int) { return false; }, ol, nullptr);
#pragma unroll
    for (int t = 0; t < 9; t++) cu[t] = octet_sum(cu[t]);
    c = octet_sum(c);
    if (live && ol == 0) {
        double C6[6];
        if (c >= 3.0) {
            for (int t = 0; t < 9; t++) cu[t] = cu[t] / c;
            C6[0] = cu[3] - cu[0] * cu[0]; C6[1] = cu[4] - cu[0] * cu[1]; C6[2] = cu[5] - cu[0] * cu[2];
            C6[3] = cu[6] - cu[1] * cu[1]; C6[4] = cu[7] - cu[1] * cu[2]; C6[5] = cu[8] - cu[2] * cu[2];
        } else { C6[0] = 1; C6[1] = 0; C6[2] = 0; C6[3] = 1; C6[4] = 0; C6[5] = 1; }
        if (a.cov6) { for (int t = 0; t < 6; t++) a.cov6[(size_t)qi * 6 + t] = (float)C6[t]; }
        if (a.normals) {
            double nv[3];
            d_fast_eigen3x3(C6, nv);
            const double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
            double px = 0, py = 0, pz = 0;
            if (a.prior) { const float4 pr = a.prior[qi]; px = pr.x; py = pr.y; pz = pr.z; }
            if (nn == 0.0 || !(nn == nn)) { if (a.prior) { nv[0] = px; nv[1] = py; nv[2] = pz; } else { nv[0] = 0; nv[1] = 0; nv[2] = 1; } }
            if (a.prior && nv[0] * px + nv[1] * py + nv[2] * pz < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
            a.normals[qi] = make_float4((float)nv[0], (float)nv[1], (float)nv[2], 0.0f);
        }
    }
}

// ============================================== neighbour lists under a radius cap (the Hybrid(r, max_nn) lists of FPFH, round 4)
// On an outdoor scan most points have FEWER than max_nn = 200 neighbours inside r (NCLT: 50-68 on average, 1-8 % capped), and then the list is
// simply every point inside the ball -- no k-best selection.  The k-best kernel nevertheless ran every in-ball candidate through a 25-slot
// sorted insertion (99 VALU instructions per round): 13.6 % of the FGR stage.  Here an octet APPENDS what it finds (same float32 test against
// r^2 as the k-best kernel, same shared walk with the fixed bound): row = the query's in-ball points in walk order, cnt = their number.  A
// piece of 8 queries in which one ball holds more than k points is listed (one atomic per piece) and redone by the k-best kernel's list form
// (k_knn_list with hard_piece), which writes the k nearest in its slot layout; cnt = -1 tells the readers to scan all k slots of such a row.
struct RadListArgs {
    OctView t; const int *n_ptr; float r2f; int k;
    int32_t *idx; int32_t *cnt;                  // rows of k int32 per query; entries per row (-1: row in the k-best kernel's slot layout, scan all k)
    int *over_list, *over_count;                 // first queries of the 8-query pieces left to the k-best kernel
    int select;                                  // 1: overfull balls are finished here (threshold selection); 0: every overfull piece goes to the k-best kernel
};
// Round 5: an OVERFULL ball (23 % of the balls of a 175k-point cloud at r = 1 m, max_nn = 200) no longer goes through the k-best kernel's
// 25-slot sorted insertion.  FPFH needs the SET of the k nearest, not their order: the first walk also counts the in-ball points of each
// query in RL_BINS bins of d^2 (LDS), the bin B in which the count passes k is read off, and a second walk -- bounded by that bin's upper edge --
// appends every point of the bins below B and collects the points of bin B (a dozen) into LDS, of which the k - below smallest complete the
// row -- smallest by (float64 d^2, caller index), the order of the reference's k-d tree (oracle/kdtree.c cmp_item), so that a tie at the k-th
// place falls as it does there (the k-best kernel keeps the first one its walk meets).  Both walks test the same float32 expressions, so they see the same points.  A boundary bin of more than RL_EDGE points
// (many equal distances) sends the piece to the k-best kernel as before.
#define RL_BINS 128            // (two 16-bit counters per LDS word: a ball holds far fewer than 65 536 points)
#define RL_EDGE 20
// squared distance as the reference's k-d tree forms it (float64 differences of the float32 coordinates, products and sums rounded one by one)
__device__ static inline double pcr_d2_f64_unfused(const float4 q, const float4 p) {
#pragma clang fp contract(off)
    const double ex = (double)q.x - (double)p.x, ey = (double)q.y - (double)p.y, ez = (double)q.z - (double)p.z;
    double d2 = ex * ex;
    d2 += ey * ey;
    d2 += ez * ez;
    return d2;
}
__device__ static inline void d_radius_list(const RadListArgs &a) {
    constexpr int OPB = KNN_BS / OCT;
    __shared__ OctMeta m;
    __shared__ OctGroupStack gstk[KNN_BS / 64];
    __shared__ unsigned hist[OPB][RL_BINS / 2];
    __shared__ double edge_d2[OPB][RL_EDGE];     // the boundary bin's points: float64 distance (as the reference's k-d tree orders them),
    __shared__ int2 edge_id[OPB][RL_EDGE];       //   (caller index, index in Morton order)
    __shared__ int nedge[OPB];
    if (threadIdx.x == 0) m = *a.t.meta;
    for (int b = threadIdx.x; b < OPB * RL_BINS / 2; b += KNN_BS) (&hist[0][0])[b] = 0u;
    if (threadIdx.x < OPB) nedge[threadIdx.x] = 0;
    __syncthreads();
    const int n = m.n;
    const int lane = threadIdx.x & 63, oct = lane >> 3, ol = lane & 7, ob = threadIdx.x >> 3;
    const int qi = blockIdx.x * OPB + ob;
    const bool live = qi < n;
    if (__ballot(live) == 0ull) return;
    const float4 q = a.t.pts[live ? qi : 0];
    int32_t *const row = a.idx + (size_t)(live ? qi : 0) * a.k;
    const float bin_scale = (float)RL_BINS / a.r2f;
    const bool select = a.select != 0;
    int cnt = 0;                                 // octet-uniform
    auto visit = [&](int first, int count) {
        for (int base = first; base < first + count; base += OCT) {
            const int idx = base + ol;
            bool hit = false; float d2 = 0.0f;
            if (live && (select || cnt <= a.k) && idx < first + count) {
                const float4 p = a.t.pts[idx];
                d2 = pcr_d2(p.x - q.x, p.y - q.y, p.z - q.z);
                hit = d2 < a.r2f;
            }
            const unsigned mask = (unsigned)(__ballot(hit) >> (oct * 8)) & 0xffu;
            const int pos = cnt + __builtin_popcount(mask & ((1u << ol) - 1u));
            if (hit && pos < a.k) row[pos] = idx;
            if (hit && select) { int b = (int)(d2 * bin_scale); b = b < RL_BINS ? b : RL_BINS - 1; atomicAdd(&hist[ob][b >> 1], 1u << (16 * (b & 1))); }
            cnt += __builtin_popcount(mask);
        }
    };
    const int g0 = blockIdx.x * OPB + (threadIdx.x >> 6) * OCT;
    const int leaf0 = a.t.leaf_of[g0 < n ? g0 : 0];
    // (without the selection a ball that is overfull already needs nothing more from the walk: its bound drops to nothing)
    oct_search_group(a.t, m, gstk[threadIdx.x >> 6], live, leaf0, q.x, q.y, q.z, [&]() { return (!select && cnt > a.k) ? -1.0f : a.r2f; }, visit,
                     [](int, int) { return false; }, ol, nullptr);
    const bool mine = live && cnt > a.k;                                   // octet-uniform
    bool over = __ballot(mine) != 0ull;                                    // (a wavefront = one piece of 8 queries)
    if (over && select) {
        // the bin in which the count passes k (octet-uniform: every lane reads the octet's 64 counters)
        int B = 0, below = 0;
        if (mine) {
            for (int b = 0; b < RL_BINS; b++) { const int h = (int)((hist[ob][b >> 1] >> (16 * (b & 1))) & 0xffffu); if (below + h >= a.k) { B = b; break; } below += h; }
        }
        const float bound2 = mine ? fminf(a.r2f, (float)(B + 1) / bin_scale * 1.0001f) : -1.0f;
        int cnt2 = 0;
        auto visit2 = [&](int first, int count) {
            for (int base = first; base < first + count; base += OCT) {
                const int idx = base + ol;
                bool lo = false, eq = false; float d2 = 0.0f;
                if (mine && idx < first + count) {
                    const float4 p = a.t.pts[idx];
                    d2 = pcr_d2(p.x - q.x, p.y - q.y, p.z - q.z);
                    if (d2 < a.r2f) { int b = (int)(d2 * bin_scale); b = b < RL_BINS ? b : RL_BINS - 1; lo = b < B; eq = b == B; }
                }
                const unsigned mask = (unsigned)(__ballot(lo) >> (oct * 8)) & 0xffu;
                const int pos = cnt2 + __builtin_popcount(mask & ((1u << ol) - 1u));
                if (lo && pos < below) row[pos] = idx;
                cnt2 += __builtin_popcount(mask);
                if (eq) {
                    const int e = atomicAdd(&nedge[ob], 1);
                    if (e < RL_EDGE) {
                        const float4 p = a.t.pts[idx];
                        edge_d2[ob][e] = pcr_d2_f64_unfused(q, p); edge_id[ob][e] = make_int2(__float_as_int(p.w), idx);
                    }
                }
            }
        };
        oct_search_group(a.t, m, gstk[threadIdx.x >> 6], mine, leaf0, q.x, q.y, q.z, [&]() { return bound2; }, visit2,
                         [](int, int) { return false; }, ol, nullptr);
        const int ne = mine ? nedge[ob] : 0;
        const bool spill = ne > RL_EDGE;
        over = __ballot(spill) != 0ull;                                    // such a piece is the k-best kernel's
        if (!over && mine) {
            const int need = a.k - below;                                  // 1 <= need <= ne
            for (int e = ol; e < ne; e += OCT) {
                const double md = edge_d2[ob][e]; const int2 mi = edge_id[ob][e];
                int rank = 0;
                for (int f = 0; f < ne; f++) {
                    const double od = edge_d2[ob][f]; const int2 oi = edge_id[ob][f];
                    rank += (od < md || (od == md && (oi.x < mi.x || (oi.x == mi.x && oi.y < mi.y)))) ? 1 : 0;
                }
                if (rank < need) row[below + rank] = mi.y;
            }
            cnt = a.k;
        }
    }
    if (live && ol == 0) a.cnt[qi] = over ? -1 : cnt;
    if (over && lane == 0) a.over_list[atomicAdd(a.over_count, 1)] = g0;
}
__global__ void __launch_bounds__(KNN_BS) k_radius_list(RadListArgs a) { d_radius_list(a); }
__global__ void __launch_bounds__(KNN_BS) k_radius_list_g(const RadListArgs *a) { d_radius_list(a[blockIdx.y]); }
// `count` clouds in one launch pair: the append kernel, then the k-best list form over the listed pieces.  Rows idx[c]: cap x k int32,
// cnt[c]: cap int32.  (Hybrid search only: radius > 0.)
int pcr_dev_radius_lists_batch(pcr_context *ctx, const DevCloud *const *cs, int count, int k, double radius, int32_t *const *idx, int32_t *const *cnt) {
    if (count < 1) return PCR_OK;
    if (count > PCR_MAX_GROUP_BATCH) { ctx->err = "radius list batch size"; return PCR_EINVAL; }
    if (!(radius > 0) || k < 1 || k > 200) { ctx->err = "radius lists: radius <= 0 or k outside 1..200"; return PCR_EINVAL; }
    std::vector<RadListArgs> ra; std::vector<KnnArgs> ka; std::vector<int> caps;
    int *counts = arena<int>(ctx, count);
    if (!counts) return PCR_ENOMEM;
    PCR_HIP_CHECK(ctx, hipMemsetAsync(counts, 0, sizeof(int) * (size_t)count, ctx->stream));
    int mc = 0;
    for (int c = 0; c < count; c++) {
        if (cs[c]->cap <= 0) continue;
        RadListArgs r;
        r.t = oct_view(cs[c]); r.n_ptr = cs[c]->n; r.r2f = (float)(radius * radius); r.k = k; r.idx = idx[c]; r.cnt = cnt[c];
        r.over_list = arena<int>(ctx, (size_t)cs[c]->cap / OCT + 1); r.over_count = counts + c;
        r.select = pcr_options().radius_list_select.load(std::memory_order_relaxed);
        if (!r.over_list) return PCR_ENOMEM;
        ra.push_back(r);
        KnnArgs a; std::memset(&a, 0, sizeof a);                        // the k-best search of pcr_dev_knn_debug, over the listed pieces only
        a.t = r.t; a.n_ptr = cs[c]->n; a.k = k;
        knn_radius(a, PCR_SEARCH_HYBRID, radius);
        a.r2cap_f = r.r2f;
        a.dbg_idx = idx[c]; a.dbg_d2 = nullptr; a.dbg_cnt = nullptr; a.seed_span = -1;
        a.hard_piece = 1; a.hard_list = r.over_list; a.hard_count = r.over_count;
        ka.push_back(a); caps.push_back(cs[c]->cap);
        mc = cs[c]->cap > mc ? cs[c]->cap : mc;
    }
    const int m = (int)ra.size();
    if (m == 0) return PCR_OK;
    if (m == 1) {
        PCR_LAUNCH(ctx, k_radius_list, dim3((unsigned)(((size_t)mc * OCT + KNN_BS - 1) / KNN_BS)), dim3(KNN_BS), 0, ctx->stream, ra[0]);
    } else {
        const RadListArgs *d = pcr_desc_upload(ctx, ra.data(), m);
        if (!d) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_radius_list_g, dim3((unsigned)(((size_t)mc * OCT + KNN_BS - 1) / KNN_BS), m), dim3(KNN_BS), 0, ctx->stream, d);
    }
    // the overfull pieces: as many wavefronts as a dense cloud may list (every piece), striding
    int lg = mc / (OCT * (KNN_BS / 64) * 4); lg = lg < 8 ? 8 : (lg > 2048 ? 2048 : lg);
    const dim3 grid(lg, m), block(KNN_BS);
    if (k <= 32) return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_list_batch<KNN_MODE_DEBUG, 4>), (k_knn_list_batchp<KNN_MODE_DEBUG, 4>), ka.data(), m, grid, block);
    if (k <= 64) return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_list_batch<KNN_MODE_DEBUG, 8>), (k_knn_list_batchp<KNN_MODE_DEBUG, 8>), ka.data(), m, grid, block);
    return PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_list_batch<KNN_MODE_DEBUG, 25>), (k_knn_list_batchp<KNN_MODE_DEBUG, 25>), ka.data(), m, grid, block);
}

// ============================================= normals of the CLEANED cloud from the SOR pass's k-best lists (K5')
// The k_nrm nearest KEPT neighbours of a kept point are among its k_list nearest in the un-cleaned cloud whenever
// at least k_nrm of those survive the filter (every kept point outside the list is farther than all list members):
// then no second search is needed.  The (rare) other points are flagged `todo` and searched over the cleaned tree.
struct NflArgs {
    const float4 *pts; const int *n_ptr;             // un-cleaned (voxel) cloud
    const int32_t *lidx;                             // rows of 32 int32 per point (-1: empty slot); distances are recomputed here
    const uint8_t *keep; const int *pos;             // filter result, old -> new index
    int k_list, k_nrm;
    const float4 *prior; float4 *normals;            // cleaned order
    uint8_t *todo; int *todo_count;
    int *piece_list, *piece_count;                   // optional: first point (un-cleaned index) of every 8-point piece that holds a todo point -- the work list of the fallback search
};
struct NflBatch { NflArgs a[PCR_MAX_BATCH]; };
// A wavefront serves 64 points: octet o takes points base + 8 o + r in rounds r = 0..7 (list filtering, farthest-survivor drops, raw
// moments: 8 lanes per point as before), lane r of the octet keeps the moments of round r, and after the eighth round ALL 64 lanes
// run the analytic eigen solver at once -- it is ~2/3 of this kernel's instructions and used to run with one live lane per octet.
__device__ static inline void d_normals_from_lists(const NflArgs &a) {
    const int n = *a.n_ptr;
    const int ol = threadIdx.x & 7, oct_id = (threadIdx.x & 63) >> 3;
    const int base = (blockIdx.x * (KNN_BS / 64) + (threadIdx.x >> 6)) * 64;
    if (base >= n) return;
    double my_cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, my_c = 0;      // raw moments of the point this lane solves (the divisions wait until all 64 lanes have one)
    bool my_exact = false; int my_j = 0;
    bool piece_todo = false;                                        // (octet-uniform) one of the octet's 8 points -- the piece [base + 8 o, + 8) -- needs the search
#pragma unroll 1
    for (int r = 0; r < OCT; r++) {
        const int i = base + oct_id * OCT + r;
        const bool act = i < n && a.keep[i];
        int id[4]; float d[4]; bool ok[4]; float4 pn[4];
        int lc = 0, vc = 0;
        const float4 qp = a.pts[act ? i : 0];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int slot = ol + OCT * s;
            id[s] = (act && slot < a.k_list) ? a.lidx[(size_t)i * 32 + slot] : -1;
            pn[s] = a.pts[id[s] >= 0 ? id[s] : 0];
            d[s] = id[s] >= 0 ? pcr_d2(pn[s].x - qp.x, pn[s].y - qp.y, pn[s].z - qp.z) : -1.0f;      // the search's own float32 distance
            ok[s] = id[s] >= 0 && a.keep[id[s]];
            lc += id[s] >= 0 ? 1 : 0; vc += ok[s] ? 1 : 0;
        }
        lc = pcr_octet_sum_i(lc); vc = pcr_octet_sum_i(vc);
        const bool exact = act && (vc >= a.k_nrm || lc < a.k_list);
        const int j = act ? a.pos[i] : 0;
        if (act && ol == 0) {
            a.todo[j] = exact ? 0 : 1;
            if (!exact) atomicAdd(a.todo_count, 1);
        }
        piece_todo = piece_todo || (act && !exact);
        // drop the farthest survivors until k_nrm remain (octet arg-max rounds; octets that are done idle along)
        int excess = exact ? vc - a.k_nrm : 0;
        while (__ballot(excess > 0) != 0ull) {
            float m = -1.0f; int ms = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) if (ok[s] && d[s] > m) { m = d[s]; ms = s; }
            const float om = pcr_octet_max(m);                                   // DPP, no LDS hop; ties -> lowest lane
            const int ml = __builtin_ctz((uint32_t)(__ballot(m == om) >> (threadIdx.x & 56)) & 0xffu);
            const bool hit = excess > 0 && ol == ml;
#pragma unroll
            for (int s = 0; s < 4; s++) ok[s] = ok[s] && !(hit && s == ms);
            excess -= excess > 0 ? 1 : 0;
        }
        double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (exact && ok[s]) {
                const float4 p = pn[s];
                const double x = p.x, y = p.y, z = p.z;
                cu[0] += x; cu[1] += y; cu[2] += z;
                cu[3] += x * x; cu[4] += x * y; cu[5] += x * z; cu[6] += y * y; cu[7] += y * z; cu[8] += z * z;
                c += 1.0;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; t++) cu[t] = octet_sum(cu[t]);
        c = octet_sum(c);
        if (ol == r) {                                  // this lane solves the point of round r
            my_exact = exact; my_j = j; my_c = c;
#pragma unroll
            for (int t = 0; t < 9; t++) my_cu[t] = cu[t];
        }
    }
    // the fallback search serves exactly the listed pieces (k_knn_list, hard_piece): the full-range launch with the todo mask started a wavefront
    // per 8 points of the whole cloud to find the ~2 % that hold one (8.9 % of the headline run's kernel time)
    if (a.piece_list && piece_todo && ol == 0) a.piece_list[atomicAdd(a.piece_count, 1)] = base + oct_id * OCT;
    if (!my_exact) return;
    double myC[6] = {1, 0, 0, 1, 0, 1};
    if (my_c >= 3.0) {                                  // (the nine float64 divisions once per wavefront, not once per round under a one-lane-in-eight branch)
#pragma unroll
        for (int t = 0; t < 9; t++) my_cu[t] = my_cu[t] / my_c;
        myC[0] = my_cu[3] - my_cu[0] * my_cu[0]; myC[1] = my_cu[4] - my_cu[0] * my_cu[1]; myC[2] = my_cu[5] - my_cu[0] * my_cu[2];
        myC[3] = my_cu[6] - my_cu[1] * my_cu[1]; myC[4] = my_cu[7] - my_cu[1] * my_cu[2]; myC[5] = my_cu[8] - my_cu[2] * my_cu[2];
    }
    double nv[3];
    d_fast_eigen3x3(myC, nv);
    const double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    double px = 0, py = 0, pz = 0;
    if (a.prior) { const float4 pr = a.prior[my_j]; px = pr.x; py = pr.y; pz = pr.z; }
    if (nn == 0.0 || !(nn == nn)) { if (a.prior) { nv[0] = px; nv[1] = py; nv[2] = pz; } else { nv[0] = 0; nv[1] = 0; nv[2] = 1; } }
    if (a.prior && nv[0] * px + nv[1] * py + nv[2] * pz < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
    a.normals[my_j] = make_float4((float)nv[0], (float)nv[1], (float)nv[2], 0.0f);
}

__global__ void __launch_bounds__(KNN_BS) k_normals_from_lists(NflArgs a) { d_normals_from_lists(a); }
__global__ void __launch_bounds__(KNN_BS) k_normals_from_lists_batch(NflBatch b) { d_normals_from_lists(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(KNN_BS) k_normals_from_lists_batchp(const NflArgs *a) { d_normals_from_lists(a[blockIdx.y]); }

// ============================================================================ SOR (K4)
// mean / Bessel std of the per-point mean neighbour distance in ONE pass over <= 128 workgroups: shifted moments
// sum(v-c), sum((v-c)^2), count with c = the first valid value (no cancellation: |mean-c| ~ sigma), fixed summation
// tree, write-through partial rows + ticket, the last workgroup gathers them with sc1 loads (no fence) and finishes.
#define SOR_STAT_BLOCKS 128
struct SorStatArgs { const double *avg; const int *n_ptr; double std_ratio; double *out3; double *partials /* SOR_STAT_BLOCKS x 4 */; unsigned int *ticket; };
struct SorStatBatch { SorStatArgs a[PCR_MAX_BATCH]; };
__device__ static inline void d_sor_stats(const SorStatArgs &aa) {
    const double *__restrict__ avg = aa.avg; const int *__restrict__ n_ptr = aa.n_ptr; const double std_ratio = aa.std_ratio;
    double *__restrict__ out3 = aa.out3; double *__restrict__ partials = aa.partials; unsigned int *__restrict__ ticket = aa.ticket;
    __shared__ double red[16][3];
    __shared__ double rows[SOR_STAT_BLOCKS * 3];
    __shared__ int is_last;
    const int n = *n_ptr, t = threadIdx.x;
    double c = 0.0;
    for (int i = 0; i < n && i < 64; i++) { const double v = avg[i]; if (v > 0) { c = v; break; } }     // same value in every workgroup
    double a = 0, b = 0, cnt = 0;
    for (int i = blockIdx.x * 256 + t; i < n; i += gridDim.x * 256) { const double v = avg[i]; if (v > 0) { const double d = v - c; a += d; b += d * d; cnt += 1.0; } }
    a = pcr_row16_sum(a); b = pcr_row16_sum(b); cnt = pcr_row16_sum(cnt);
    if ((t & 15) == 0) { red[t >> 4][0] = a; red[t >> 4][1] = b; red[t >> 4][2] = cnt; }
    __syncthreads();
    if (t < 3) {
        double s = 0;
        for (int r = 0; r < 16; r++) s += red[r][t];
        __hip_atomic_store(&partials[blockIdx.x * 4 + t], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    for (int e = t; e < 3 * (int)gridDim.x; e += 256) {
        const double *p = partials + (e / 3) * 4 + (e % 3);
        double v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
        rows[e] = v;
    }
    __syncthreads();
    if (t == 0) {
        double sa = 0, sb = 0, sc = 0;
        for (int r = 0; r < (int)gridDim.x; r++) { sa += rows[3 * r]; sb += rows[3 * r + 1]; sc += rows[3 * r + 2]; }
        const double valid = sc;
        const double mean = valid > 0 ? c + sa / valid : 0.0;
        const double var = valid > 1 ? (sb - sa * sa / valid) / (valid - 1.0) : 0.0;
        const double sd = sqrt(var > 0 ? var : 0.0);
        out3[0] = mean; out3[1] = sd; out3[2] = valid > 0 ? mean + std_ratio * sd : -1.0;
        *ticket = 0u;
    }
}
__global__ void __launch_bounds__(256) k_sor_stats(SorStatArgs a) { d_sor_stats(a); }
__global__ void __launch_bounds__(256) k_sor_stats_batch(SorStatBatch b) { d_sor_stats(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(256) k_sor_stats_batchp(const SorStatArgs *a) { d_sor_stats(a[blockIdx.y]); }

struct CompactArgs {
    const float4 *pts, *nrm; const uint8_t *flags; const int *pos; const int *n_ptr; float4 *out_pts, *out_nrm;
    const uint64_t *keys; uint64_t *out_keys; int *cnt_in_out, *cnt_kept_out; const int *kept_n;
};
struct CompactBatch { CompactArgs a[PCR_MAX_BATCH]; };
__device__ static inline void d_compact_cloud(const CompactArgs &a) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i == 0) { if (a.cnt_in_out) *a.cnt_in_out = *a.n_ptr; if (a.cnt_kept_out) *a.cnt_kept_out = *a.kept_n; }    // counts for the host, one copy later
    if (i >= *a.n_ptr || !a.flags[i]) return;
    const int o = a.pos[i];
    a.out_pts[o] = a.pts[i];
    a.out_keys[o] = a.keys[i];
    if (a.nrm && a.out_nrm) a.out_nrm[o] = a.nrm[i];
}
__global__ void __launch_bounds__(BS) k_compact_cloud(CompactArgs a) { d_compact_cloud(a); }
__global__ void __launch_bounds__(BS) k_compact_cloud_batch(CompactBatch b) { d_compact_cloud(b.a[blockIdx.y]); }
__global__ void __launch_bounds__(BS) k_compact_cloud_batchp(const CompactArgs *a) { d_compact_cloud(a[blockIdx.y]); }

// The SOR chain of `count` clouds in ONE launch per stage (blockIdx.y picks the cloud): k-NN + k-best lists, statistics, keep flags
// inside the two-kernel scan, compaction, normals of the cleaned cloud from the lists, exact fallback search for the incomplete lists.
// count == 1 is the single-cloud call; count > 1 serves the voxel clouds of all scales of a multiscale registration (they do not
// depend on each other), 7 launches instead of 7 per scale.  Per-problem arguments as in pcr_dev_sor.
struct SorProblem {
    const DevCloud *in; DevCloud *out; uint8_t *keep_sorted; double *avg_sorted; const float4 *prior_out; uint8_t *todo_out; int *todo_count;
    int *cnt_in_out, *cnt_kept_out;
};
static int sor_batch(pcr_context *ctx, SorProblem *pr, int count, int nb_neighbors, double std_ratio, int normal_k, bool fallback_here) {
    if (nb_neighbors < 1 || !(std_ratio > 0.0)) { ctx->err = "nb_neighbors < 1 or std_ratio <= 0"; return PCR_EINVAL; }
    if (count < 1 || count > PCR_MAX_GROUP_BATCH) { ctx->err = "SOR batch size"; return PCR_EINVAL; }
    std::vector<KnnArgs> kbv((size_t)count), fbv((size_t)count); std::vector<SorStatArgs> sbv((size_t)count); std::vector<ScanArgs> cbv((size_t)count);
    std::vector<CompactArgs> mbv((size_t)count); std::vector<NflArgs> nbv((size_t)count);
    std::memset(kbv.data(), 0, sizeof(KnnArgs) * (size_t)count); std::memset(fbv.data(), 0, sizeof(KnnArgs) * (size_t)count);
    std::memset(sbv.data(), 0, sizeof(SorStatArgs) * (size_t)count); std::memset(cbv.data(), 0, sizeof(ScanArgs) * (size_t)count);
    std::memset(mbv.data(), 0, sizeof(CompactArgs) * (size_t)count); std::memset(nbv.data(), 0, sizeof(NflArgs) * (size_t)count);
    struct { KnnArgs *a; } kb{kbv.data()}, fb{fbv.data()}; struct { SorStatArgs *a; } sb{sbv.data()}; struct { ScanArgs *a; } cb{cbv.data()};
    struct { CompactArgs *a; } mb{mbv.data()}; struct { NflArgs *a; } nb_{nbv.data()};
    std::vector<int> capsv((size_t)count); int *caps = capsv.data(); int m = 0;
    bool any_todo = false, fuse_all = true;
    // work lists of the fallback normals (pieces of 8 points that hold a todo point, filled by k_normals_from_lists): one counter per problem, zeroed together
    int *piece_counts = (fallback_here && normal_k > 0 && normal_k <= 32) ? arena<int>(ctx, count) : nullptr;
    if (piece_counts) PCR_HIP_CHECK(ctx, hipMemsetAsync(piece_counts, 0, sizeof(int) * (size_t)count, ctx->stream));
    for (int k = 0; k < count; k++) {
        SorProblem &q = pr[k];
        const DevCloud *in = q.in; DevCloud *out = q.out;
        for (int d = 0; d < 3; d++) { out->key_org[d] = in->key_org[d]; out->key_unit[d] = in->key_unit[d]; }
        out->voxel_lattice = in->voxel_lattice;
        if (in->cap <= 0) { PCR_HIP_CHECK(ctx, hipMemsetAsync(out->n, 0, sizeof(int), ctx->stream)); continue; }
        double *avg = q.avg_sorted ? q.avg_sorted : arena<double>(ctx, in->cap);
        double *stats3 = arena<double>(ctx, 4);
        double *stat_partials = arena<double>(ctx, SOR_STAT_BLOCKS * 4);
        unsigned int *stat_ticket = arena<unsigned int>(ctx, 1);
        uint8_t *flags = q.keep_sorted ? q.keep_sorted : arena<uint8_t>(ctx, in->cap);
        int *pos = arena<int>(ctx, in->cap);
        if (!avg || !stats3 || !stat_partials || !stat_ticket || !flags || !pos) return PCR_ENOMEM;
        // normals of the cleaned cloud straight from this pass's lists when they can be exact (see k_normals_from_lists)
        const bool fuse = normal_k > 0 && q.todo_out && nb_neighbors <= 32 && normal_k <= nb_neighbors;
        const int pitch = nb_neighbors <= 32 ? 32 : 64;           // k-best rows: input of k_normals_from_lists (pitch 32) and the working rows of the wavefront search
        int32_t *lidx = (fuse || nb_neighbors <= 64) ? arena<int32_t>(ctx, (size_t)in->cap * pitch) : nullptr;
        if ((fuse || nb_neighbors <= 64) && !lidx) return PCR_ENOMEM;
        KnnArgs &a = kb.a[m];
        a.t = oct_view(in); a.n_ptr = in->n; a.k = nb_neighbors; a.avg = avg; a.list_idx = lidx; a.list_pitch = pitch;
        a.zero_a = (int *)stat_ticket; a.zero_b = (fuse && q.todo_out) ? q.todo_count : nullptr;      // zeroed by the search kernel for the kernels after it
        a.seed_span = -1;
        knn_radius(a, PCR_SEARCH_KNN, 0);
        sb.a[m] = SorStatArgs{avg, in->n, std_ratio, stats3, stat_partials, stat_ticket};
        PCR_TRY(scan_args(ctx, &cb.a[m], flags, in->n, in->cap, pos, out->n, FlagSrc{nullptr, avg, stats3}));      // keep flags produced inside the scan
        mb.a[m] = CompactArgs{in->pts, in->nrm, flags, pos, in->n, out->pts, out->nrm, in->keys, out->keys, q.cnt_in_out, q.cnt_kept_out, out->n};
        if (q.todo_out) {
            any_todo = true;
            if (fuse) {
                NflArgs &f = nb_.a[m];
                f.pts = in->pts; f.n_ptr = in->n; f.lidx = lidx; f.keep = flags; f.pos = pos; f.k_list = nb_neighbors; f.k_nrm = normal_k;
                f.prior = q.prior_out; f.normals = out->nrm_final; f.todo = q.todo_out; f.todo_count = q.todo_count;
            } else {
                fuse_all = false;
                PCR_HIP_CHECK(ctx, hipMemsetAsync(q.todo_out, 1, (size_t)in->cap, ctx->stream));
            }
            if (fallback_here && normal_k > 0) {
                // the incomplete lists are searched right here, over the INPUT cloud's tree restricted to the kept points: the
                // cleaned cloud then needs no tree of its own (7 launches less; only a GICP target needs one)
                KnnArgs &b = fb.a[m];
                // (lists that cannot serve -- normal_k > nb_neighbors: config 5's 64-NN normals -- leave EVERY kept point to this search: no todo
                // mask then, so that it may run as the one-query-per-lane kernel; with the mask of all ones it ran as the octet kernel over the
                // whole cloud, 25 % of config 5's kernel time)
                b.t = oct_view(in); b.n_ptr = in->n; b.k = normal_k; b.prior = q.prior_out; b.normals = out->nrm_final; b.todo = fuse ? q.todo_out : nullptr;
                b.keep = flags; b.pos = pos; b.seed_span = -1;
                knn_radius(b, PCR_SEARCH_KNN, 0);
                if (fuse && piece_counts) {
                    int *pl = arena<int>(ctx, (size_t)in->cap / OCT + 1);
                    if (!pl) return PCR_ENOMEM;
                    nb_.a[m].piece_list = pl; nb_.a[m].piece_count = piece_counts + m;
                    b.hard_list = pl; b.hard_count = piece_counts + m; b.hard_piece = 1;
                }
            }
        } else fuse_all = false;
        caps[m] = in->cap;
        m++;
    }
    if (m == 0) return PCR_OK;
    int mc = 0;
    for (int k = 0; k < m; k++) mc = caps[k] > mc ? caps[k] : mc;
    if (m == 1 && nb_neighbors > 32) {      // the single-cloud call with a long list: the wide-slot kernels
        DevCloud tmp; tmp.cap = caps[0];
        PCR_TRY(launch_knn<KNN_MODE_SOR>(ctx, &tmp, kb.a[0]));
    } else {
        if (nb_neighbors > 32) { ctx->err = "batched SOR: nb_neighbors must be <= 32"; return PCR_EINVAL; }
        PCR_TRY(launch_knn_batch<KNN_MODE_SOR>(ctx, kb.a, caps, m));
    }
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, SorStatBatch, k_sor_stats_batch, k_sor_stats_batchp, sb.a, m, dim3(SOR_STAT_BLOCKS, m), dim3(256)));
    PCR_TRY(flag_scan_batch(ctx, cb.a, m));
    PCR_TRY(PCR_BATCH_LAUNCH(ctx, CompactBatch, k_compact_cloud_batch, k_compact_cloud_batchp, mb.a, m, dim3((mc + BS - 1) / BS, m), dim3(BS)));
    if (any_todo) {
        if (fuse_all) PCR_TRY(PCR_BATCH_LAUNCH(ctx, NflBatch, k_normals_from_lists_batch, k_normals_from_lists_batchp, nb_.a, m, dim3((unsigned)(((size_t)mc + KNN_BS - 1) / KNN_BS), m), dim3(KNN_BS)));
        else for (int k = 0; k < m; k++) if (nb_.a[k].pts) PCR_LAUNCH(ctx, k_normals_from_lists, dim3((unsigned)(((size_t)caps[k] + KNN_BS - 1) / KNN_BS)), dim3(KNN_BS), 0, ctx->stream, nb_.a[k]);
        if (fallback_here && normal_k > 0) {
            bool listed = fuse_all && piece_counts != nullptr;
            for (int k = 0; k < m; k++) listed = listed && fb.a[k].hard_list != nullptr;
            if (listed) {       // every problem's todo points come as a list of pieces: the octet kernel's list form, a fixed small grid striding over them
                const dim3 grid(knn_list_grid(mc), m), block(KNN_BS);
                PCR_TRY(PCR_BATCH_LAUNCH(ctx, KnnBatch, (k_knn_list_batch<KNN_MODE_NORMALS, 4>), (k_knn_list_batchp<KNN_MODE_NORMALS, 4>), fb.a, m, grid, block));
            } else if (normal_k <= 32) PCR_TRY(launch_knn_batch<KNN_MODE_NORMALS>(ctx, fb.a, caps, m));
            else for (int k = 0; k < m; k++) { DevCloud tmp; tmp.cap = caps[k]; PCR_TRY(launch_knn<KNN_MODE_NORMALS>(ctx, &tmp, fb.a[k])); }
        }
    }
    return PCR_OK;
}

int pcr_dev_sor(pcr_context *ctx, const DevCloud *in, int nb_neighbors, double std_ratio, DevCloud *out, uint8_t *keep_sorted, double *avg_sorted,
                int normal_k, const float4 *prior_out, uint8_t *todo_out, int *todo_count, int *cnt_in_out, int *cnt_kept_out, bool fallback_here) {
    ArenaMark mark(ctx);
    SorProblem p{in, out, keep_sorted, avg_sorted, prior_out, todo_out, todo_count, cnt_in_out, cnt_kept_out};
    return sor_batch(ctx, &p, 1, nb_neighbors, std_ratio, normal_k, fallback_here);
}
// the SOR chains of `count` clouds (the voxel clouds of all scales) in one set of launches; scratch above the caller's mark
int pcr_dev_sor_batch(pcr_context *ctx, const DevCloud *const *ins, DevCloud *const *outs, int count, int nb_neighbors, double std_ratio, int normal_k,
                      const float4 *const *priors, uint8_t *const *todos, int *const *todo_counts, int *const *cnt_in, int *const *cnt_kept, bool fallback_here) {
    if (count > PCR_MAX_GROUP_BATCH) { ctx->err = "SOR batch size"; return PCR_EINVAL; }
    std::vector<SorProblem> pv((size_t)(count > 0 ? count : 1)); SorProblem *p = pv.data();
    for (int k = 0; k < count; k++) p[k] = SorProblem{ins[k], outs[k], nullptr, nullptr, priors ? priors[k] : nullptr, todos[k], todo_counts[k], cnt_in ? cnt_in[k] : nullptr, cnt_kept ? cnt_kept[k] : nullptr};
    return sor_batch(ctx, p, count, nb_neighbors, std_ratio, normal_k, fallback_here);
}
// k-NN normals of `count` clouds over their own trees in one launch (todo masks optional): the incomplete lists of the cleaned targets
int pcr_dev_normals_batch(pcr_context *ctx, DevCloud *const *cs, int count, int search_kind, int knn, double radius, const float4 *const *priors, float4 *const *normals_out,
                          const uint8_t *const *todos) {
    if (search_kind != PCR_SEARCH_KNN && search_kind != PCR_SEARCH_HYBRID) { ctx->err = "normals batch: KNN or Hybrid search"; return PCR_EINVAL; }
    if (search_kind == PCR_SEARCH_HYBRID && !(radius > 0)) { ctx->err = "radius <= 0"; return PCR_EINVAL; }
    if (knn < 1) { ctx->err = "knn < 1"; return PCR_EINVAL; }
    if (count > PCR_MAX_GROUP_BATCH) { ctx->err = "normals batch size"; return PCR_EINVAL; }
    std::vector<KnnArgs> bv((size_t)(count > 0 ? count : 1)); std::memset(bv.data(), 0, sizeof(KnnArgs) * bv.size());
    struct { KnnArgs *a; } b{bv.data()}; std::vector<int> capsv((size_t)(count > 0 ? count : 1)); int *caps = capsv.data(); int m = 0;
    for (int k = 0; k < count; k++) {
        if (cs[k]->cap <= 0) continue;
        KnnArgs &a = b.a[m];
        a.t = oct_view(cs[k]); a.n_ptr = cs[k]->n; a.k = knn; a.prior = priors ? priors[k] : nullptr; a.normals = normals_out[k]; a.todo = todos ? todos[k] : nullptr; a.seed_span = -1;
        knn_radius(a, search_kind, radius);
        caps[m++] = cs[k]->cap;
    }
    if (m == 0) return PCR_OK;
    if (knn <= 32) return launch_knn_batch<KNN_MODE_NORMALS>(ctx, b.a, caps, m);
    for (int k = 0; k < m; k++) { DevCloud tmp; tmp.cap = caps[k]; PCR_TRY(launch_knn<KNN_MODE_NORMALS>(ctx, &tmp, b.a[k])); }
    return PCR_OK;
}
// the k-best lists of `count` clouds in one launch of the octet kernel (the FPFH neighbour lists of a lockstep FGR group): per cloud
// exactly pcr_dev_knn_debug
int pcr_dev_knn_lists_batch(pcr_context *ctx, const DevCloud *const *cs, int count, int k, double radius, int32_t *const *idx, float *const *d2) {
    if (count < 1) return PCR_OK;
    if (count > PCR_MAX_GROUP_BATCH) { ctx->err = "k-NN list batch size"; return PCR_EINVAL; }
    std::vector<KnnArgs> av((size_t)count); std::memset(av.data(), 0, sizeof(KnnArgs) * av.size());
    std::vector<int> caps((size_t)count); int m = 0;
    for (int c = 0; c < count; c++) {
        if (cs[c]->cap <= 0) continue;
        KnnArgs &a = av[m];
        a.t = oct_view(cs[c]); a.n_ptr = cs[c]->n; a.k = k;
        knn_radius(a, radius > 0 ? PCR_SEARCH_HYBRID : PCR_SEARCH_KNN, radius);
        if (radius > 0) a.r2cap_f = (float)(radius * radius);
        a.dbg_idx = idx[c]; a.dbg_d2 = d2[c]; a.dbg_cnt = nullptr; a.dbg_visits = 0;
        caps[m++] = cs[c]->cap;
    }
    if (m == 0) return PCR_OK;
    return launch_knn_octet_batch<KNN_MODE_DEBUG>(ctx, av.data(), caps.data(), m);
}
int pcr_dev_flag_scan_batch(pcr_context *ctx, int count, uint8_t *const *flags, const int *const *n_ptr, const int *n_cap, int *const *pos, int *const *total_dev) {
    if (count < 1) return PCR_OK;
    std::vector<ScanArgs> a((size_t)count);
    for (int c = 0; c < count; c++) PCR_TRY(scan_args(ctx, &a[c], flags[c], n_ptr ? n_ptr[c] : nullptr, n_cap[c], pos[c], total_dev[c], FlagSrc{nullptr, nullptr, nullptr}));
    return flag_scan_batch(ctx, a.data(), count);
}

// ================================================================== covariances / normals (K5)
int pcr_dev_normals(pcr_context *ctx, DevCloud *c, int search_kind, int knn, double radius, const float4 *prior, float4 *normals_out, float *cov6_out,
                    const uint8_t *todo) {
    if (search_kind == PCR_SEARCH_RADIUS) {
        if (!(radius > 0)) { ctx->err = "radius <= 0"; return PCR_EINVAL; }
        if (c->cap <= 0) return PCR_OK;
        RadArgs r;
        r.t = oct_view(c); r.n_ptr = c->n; r.r2 = radius * radius; r.r2f = (float)(r.r2 * (1.0 + 1e-6));
        r.prior = prior; r.normals = normals_out; r.cov6 = cov6_out;
        PCR_LAUNCH(ctx, k_radius_moments, dim3((unsigned)(((size_t)c->cap * OCT + KNN_BS - 1) / KNN_BS)), dim3(KNN_BS), 0, ctx->stream, r);
        return PCR_OK;
    }
    if (knn < 1) { ctx->err = "knn < 1"; return PCR_EINVAL; }
    if (search_kind == PCR_SEARCH_HYBRID && !(radius > 0)) { ctx->err = "radius <= 0"; return PCR_EINVAL; }
    KnnArgs a = {};
    a.t = oct_view(c); a.n_ptr = c->n; a.k = knn; a.prior = prior; a.normals = normals_out; a.cov6 = cov6_out; a.todo = todo;
    knn_radius(a, search_kind, radius);
    return launch_knn<KNN_MODE_NORMALS>(ctx, c, a);
}

size_t pcr_scratch_bytes_for(int64_t n) {
    // voxel/sort temporaries (2x u64 keys, 2x u32 vals, flags, pos, sort temp) + clouds + boxes, with slack
    // (+ 256 B per point: the k-best rows of a search with k <= 64, pcr_knn_wave.h)
    // (+ 84 B per point: the cell hash of a GICP target -- 16 B x next_pow2(2 n) slots, pcr_dev_build_grid_batch -- and the pending-query lists of
    // the streaming iteration, 20 B per source point; they used to fit only because the filter's scratch had been released by then)
    // (+ 16 B per point: the list certificates of the cell-hash GICP searches, int4 per source point)
    return (size_t)(n > 0 ? n : 1) * 676 + pcr_sort_temp_bytes((size_t)(n > 0 ? n : 1)) + (4u << 20);
}

