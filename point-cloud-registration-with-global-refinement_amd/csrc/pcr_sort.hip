// pcr_sort.hip -- stable LSD radix sort of (uint64 key, uint32 value) pairs, hand-written for gfx950 (no library primitive).
// Serves the voxel grids (key = scale index above the Morton code of the voxel, ~41 significant bits, 600k keys for the merged pass of
// a 200k-point cloud) and the Morton ordering of raw clouds (48 bits).  Stability is the contract: points of one voxel keep their
// input order, so the float64 voxel sums are bit-identical to the oracle's (DESIGN.md "voxel").
// Reference op: PointCloud.voxel_down_sample as called at ALL_FUNCTIONS.py:293-294 / 2_MGICP_refinement_in_NCLT_dataset.py:146-147.
//
// 8-bit digits, ceil(end_bit / 8) passes, three launches per pass and no memsets:
//   k_rs_count    per-tile histogram of the pass's digit (LDS atomics; tile = 4 wavefronts x 16 rows x 64 lanes = 4096 keys)
//   k_rs_scan     one workgroup: for every digit the exclusive prefix over the tiles, on top of the digit's global base
//   k_rs_scatter  ranks inside the tile WITHOUT sorting it: a wavefront owns 1024 consecutive keys; the lanes of a 64-key row that
//                 share a digit find each other with 8 ballots (one per digit bit), the lowest of them bumps the wavefront's running
//                 count of the digit in LDS, and rank = count before the row + earlier lanes of the same digit.  Wavefront w then
//                 adds the counts of wavefronts 0..w-1 and the tile's base: stable by construction (tile, wavefront, row, lane order).
// A pass moves 36 bytes per key (read twice, written once); the kernels are latency-bound at these sizes (6 passes over 600k keys:
// ~0.2 ms alone), which is why they are few and fat rather than many and tuned.
#include <cstring>
#include <cstdlib>
#include <vector>
#include "pcr_internal.h"

#define RS_WAVES 4
#define RS_ROWS 16
#define RS_TILE (RS_WAVES * RS_ROWS * 64)

__global__ void __launch_bounds__(RS_WAVES * 64) k_rs_count(const uint64_t *__restrict__ keys, int n, int shift, int *__restrict__ tile_hist) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_WAVES * RS_ROWS; r += RS_WAVES) {
        const int i = base + (r + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
        if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1);
    }
    __syncthreads();
    tile_hist[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// tile_off[t][d] = (number of keys with a smaller digit) + (keys with digit d in tiles before t)
__global__ void __launch_bounds__(1024) k_rs_scan(const int *__restrict__ tile_hist, int n_tiles, int *__restrict__ tile_off) {
    __shared__ int part[4][256];
    __shared__ int basev[256];
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    const int per = (n_tiles + 3) / 4, t0 = q * per, t1 = min(n_tiles, t0 + per);
    int s = 0;
    for (int t = t0; t < t1; t += 8) {              // eight loads in flight (one after the other this walk is pure latency: 340 us at 490 tiles, 2M keys)
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = t + u < t1 ? tile_hist[(t + u) * 256 + d] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) s += c[u];
    }
    part[q][d] = s;
    __syncthreads();
    if (q == 0) basev[d] = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    __syncthreads();
    if (threadIdx.x < 64) {      // exclusive scan of the 256 digit totals by one wavefront: 4 digits per lane
        const int l = threadIdx.x;
        const int v0 = basev[4 * l], v1 = basev[4 * l + 1], v2 = basev[4 * l + 2], v3 = basev[4 * l + 3];
        int inc = v0 + v1 + v2 + v3;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (l >= o) inc += t; }
        const int ex = inc - (v0 + v1 + v2 + v3);
        basev[4 * l] = ex; basev[4 * l + 1] = ex + v0; basev[4 * l + 2] = ex + v0 + v1; basev[4 * l + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    int run = basev[d];
    for (int k = 0; k < q; k++) run += part[k][d];
    for (int t = t0; t < t1; t += 8) {
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = t + u < t1 ? tile_hist[(t + u) * 256 + d] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) { if (t + u < t1) tile_off[(t + u) * 256 + d] = run; run += c[u]; }
    }
}

__global__ void __launch_bounds__(RS_WAVES * 64) k_rs_scatter(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, int n, int shift,
                                                              const int *__restrict__ tile_off, uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ int cnt[RS_WAVES][256];           // running / final count of every digit per wavefront
    __shared__ int off[RS_WAVES][256];           // where wavefront w's keys of digit d start in the output
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) cnt[k][threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE + w * (RS_ROWS * 64);
    uint64_t key[RS_ROWS]; uint32_t val[RS_ROWS]; int rank[RS_ROWS];
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int i = base + r * 64 + lane;
        const bool live = i < n;
        key[r] = live ? keys_in[i] : ~0ull;
        val[r] = live ? vals_in[i] : 0u;
        const int d = (int)((key[r] >> shift) & 255ull);
        // lanes of this row with the same digit (dead lanes are grouped apart through the ninth ballot)
        unsigned long long same = __ballot(live) ^ (live ? 0ull : ~0ull);
#pragma unroll
        for (int b = 0; b < 8; b++) { const unsigned long long bal = __ballot((d >> b) & 1); same &= ((d >> b) & 1) ? bal : ~bal; }
        const int before = __builtin_popcountll(same & lt);
        int pre = 0;
        if (live) pre = cnt[w][d];                                  // every lane of the group reads the count before the row ...
        if (live && before == 0) cnt[w][d] = pre + __builtin_popcountll(same);      // ... and its lowest lane then adds the group
        rank[r] = pre + before;
    }
    __syncthreads();
    {   // digit d = threadIdx.x: wavefront offsets on top of the tile's base
        int run = tile_off[blockIdx.x * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) { off[k][threadIdx.x] = run; run += cnt[k][threadIdx.x]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int i = base + r * 64 + lane;
        if (i < n) {
            const int o = off[w][(int)((key[r] >> shift) & 255ull)] + rank[r];
            keys_out[o] = key[r]; vals_out[o] = val[r];
        }
    }
}

// ---- several independent sorts through the same launches (blockIdx.y = problem; argument structs in device memory): the voxel
// passes of all clouds of a group of pairs.  Every problem runs the same number of passes (digits above a problem's keys are zero:
// a stable pass on them is the identity).
struct RsArgs { const uint64_t *keys_in; const uint32_t *vals_in; uint64_t *keys_out; uint32_t *vals_out; int *tile_hist; int *tile_off; int n; int tiles; int shift; };
__global__ void __launch_bounds__(RS_WAVES * 64) k_rs_count_g(const RsArgs *a_) {
    const RsArgs &a = a_[blockIdx.y];
    if ((int)blockIdx.x >= a.tiles) return;
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_WAVES * RS_ROWS; r += RS_WAVES) {
        const int i = base + (r + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
        if (i < a.n) atomicAdd(&h[(int)((a.keys_in[i] >> a.shift) & 255ull)], 1);
    }
    __syncthreads();
    a.tile_hist[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}
__global__ void __launch_bounds__(1024) k_rs_scan_g(const RsArgs *a_) {
    const RsArgs &a = a_[blockIdx.y];
    const int *__restrict__ tile_hist = a.tile_hist; int *__restrict__ tile_off = a.tile_off; const int n_tiles = a.tiles;
    __shared__ int part[4][256];
    __shared__ int basev[256];
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    const int per = (n_tiles + 3) / 4, t0 = q * per, t1 = min(n_tiles, t0 + per);
    int s = 0;
    for (int t = t0; t < t1; t += 8) {              // eight loads in flight (one after the other this walk is pure latency: 340 us at 490 tiles, 2M keys)
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = t + u < t1 ? tile_hist[(t + u) * 256 + d] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) s += c[u];
    }
    part[q][d] = s;
    __syncthreads();
    if (q == 0) basev[d] = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        const int v0 = basev[4 * l], v1 = basev[4 * l + 1], v2 = basev[4 * l + 2], v3 = basev[4 * l + 3];
        int inc = v0 + v1 + v2 + v3;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (l >= o) inc += t; }
        const int ex = inc - (v0 + v1 + v2 + v3);
        basev[4 * l] = ex; basev[4 * l + 1] = ex + v0; basev[4 * l + 2] = ex + v0 + v1; basev[4 * l + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    int run = basev[d];
    for (int k = 0; k < q; k++) run += part[k][d];
    for (int t = t0; t < t1; t += 8) {
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = t + u < t1 ? tile_hist[(t + u) * 256 + d] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) { if (t + u < t1) tile_off[(t + u) * 256 + d] = run; run += c[u]; }
    }
}
__global__ void __launch_bounds__(RS_WAVES * 64) k_rs_scatter_g(const RsArgs *a_) {
    const RsArgs &a = a_[blockIdx.y];
    if ((int)blockIdx.x >= a.tiles) return;
    const uint64_t *__restrict__ keys_in = a.keys_in; const uint32_t *__restrict__ vals_in = a.vals_in; const int n = a.n, shift = a.shift;
    __shared__ int cnt[RS_WAVES][256];
    __shared__ int off[RS_WAVES][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) cnt[k][threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE + w * (RS_ROWS * 64);
    uint64_t key[RS_ROWS]; uint32_t val[RS_ROWS]; int rank[RS_ROWS];
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int i = base + r * 64 + lane;
        const bool live = i < n;
        key[r] = live ? keys_in[i] : ~0ull;
        val[r] = live ? vals_in[i] : 0u;
        const int d = (int)((key[r] >> shift) & 255ull);
        unsigned long long same = __ballot(live) ^ (live ? 0ull : ~0ull);
#pragma unroll
        for (int b = 0; b < 8; b++) { const unsigned long long bal = __ballot((d >> b) & 1); same &= ((d >> b) & 1) ? bal : ~bal; }
        const int before = __builtin_popcountll(same & lt);
        int pre = 0;
        if (live) pre = cnt[w][d];
        if (live && before == 0) cnt[w][d] = pre + __builtin_popcountll(same);
        rank[r] = pre + before;
    }
    __syncthreads();
    {
        int run = a.tile_off[blockIdx.x * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) { off[k][threadIdx.x] = run; run += cnt[k][threadIdx.x]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int i = base + r * 64 + lane;
        if (i < n) {
            const int o = off[w][(int)((key[r] >> shift) & 255ull)] + rank[r];
            a.keys_out[o] = key[r]; a.vals_out[o] = val[r];
        }
    }
}

static inline size_t rs_tiles(size_t n) { return (n + RS_TILE - 1) / RS_TILE; }

size_t pcr_sort_temp_bytes(size_t n) {
    // ping buffer for keys and values + two tile tables
    return (n + 64) * (sizeof(uint64_t) + sizeof(uint32_t)) + 2 * (rs_tiles(n) + 1) * 256 * sizeof(int) + 1024;
}

int pcr_sort_pairs(pcr_context *ctx, void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const uint32_t *vals_in, uint32_t *vals_out, size_t n, int end_bit) {
    if (n == 0) return PCR_OK;
    if (n > 0x7fffffffull) { ctx->err = "sort: too many keys"; return PCR_EINVAL; }
    if (temp_bytes < pcr_sort_temp_bytes(n)) { ctx->err = "sort: temporary storage too small"; return PCR_ENOMEM; }
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    const int passes = (end_bit + 7) / 8;
    const int tiles = (int)rs_tiles(n);
    char *p = (char *)temp;
    uint64_t *keys_tmp = (uint64_t *)p; p += ((n + 64) * sizeof(uint64_t) + 255) & ~(size_t)255;
    uint32_t *vals_tmp = (uint32_t *)p; p += ((n + 64) * sizeof(uint32_t) + 255) & ~(size_t)255;
    int *tile_hist = (int *)p; p += (size_t)(tiles + 1) * 256 * sizeof(int);
    int *tile_off = (int *)p;
    const uint64_t *ki = keys_in; const uint32_t *vi = vals_in;
    for (int ps = 0; ps < passes; ps++) {
        // the last pass must write the caller's output buffers: alternate so that pass `passes - 1` lands there
        const bool to_out = ((passes - 1 - ps) & 1) == 0;
        uint64_t *ko = to_out ? keys_out : keys_tmp; uint32_t *vo = to_out ? vals_out : vals_tmp;
        const int shift = 8 * ps;
        PCR_LAUNCH(ctx, k_rs_count, dim3(tiles), dim3(RS_WAVES * 64), 0, ctx->stream, ki, (int)n, shift, tile_hist);
        PCR_LAUNCH(ctx, k_rs_scan, dim3(1), dim3(1024), 0, ctx->stream, tile_hist, tiles, tile_off);
        PCR_LAUNCH(ctx, k_rs_scatter, dim3(tiles), dim3(RS_WAVES * 64), 0, ctx->stream, ki, vi, (int)n, shift, tile_off, ko, vo);
        ki = ko; vi = vo;
    }
    return PCR_OK;
}

// `count` independent sorts in 3 launches per digit; temps[k] holds pcr_sort_temp_bytes(n[k]) bytes; end_bit = the widest key
int pcr_sort_pairs_batch(pcr_context *ctx, int count, void *const *temps, const uint64_t *const *keys_in, uint64_t *const *keys_out,
                         const uint32_t *const *vals_in, uint32_t *const *vals_out, const size_t *n, int end_bit) {
    if (count < 1) return PCR_OK;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    const int passes = (end_bit + 7) / 8;
    std::vector<RsArgs> a((size_t)count);
    std::vector<uint64_t *> ktmp((size_t)count); std::vector<uint32_t *> vtmp((size_t)count);
    int max_tiles = 0;
    for (int k = 0; k < count; k++) {
        if (n[k] > 0x7fffffffull) { ctx->err = "sort: too many keys"; return PCR_EINVAL; }
        const int tiles = (int)rs_tiles(n[k]);
        char *p = (char *)temps[k];
        ktmp[k] = (uint64_t *)p; p += ((n[k] + 64) * sizeof(uint64_t) + 255) & ~(size_t)255;
        vtmp[k] = (uint32_t *)p; p += ((n[k] + 64) * sizeof(uint32_t) + 255) & ~(size_t)255;
        a[k].tile_hist = (int *)p; p += (size_t)(tiles + 1) * 256 * sizeof(int);
        a[k].tile_off = (int *)p;
        a[k].n = (int)n[k]; a[k].tiles = tiles;
        a[k].keys_in = keys_in[k]; a[k].vals_in = vals_in[k];
        max_tiles = tiles > max_tiles ? tiles : max_tiles;
    }
    if (max_tiles == 0) return PCR_OK;
    for (int ps = 0; ps < passes; ps++) {
        const bool to_out = ((passes - 1 - ps) & 1) == 0;
        for (int k = 0; k < count; k++) { a[k].keys_out = to_out ? keys_out[k] : ktmp[k]; a[k].vals_out = to_out ? vals_out[k] : vtmp[k]; a[k].shift = 8 * ps; }
        const RsArgs *d = pcr_desc_upload(ctx, a.data(), count);
        if (!d) return PCR_ENOMEM;
        PCR_LAUNCH(ctx, k_rs_count_g, dim3(max_tiles, count), dim3(RS_WAVES * 64), 0, ctx->stream, d);
        PCR_LAUNCH(ctx, k_rs_scan_g, dim3(1, count), dim3(1024), 0, ctx->stream, d);
        PCR_LAUNCH(ctx, k_rs_scatter_g, dim3(max_tiles, count), dim3(RS_WAVES * 64), 0, ctx->stream, d);
        for (int k = 0; k < count; k++) { a[k].keys_in = a[k].keys_out; a[k].vals_in = a[k].vals_out; }
    }
    return PCR_OK;
}
