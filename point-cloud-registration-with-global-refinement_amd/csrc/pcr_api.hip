// pcr_api.hip -- the extern "C" boundary of libpcr_hip.so (declared in include/pcr_hip.h) plus the context /
// scratch-arena plumbing.  Each entry point replaces one Open3D binding call of the reference; see the header.
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <atomic>
#include <mutex>
#include <thread>
#include <chrono>
#include <vector>
#include "pcr_octree.h"

// ------------------------------------------------------------------------------------------- context
extern "C" int pcr_version(void) { return 220; }

PcrOptions &pcr_options() {
    static PcrOptions o;
    static std::once_flag once;
    std::call_once(once, []() {
        if (const char *e = getenv("PCR_KNN_WAVE")) o.knn_wave = atoi(e);
        if (const char *e = getenv("PCR_KNNW_BUDGET")) o.knnw_budget = atoi(e);
        if (const char *e = getenv("PCR_FENCE_PREP")) o.fence_prep = atoi(e);
        if (const char *e = getenv("PCR_ICP_PHASE")) o.icp_phase = atoi(e);
        if (getenv("PCR_ICP_VERIFY")) o.icp_verify = 1;
        if (getenv("PCR_DEBUG_STAMPS")) o.debug_stamps = 1;
        if (const char *e = getenv("PCR_DEBUG_VISITS")) o.debug_visits = atoi(e) ? atoi(e) : 1;
        if (const char *e = getenv("PCR_SPFH_FLOAT64")) o.spfh_float64 = atoi(e);
        if (const char *e = getenv("PCR_RADIUS_LIST_SELECT")) o.radius_list_select = atoi(e);
        if (const char *e = getenv("PCR_FEATNN_MUTUAL")) o.featnn_mutual = atoi(e);
        if (const char *e = getenv("PCR_PLAN_STAGGER_US")) o.plan_stagger_us = atoi(e);
        if (const char *e = getenv("PCR_PLAN_PREFETCH")) o.plan_prefetch = atoi(e);
        if (const char *e = getenv("PCR_ICP_SCALES")) o.icp_scales = atoi(e);
    });
    return o;
}
PcrCounters &pcr_counters() { static PcrCounters c; return c; }
extern "C" long long pcr_counter(const char *name, int reset) {
    if (!name) return -1;
    PcrCounters &c = pcr_counters();
    std::atomic<long long> *v = !strcmp(name, "fgr_group_barrier_timeouts") ? &c.fgr_group_barrier_timeouts : !strcmp(name, "fgr_group_pool_overflows") ? &c.fgr_group_pool_overflows
                               : !strcmp(name, "fgr_group_pairs_redone_alone") ? &c.fgr_group_pairs_redone_alone : nullptr;
    if (!v) return -1;
    return reset ? v->exchange(0) : v->load();
}
extern "C" int pcr_set_option(const char *name, long long value) {
    if (!name) return PCR_EINVAL;
    PcrOptions &o = pcr_options();
    if (!strcmp(name, "knn_wave")) { o.knn_wave = (int)value; return PCR_OK; }
    if (!strcmp(name, "knnw_budget")) { o.knnw_budget = (int)value; return PCR_OK; }
    if (!strcmp(name, "fence_prep")) { o.fence_prep = (int)value; return PCR_OK; }
    if (!strcmp(name, "icp_phase")) { o.icp_phase = (int)value; return PCR_OK; }
    if (!strcmp(name, "icp_verify")) { o.icp_verify = (int)value; return PCR_OK; }
    if (!strcmp(name, "debug_stamps")) { o.debug_stamps = (int)value; return PCR_OK; }
    if (!strcmp(name, "debug_visits")) { o.debug_visits = (int)value; return PCR_OK; }
    if (!strcmp(name, "spfh_float64")) { o.spfh_float64 = (int)value; return PCR_OK; }
    if (!strcmp(name, "radius_list_select")) { o.radius_list_select = (int)value; return PCR_OK; }
    if (!strcmp(name, "arena_poison")) { o.arena_poison = (int)value; return PCR_OK; }
    if (!strcmp(name, "featnn_mutual")) { o.featnn_mutual = (int)value; return PCR_OK; }
    if (!strcmp(name, "plan_stagger_us")) { o.plan_stagger_us = (int)value; return PCR_OK; }
    if (!strcmp(name, "plan_prefetch")) { o.plan_prefetch = (int)value; return PCR_OK; }
    if (!strcmp(name, "icp_scales")) { o.icp_scales = (int)value; return PCR_OK; }
    return PCR_EINVAL;
}

extern "C" int pcr_create(int device, pcr_context **out) {
    if (!out) return PCR_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PCR_EHIP;     // no GPU: fail loudly, no CPU fallback
    if (device < 0 || device >= count) return PCR_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return PCR_EHIP;
    pcr_context *ctx = new pcr_context();
    ctx->device = device;
    // Streams are created on first use and only as many as needed: the HIP runtime maps streams onto a small pool of
    // hardware queues (GPU_MAX_HW_QUEUES, default 4) in creation order, and two streams that share a queue run their
    // kernels one after the other -- an unused stream would push a busy one onto a shared queue (measured: 110 vs
    // 180 pairs/s with three pairs in flight).
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&ctx->side_ev[i], hipEventDisableTiming) != hipSuccess) { delete ctx; return PCR_EHIP; }
    for (int i = 0; i < 8; i++)
        if (hipEventCreateWithFlags(&ctx->lane_ev[i], hipEventDisableTiming) != hipSuccess) { delete ctx; return PCR_EHIP; }
    ctx->pinned_cap = 1 << 16;
    if (hipHostMalloc((void **)&ctx->pinned, ctx->pinned_cap, hipHostMallocDefault) != hipSuccess) { delete ctx; return PCR_EHIP; }
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&ctx->ev[i], hipEventDisableTiming) != hipSuccess) { delete ctx; return PCR_EHIP; }
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&ctx->fence_ev[i], hipEventDisableTiming) != hipSuccess) { delete ctx; return PCR_EHIP; }
    ctx->fence_default = true;          // until pcr_set_stream says otherwise the caller is assumed to work on the default stream
    *out = ctx;
    return PCR_OK;
}

extern "C" int pcr_destroy(pcr_context *ctx) {
    if (!ctx) return PCR_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->aux) (void)hipFree(ctx->aux);
    if (ctx->icp_group_dev) (void)hipFree(ctx->icp_group_dev);
    if (ctx->icp_group_host) (void)hipHostFree(ctx->icp_group_host);
    if (ctx->desc_dev) (void)hipFree(ctx->desc_dev);
    if (ctx->desc_host) (void)hipHostFree(ctx->desc_host);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (int i = 0; i < 2; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 2; i++) if (ctx->fence_ev[i]) (void)hipEventDestroy(ctx->fence_ev[i]);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    for (auto &g : ctx->icp_graphs) { (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->side_stream2) { (void)hipStreamSynchronize(ctx->side_stream2); (void)hipStreamDestroy(ctx->side_stream2); }
    for (int i = 0; i < 2; i++) if (ctx->side_ev[i]) (void)hipEventDestroy(ctx->side_ev[i]);
    for (int i = 0; i < 8; i++) if (ctx->lane_ev[i]) (void)hipEventDestroy(ctx->lane_ev[i]);
    delete ctx;
    return PCR_OK;
}

extern "C" int pcr_set_stream(pcr_context *ctx, void *s) {
    if (!ctx) return PCR_EINVAL;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    // NULL = the legacy default stream (what torch.cuda.current_stream().cuda_stream is for torch's default stream): the work runs
    // on the context's own stream, created on first use, fenced against the default stream on both sides of every call
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    ctx->fence_default = (s == nullptr);
    return PCR_OK;
}
// worker contexts of pcr_register_pairs: own stream, no fences (the call orders its workers against `after_stream` itself)
static void use_private_stream(pcr_context *ctx) {
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->stream = ctx->own_stream;
    ctx->fence_default = false;
}

extern "C" int pcr_profile_enable(pcr_context *ctx, int on) {
    if (!ctx) return PCR_EINVAL;
    ctx->profiling = on ? 1 : 0;
    return PCR_OK;
}
extern "C" int pcr_profile_read(pcr_context *ctx, double *out16, int reset) {
    if (!ctx || !out16) return PCR_EINVAL;
    for (int i = 0; i < 16; i++) { out16[i] = ctx->prof[i]; if (reset) ctx->prof[i] = 0; }
    return PCR_OK;
}

extern "C" const char *pcr_last_error(const pcr_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pcr_arena_reserve(pcr_context *ctx, size_t bytes) {
    ctx->arena_off = 0;
    if (const int poison = pcr_options().arena_poison.load(std::memory_order_relaxed)) {      // diagnostic: a read of scratch nobody wrote shows as a result that follows the pattern
        if (ctx->arena) { PCR_HIP_CHECK(ctx, hipDeviceSynchronize()); PCR_HIP_CHECK(ctx, hipMemset(ctx->arena, poison & 0xff, ctx->arena_cap)); }
    }
    if (bytes <= ctx->arena_cap) return PCR_OK;
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) { PCR_HIP_CHECK(ctx, hipFree(ctx->arena)); ctx->arena = nullptr; ctx->arena_cap = 0; }
    size_t cap = bytes + bytes / 4;
    hipError_t e = hipMalloc((void **)&ctx->arena, cap);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc(arena): ") + hipGetErrorString(e); return PCR_ENOMEM; }
    ctx->arena_cap = cap;
    return PCR_OK;
}

void *pcr_arena_alloc(pcr_context *ctx, size_t bytes) {
    size_t off = (ctx->arena_off + 255) & ~(size_t)255;
    if (off + bytes > ctx->arena_cap) { ctx->err = "scratch arena exhausted"; return nullptr; }
    ctx->arena_off = off + bytes;
    return ctx->arena + off;
}

static int ensure_stream(pcr_context *ctx) {
    if (ctx->stream) return PCR_OK;
    if (!ctx->own_stream && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return PCR_EHIP; }
    ctx->stream = ctx->own_stream;
    return PCR_OK;
}
int pcr_ensure_lanes(pcr_context *ctx, int lanes) {
    if (!ctx->side_stream && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return PCR_EHIP; }
    if (lanes > 1 && !ctx->side_stream2 && hipStreamCreateWithFlags(&ctx->side_stream2, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return PCR_EHIP; }
    return PCR_OK;
}
int pcr_enter(pcr_context *ctx) {
    if (!ctx) return PCR_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return PCR_EHIP;
    ctx->err.clear();
    ctx->launch_err = hipSuccess;
    (void)hipGetLastError();                                  // errors of earlier, unrelated calls on this thread are not ours
    if (ensure_stream(ctx) != PCR_OK) return PCR_EHIP;
    if (ctx->fence_default) {                                 // ordered after everything the caller has enqueued on the default stream
        PCR_HIP_CHECK(ctx, hipEventRecord(ctx->fence_ev[0], (hipStream_t)0));
        PCR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->fence_ev[0], 0));
    }
    return PCR_OK;
}
int pcr_leave(pcr_context *ctx, int rc) {
    if (ctx->launch_err != hipSuccess && rc == PCR_OK) {
        ctx->err = std::string("kernel launch failed at ") + (ctx->launch_file ? ctx->launch_file : "?") + ":" + std::to_string(ctx->launch_line) + ": " + hipGetErrorString(ctx->launch_err);
        rc = PCR_EHIP;
    }
    if (ctx->fence_default && ctx->stream) {                  // what the caller enqueues on the default stream next sees the results
        if (hipEventRecord(ctx->fence_ev[1], ctx->stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)0, ctx->fence_ev[1], 0) != hipSuccess) {
            if (rc == PCR_OK) { ctx->err = "default-stream fence failed"; rc = PCR_EHIP; }
        }
    }
    return rc;
}

int pcr_alloc_cloud(pcr_context *ctx, DevCloud *c, int cap, bool with_nrm, bool with_tree) {
    const int cc = cap > 0 ? cap : 1;
    c->cap = cap;
    c->pts = arena<float4>(ctx, cc);
    c->nrm = with_nrm ? arena<float4>(ctx, cc) : nullptr;
    c->n = arena<int>(ctx, 1);
    c->keys = arena<uint64_t>(ctx, cc);
    if (!c->pts || !c->n || !c->keys || (with_nrm && !c->nrm)) return PCR_ENOMEM;
    c->oct_child = nullptr; c->oct_nodes = nullptr; c->oct_up = nullptr; c->oct_meta = nullptr; c->leaf_of = nullptr; c->pinfo = nullptr; c->oct_l1 = nullptr;
    if (with_tree) {
        const size_t nodes = oct_node_capacity(cc);
        c->oct_child = arena<int>(ctx, nodes);
        c->oct_nodes = arena<float4>(ctx, 2 * nodes);
        c->oct_up = arena<int4>(ctx, nodes);
        c->oct_meta = arena<OctMeta>(ctx, 1);
        c->leaf_of = arena<int>(ctx, cc);
        c->pinfo = arena<int4>(ctx, cc);
        c->oct_l1 = arena<int2>(ctx, cc);
        if (!c->oct_child || !c->oct_nodes || !c->oct_up || !c->oct_meta || !c->leaf_of || !c->pinfo || !c->oct_l1) return PCR_ENOMEM;
    }
    return PCR_OK;
}

// Morton-sorted copy of a caller cloud (+ optional normals) with its BVH; perm maps sorted -> caller index
int pcr_import_cloud(pcr_context *ctx, const float *xyz, const float *nrm, int64_t n, DevCloud *c, uint32_t **perm_out, bool force_nrm) {
    double b6[6];
    PCR_TRY(pcr_dev_bounds(ctx, xyz, n, b6));
    PCR_TRY(pcr_alloc_cloud(ctx, c, (int)n, nrm != nullptr || force_nrm, true));
    uint32_t *perm = arena<uint32_t>(ctx, n > 0 ? n : 1);
    if (!perm) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_sort_cloud(ctx, xyz, n, b6, c, perm));
    if (nrm) PCR_TRY(pcr_dev_gather_f3_to_f4(ctx, nrm, perm, n, c->nrm));
    PCR_TRY(pcr_dev_build_bvh(ctx, c));
    if (perm_out) *perm_out = perm;
    return PCR_OK;
}

// -------------------------------------------------------------------------------------- geometry API
extern "C" int pcr_bounds(pcr_context *ctx, const float *xyz, int64_t n, double *b6) {
    return pcr_api_call(ctx, [&]() -> int {
    if (n < 0 || !b6 || (n > 0 && !xyz)) return PCR_EINVAL;
    PCR_TRY(pcr_arena_reserve(ctx, 1 << 20));
    return pcr_dev_bounds(ctx, xyz, n, b6);
    });
}

extern "C" int pcr_voxel_down_sample(pcr_context *ctx, const float *xyz, const float *normals_in, int64_t n, double voxel,
                                     float *out_xyz, float *out_normals, int64_t *out_n) {
    return pcr_api_call(ctx, [&]() -> int {
    if (n < 0 || !out_n || (n > 0 && (!xyz || !out_xyz))) return PCR_EINVAL;
    if (!(voxel > 0.0)) { ctx->err = "voxel_size <= 0"; return PCR_EINVAL; }
    *out_n = 0;
    if (n == 0) return PCR_OK;
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n)));
    double b6[6];
    PCR_TRY(pcr_dev_bounds(ctx, xyz, n, b6));
    DevCloud v;
    PCR_TRY(pcr_alloc_cloud(ctx, &v, (int)n, normals_in != nullptr, false));
    PCR_TRY(pcr_dev_voxel(ctx, xyz, normals_in, n, b6, voxel, &v));
    PCR_TRY(pcr_dev_pack_f4_to_f3(ctx, v.pts, v.n, v.cap, out_xyz));
    if (normals_in && out_normals) PCR_TRY(pcr_dev_pack_f4_to_f3(ctx, v.nrm, v.n, v.cap, out_normals));
    return pcr_read_count(ctx, v.n, out_n);
    });
}

__global__ void k_keep_to_caller(const uint8_t *__restrict__ keep_sorted, const uint32_t *__restrict__ perm, int n, uint8_t *__restrict__ keep_caller) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keep_caller[perm[i]] = keep_sorted[i];
}
__global__ void k_emit_kept(const float *__restrict__ xyz, const uint8_t *__restrict__ keep, const int *__restrict__ pos, int n, float *__restrict__ out_xyz, int64_t *__restrict__ out_index) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const int o = pos[i];
    if (out_xyz) { out_xyz[o * 3] = xyz[i * 3]; out_xyz[o * 3 + 1] = xyz[i * 3 + 1]; out_xyz[o * 3 + 2] = xyz[i * 3 + 2]; }
    if (out_index) out_index[o] = i;
}

extern "C" int pcr_remove_statistical_outlier(pcr_context *ctx, const float *xyz, int64_t n, int nb_neighbors, double std_ratio,
                                              uint8_t *keep_mask, float *out_xyz, int64_t *out_index, int64_t *out_n) {
    return pcr_api_call(ctx, [&]() -> int {
    if (n < 0 || (n > 0 && !xyz)) return PCR_EINVAL;
    if (nb_neighbors < 1 || !(std_ratio > 0.0)) { ctx->err = "nb_neighbors < 1 or std_ratio <= 0"; return PCR_EINVAL; }
    if (out_n) *out_n = 0;
    if (n == 0) return PCR_OK;
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n) + (size_t)n * 64));
    DevCloud c, kept; uint32_t *perm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, xyz, nullptr, n, &c, &perm, false));
    PCR_TRY(pcr_alloc_cloud(ctx, &kept, (int)n, false, false));
    uint8_t *keep_sorted = arena<uint8_t>(ctx, n);
    uint8_t *keep_caller = keep_mask ? keep_mask : arena<uint8_t>(ctx, n);
    int *pos = arena<int>(ctx, n);
    int *total = arena<int>(ctx, 1);
    if (!keep_sorted || !keep_caller || !pos || !total) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_sor(ctx, &c, nb_neighbors, std_ratio, &kept, keep_sorted, nullptr));
    const int nb = (int)((n + 255) / 256);
    PCR_LAUNCH(ctx, k_keep_to_caller, dim3(nb), dim3(256), 0, ctx->stream, keep_sorted, perm, (int)n, keep_caller);
    // emit the kept points in CALLER order (select_by_index semantics)
    PCR_TRY(pcr_dev_flag_scan(ctx, keep_caller, nullptr, (int)n, pos, total));
    if (out_xyz || out_index) PCR_LAUNCH(ctx, k_emit_kept, dim3(nb), dim3(256), 0, ctx->stream, xyz, keep_caller, pos, (int)n, out_xyz, out_index);
    int64_t m = 0;
    PCR_TRY(pcr_read_count(ctx, total, &m));
    if (out_n) *out_n = m;
    return PCR_OK;
    });
}

extern "C" int pcr_estimate_normals(pcr_context *ctx, const float *xyz, int64_t n, int search_kind, int knn, double radius,
                                    const float *prior_normals, float *normals) {
    return pcr_api_call(ctx, [&]() -> int {
    if (n < 0 || (n > 0 && (!xyz || !normals))) return PCR_EINVAL;
    if (n == 0) return PCR_OK;
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n) + (size_t)n * 64));
    DevCloud c; uint32_t *perm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, xyz, prior_normals, n, &c, &perm, false));
    float4 *nout = arena<float4>(ctx, n);
    if (!nout) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_normals(ctx, &c, search_kind, knn, radius, c.nrm, nout, nullptr));
    PCR_TRY(pcr_dev_scatter_rows_f4_to_f3(ctx, nout, perm, c.n, c.cap, normals));
    return PCR_OK;                      // no scalar output: asynchronous on the context's stream
    });
}

__global__ void k_scatter_cov6(const float *__restrict__ src, const uint32_t *__restrict__ perm, int n, float *__restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = perm[i];
    for (int t = 0; t < 6; t++) dst[(size_t)v * 6 + t] = src[(size_t)i * 6 + t];
}

extern "C" int pcr_estimate_covariances(pcr_context *ctx, const float *xyz, int64_t n, int search_kind, int knn, double radius, float *cov6) {
    return pcr_api_call(ctx, [&]() -> int {
    if (n < 0 || (n > 0 && (!xyz || !cov6))) return PCR_EINVAL;
    if (n == 0) return PCR_OK;
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n) + (size_t)n * 64));
    DevCloud c; uint32_t *perm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, xyz, nullptr, n, &c, &perm, false));
    float *cs = arena<float>(ctx, (size_t)n * 6);
    if (!cs) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_normals(ctx, &c, search_kind, knn, radius, nullptr, nullptr, cs));
    PCR_LAUNCH(ctx, k_scatter_cov6, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cs, perm, (int)n, cov6);
    return PCR_OK;                      // asynchronous on the context's stream
    });
}

__global__ void k_knn_unpermute(const int32_t *si, const float *sd, const int32_t *sc, const uint32_t *perm, int n, int k, int32_t *idx, float *d2, int32_t *counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = perm[i];
    for (int t = 0; t < k; t++) {
        const int32_t v = si[(size_t)i * k + t];
        idx[(size_t)o * k + t] = v >= 0 ? (int32_t)perm[v] : -1;
        d2[(size_t)o * k + t] = sd[(size_t)i * k + t];
    }
    if (counts) counts[o] = sc[i];
}
__global__ void k_match_unpermute(const int32_t *m, const uint32_t *sp, const uint32_t *tp, int n, int32_t *out, int raw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[sp[i]] = raw ? m[i] : (m[i] >= 0 ? (int32_t)tp[m[i]] : -1);
}

extern "C" int pcr_debug_knn(pcr_context *ctx, const float *xyz, int64_t n, int k, double radius, int32_t *idx, float *d2, int32_t *counts) {
    return pcr_api_call(ctx, [&]() -> int {
    if (n <= 0 || !xyz || !idx || !d2) return PCR_EINVAL;
    // sort, search over the BVH, then report rows and indices in the CALLER's point order
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n) + (size_t)n * (size_t)k * 16 + (size_t)n * 64));
    DevCloud c; uint32_t *perm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, xyz, nullptr, n, &c, &perm, false));
    int32_t *si = arena<int32_t>(ctx, (size_t)n * k);
    float *sd = arena<float>(ctx, (size_t)n * k);
    int32_t *sc = arena<int32_t>(ctx, n);
    if (!si || !sd || !sc) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_knn_debug(ctx, &c, k, radius, si, sd, sc));
    PCR_LAUNCH(ctx, k_knn_unpermute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, si, sd, sc, perm, (int)n, k, idx, d2, counts);
    return PCR_OK;
    });
}

// ---------------------------------------------------------------------------------- registration API
static int check_T(pcr_context *ctx, const double *T) {
    if (!T) return PCR_EINVAL;
    for (int k = 0; k < 16; k++) if (!std::isfinite(T[k])) { ctx->err = "non-finite init pose"; return PCR_EINVAL; }
    return PCR_OK;
}

extern "C" int pcr_registration_generalized_icp(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src,
                                                const float *tgt_xyz, const float *tgt_normals, int64_t n_tgt, double max_dist,
                                                const double *init_T, const pcr_gicp_params *params, pcr_result *result,
                                                int32_t *correspondences) {
    return pcr_api_call(ctx, [&]() -> int {
    if (!params || !result || n_src < 0 || n_tgt < 0) return PCR_EINVAL;
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    if ((n_src > 0 && (!src_xyz || !src_normals)) || (n_tgt > 0 && (!tgt_xyz || !tgt_normals))) { ctx->err = "missing cloud or normals"; return PCR_EINVAL; }
    PCR_TRY(check_T(ctx, init_T));
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n_src) + pcr_scratch_bytes_for(n_tgt)));
    DevCloud s, t; uint32_t *sperm = nullptr, *tperm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, src_xyz, src_normals, n_src, &s, &sperm, true));
    PCR_TRY(pcr_import_cloud(ctx, tgt_xyz, tgt_normals, n_tgt, &t, &tperm, true));
    int32_t *match = arena<int32_t>(ctx, n_src > 0 ? n_src : 1);
    if (!match) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_gicp(ctx, &s, &t, max_dist, init_T, params, result, match));
    if (correspondences) {
        int64_t nc = 0;
        PCR_TRY(pcr_dev_compact_matches(ctx, match, s.n, s.cap, sperm, tperm, correspondences, &nc));
    }
    return PCR_OK;
    });
}

__global__ void k_gather_cov6(const float *__restrict__ src, const uint32_t *__restrict__ perm, int n, float *__restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = perm[i];
    for (int t = 0; t < 6; t++) dst[(size_t)i * 6 + t] = src[(size_t)v * 6 + t];
}

extern "C" int pcr_registration_generalized_icp_cov(pcr_context *ctx, const float *src_xyz, const float *src_cov6, int64_t n_src,
                                                    const float *tgt_xyz, const float *tgt_cov6, int64_t n_tgt, double max_dist,
                                                    const double *init_T, const pcr_gicp_params *params, pcr_result *result,
                                                    int32_t *correspondences) {
    return pcr_api_call(ctx, [&]() -> int {
    if (!params || !result || n_src < 0 || n_tgt < 0) return PCR_EINVAL;
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    if ((n_src > 0 && (!src_xyz || !src_cov6)) || (n_tgt > 0 && (!tgt_xyz || !tgt_cov6))) { ctx->err = "missing cloud or covariances"; return PCR_EINVAL; }
    PCR_TRY(check_T(ctx, init_T));
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n_src) + pcr_scratch_bytes_for(n_tgt) + (size_t)(n_src + n_tgt) * 32));
    DevCloud s, t; uint32_t *sperm = nullptr, *tperm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, src_xyz, nullptr, n_src, &s, &sperm, false));
    PCR_TRY(pcr_import_cloud(ctx, tgt_xyz, nullptr, n_tgt, &t, &tperm, false));
    s.cov6 = arena<float>(ctx, (size_t)(n_src > 0 ? n_src : 1) * 6);
    t.cov6 = arena<float>(ctx, (size_t)(n_tgt > 0 ? n_tgt : 1) * 6);
    int32_t *match = arena<int32_t>(ctx, n_src > 0 ? n_src : 1);
    if (!s.cov6 || !t.cov6 || !match) return PCR_ENOMEM;
    if (n_src > 0) PCR_LAUNCH(ctx, k_gather_cov6, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, ctx->stream, src_cov6, sperm, (int)n_src, s.cov6);
    if (n_tgt > 0) PCR_LAUNCH(ctx, k_gather_cov6, dim3((unsigned)((n_tgt + 255) / 256)), dim3(256), 0, ctx->stream, tgt_cov6, tperm, (int)n_tgt, t.cov6);
    PCR_TRY(pcr_dev_gicp(ctx, &s, &t, max_dist, init_T, params, result, match));
    if (correspondences) {
        int64_t nc = 0;
        PCR_TRY(pcr_dev_compact_matches(ctx, match, s.n, s.cap, sperm, tperm, correspondences, &nc));
    }
    return PCR_OK;
    });
}

extern "C" int pcr_debug_gicp_linearize(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src,
                                        const float *tgt_xyz, const float *tgt_normals, int64_t n_tgt, double max_dist,
                                        const double *T, const pcr_gicp_params *params, double *JTJ36, double *JTr6,
                                        double *stats3, int32_t *match_out) {
    return pcr_api_call(ctx, [&]() -> int {
    if (!params || !JTJ36 || !JTr6 || !stats3 || n_src <= 0 || n_tgt <= 0) return PCR_EINVAL;
    PCR_TRY(check_T(ctx, T));
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n_src) + pcr_scratch_bytes_for(n_tgt)));
    DevCloud s, t; uint32_t *sperm = nullptr, *tperm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, src_xyz, src_normals, n_src, &s, &sperm, false));
    PCR_TRY(pcr_import_cloud(ctx, tgt_xyz, tgt_normals, n_tgt, &t, &tperm, false));
    int32_t *match = arena<int32_t>(ctx, n_src);
    if (!match) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_linearize_once(ctx, &s, &t, max_dist, T, params, JTJ36, JTr6, stats3, match));
    if (match_out) {
        PCR_LAUNCH(ctx, k_match_unpermute, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, ctx->stream, match, sperm, tperm, (int)n_src, match_out, pcr_options().debug_visits.load(std::memory_order_relaxed) ? 1 : 0);
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PCR_OK;
    });
}

// one cloud of one scale: voxel -> BVH -> SOR -> BVH -> normals   (ALL_FUNCTIONS.py:293-302)
static int prep_scale(pcr_context *ctx, const float *xyz, const float *nrm, int64_t n, const double *b6, double voxel, int sor_k,
                      double sor_std, int normal_k, DevCloud *clean, int *cnt_voxel_out, int *cnt_clean_out, bool need_tree,
                      const DevCloud *voxel_done = nullptr /* the voxel cloud of this scale when the merged pass has made it */) {
    DevCloud v;
    PCR_TRY(pcr_alloc_cloud(ctx, clean, (int)n, true, need_tree));   // survives the mark below (allocated first)
    float4 *prior = nrm ? arena<float4>(ctx, n > 0 ? n : 1) : nullptr;
    int *nv_keep = arena<int>(ctx, 2);
    uint8_t *todo = arena<uint8_t>(ctx, n > 0 ? n : 1);
    if (!nv_keep || !todo || (nrm && !prior)) return PCR_ENOMEM;
    {
        ArenaMark mark(ctx);
        if (voxel_done) {
            v = *voxel_done;                                         // voxel cloud and its tree made by the merged passes
        } else {
            PCR_TRY(pcr_alloc_cloud(ctx, &v, (int)n, nrm != nullptr, true));
            PCR_TRY(pcr_dev_voxel(ctx, xyz, nrm, n, b6, voxel, &v));
            PCR_TRY(pcr_dev_build_bvh(ctx, &v));
        }
        DevCloud tmp = *clean;
        tmp.nrm = prior;                                         // compacted voxel-mean normals = orientation prior
        tmp.nrm_final = clean->nrm;                              // normals of the cleaned cloud, straight from the SOR lists
        PCR_TRY(pcr_dev_sor(ctx, &v, sor_k, sor_std, &tmp, nullptr, nullptr, normal_k, prior, todo, nv_keep + 1, cnt_voxel_out, cnt_clean_out, !need_tree));
        for (int d = 0; d < 3; d++) { clean->key_org[d] = tmp.key_org[d]; clean->key_unit[d] = tmp.key_unit[d]; }
        clean->voxel_lattice = tmp.voxel_lattice;
    }
    if (need_tree) {     // a GICP target: its tree serves the correspondence search and the few incomplete normal lists
        PCR_TRY(pcr_dev_build_bvh(ctx, clean));
        // (the SOR lists serve the normals only while normal_k <= sor_k <= 32: otherwise every point is searched and no mask is passed)
        PCR_TRY(pcr_dev_normals(ctx, clean, PCR_SEARCH_KNN, normal_k, 0.0, prior, clean->nrm, nullptr, (sor_k <= 32 && normal_k <= sor_k) ? todo : nullptr));
    }
    return PCR_OK;
}

// ---- Multiscale_GICP with the preprocessing of ALL scales of a cloud in one set of batched launches (blockIdx.y = scale):
// voxel grids (one key / sort / scan / mean pass), voxel trees (6 launches), SOR chain (7), cleaned target trees (6), exact fallback
// normals (1): ~33 launches per cloud whatever the number of scales, instead of 13 + 20 per scale.  The two clouds run on their own
// lanes; the GICP loops of the scales follow on the caller's stream.  Returns 1 (nothing enqueued) when the scales cannot share a
// pass (pcr_dev_voxel_multi declines): the caller then takes the scale-by-scale path.
static int multiscale_batched(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src, const float *tgt_xyz, const float *tgt_normals,
                              int64_t n_tgt, const double *voxels, const double *dists, int n_scales, int sor_k, double sor_std, int normal_k, const double *init_T,
                              const pcr_gicp_params *params, pcr_scale_record *records, int32_t *correspondences, const double *bs, const double *bt) {
    if (n_scales < 2 || n_scales > 8 || sor_k > 32 || normal_k > 32 || n_src <= 0 || n_tgt <= 0) return 1;
    const double t_entry = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    auto lane_bytes = [&](int64_t n) { return (size_t)n_scales * ((size_t)n + 512) * 1100 + pcr_sort_temp_bytes((size_t)n * n_scales) + (size_t)n * n_scales * 40 + (32u << 20); };
    const size_t blk_s = lane_bytes(n_src), blk_t = lane_bytes(n_tgt);
    PCR_TRY(pcr_arena_reserve(ctx, blk_s + blk_t + pcr_scratch_bytes_for(n_src) + (1u << 20)));
    char *block_s = (char *)pcr_arena_alloc(ctx, blk_s), *block_t = (char *)pcr_arena_alloc(ctx, blk_t);
    int *cnt4 = arena<int>(ctx, 4 * 8);                  // per scale: voxel counts (source, target), clean counts (source, target)
    if (!block_s || !block_t || !cnt4) return PCR_ENOMEM;
    PCR_TRY(pcr_ensure_lanes(ctx, 2));
    hipStream_t lane_t = ctx->side_stream, lane_s = ctx->side_stream2;
    struct LaneGuard {            // an early error return must not leave lane work running over a recycled arena
        hipStream_t a, b;
        ~LaneGuard() { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); }
    } guard{lane_t, lane_s};
    PCR_HIP_CHECK(ctx, hipEventRecord(ctx->side_ev[0], ctx->stream));       // inputs are ready once the caller's stream gets here
    PCR_HIP_CHECK(ctx, hipStreamWaitEvent(lane_t, ctx->side_ev[0], 0));
    PCR_HIP_CHECK(ctx, hipStreamWaitEvent(lane_s, ctx->side_ev[0], 0));
    DevCloud clean[2][8];
    bool declined = false;
    auto prep_cloud = [&](int which, const float *xyz, const float *nrm, int64_t n, const double *b6, char *block, size_t bytes, hipStream_t lane) -> int {
        SideLane sl(ctx, block, bytes, lane);
        const bool need_tree = which == 1;               // a GICP target: its tree serves the correspondence search
        DevCloud vox[8], tmp[8];
        const DevCloud *ins[8]; DevCloud *outs[8], *trees[8];
        const float4 *priors[8]; uint8_t *todos[8]; int *todo_counts[8], *cnt_in[8], *cnt_kept[8];
        float4 *nouts[8];
        for (int s = 0; s < n_scales; s++) {
            PCR_TRY(pcr_alloc_cloud(ctx, &vox[s], (int)n, nrm != nullptr, true));
            PCR_TRY(pcr_alloc_cloud(ctx, &clean[which][s], (int)n, true, need_tree));
            float4 *prior = nrm ? arena<float4>(ctx, n) : nullptr;
            uint8_t *todo = arena<uint8_t>(ctx, n);
            int *tc = arena<int>(ctx, 2);
            if (!todo || !tc || (nrm && !prior)) return PCR_ENOMEM;
            tmp[s] = clean[which][s];
            tmp[s].nrm = prior;                          // compacted voxel-mean normals = orientation prior
            tmp[s].nrm_final = clean[which][s].nrm;      // normals of the cleaned cloud, straight from the SOR lists
            ins[s] = &vox[s]; outs[s] = &tmp[s]; trees[s] = &clean[which][s];
            priors[s] = prior; todos[s] = todo; todo_counts[s] = tc + 1;
            cnt_in[s] = cnt4 + 4 * s + which; cnt_kept[s] = cnt4 + 4 * s + 2 + which;
            nouts[s] = clean[which][s].nrm;
        }
        bool merged = false;
        PCR_TRY(pcr_dev_voxel_multi(ctx, xyz, nrm, n, b6, voxels, n_scales, vox, &merged));
        if (!merged) { declined = true; return PCR_OK; }
        DevCloud *vp[8];
        for (int s = 0; s < n_scales; s++) vp[s] = &vox[s];
        PCR_TRY(pcr_dev_build_bvh_batch(ctx, vp, n_scales));
        // (round 5: the incomplete normal lists of the TARGETS are searched over the voxel trees too, through the piece lists of the filter pass -- they
        // used to wait for the cleaned trees and run as a full-range launch with a todo mask)
        PCR_TRY(pcr_dev_sor_batch(ctx, ins, outs, n_scales, sor_k, sor_std, normal_k, priors, todos, todo_counts, cnt_in, cnt_kept, true));
        for (int s = 0; s < n_scales; s++) { for (int d = 0; d < 3; d++) { clean[which][s].key_org[d] = tmp[s].key_org[d]; clean[which][s].key_unit[d] = tmp[s].key_unit[d]; } clean[which][s].voxel_lattice = tmp[s].voxel_lattice; }
        if (need_tree) PCR_TRY(pcr_dev_build_bvh_batch(ctx, trees, n_scales));
        PCR_HIP_CHECK(ctx, hipEventRecord(ctx->lane_ev[which], ctx->stream));
        return PCR_OK;
    };
    PCR_TRY(prep_cloud(1, tgt_xyz, tgt_normals, n_tgt, bt, block_t, blk_t, lane_t));
    if (declined) return 1;
    PCR_TRY(prep_cloud(0, src_xyz, src_normals, n_src, bs, block_s, blk_s, lane_s));
    if (declined) { ctx->err = "voxel pass accepted one cloud and declined the other"; return PCR_EINVAL; }      // same scales, same key width: cannot happen
    PCR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_ev[0], 0));
    PCR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_ev[1], 0));
    // diagnostics (PCR_PAIR_TIMELINE=1, with profiling on): host-clock breakdown of a pair: enqueue of the preprocessing, wait for it,
    // the GICP loops of the scales (out16[11..14], seconds; [15] pairs).  The extra synchronisation serialises prep and loop.
    static const bool timeline = getenv("PCR_PAIR_TIMELINE") != nullptr;
    const bool tl = timeline && ctx->profiling;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enq = now();
    if (tl) { PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); }
    const double t_prep = now();
    double T[16];
    memcpy(T, init_T, sizeof T);
    int32_t *match = arena<int32_t>(ctx, n_src);
    if (!match) return PCR_ENOMEM;
    for (int s = 0; s < n_scales; s++) {
        PCR_TRY(pcr_dev_gicp(ctx, &clean[0][s], &clean[1][s], dists[s], T, params, &records[s].icp, match));
        memcpy(T, records[s].icp.transformation, sizeof T);
    }
    if (tl) { ctx->prof[11] += t_enq - t_entry; ctx->prof[12] += t_prep - t_enq; ctx->prof[13] += now() - t_prep; ctx->prof[15] += 1.0; }
    if (correspondences) {
        PCR_TRY(pcr_dev_compact_matches(ctx, match, clean[0][n_scales - 1].n, clean[0][n_scales - 1].cap, nullptr, nullptr, correspondences, nullptr));      // (the count is the result's n_correspondences: no read-back)
    }
    int h[32];
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(h, cnt4, 4 * n_scales * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < n_scales; s++) { records[s].n_voxel[0] = h[4 * s]; records[s].n_voxel[1] = h[4 * s + 1]; records[s].n_clean[0] = h[4 * s + 2]; records[s].n_clean[1] = h[4 * s + 3]; }
    return PCR_OK;
}

// ---- Multiscale_GICP of a GROUP of G pairs through the same launches: bounds (2 launches), voxel grids of all clouds and scales
// (one key / sort / scan / mean pass), voxel trees, SOR chains, cleaned target trees and fallback normals (one batch each, blockIdx.y
// = cloud x scale), then per scale ONE lockstep GICP loop over the G pairs (pcr_dev_gicp_group).  Everything on the context's stream:
// the kernels are G times fatter, so they fill the device by themselves.  Same per-problem arithmetic as the pair-by-pair path:
// bit-identical results.  Returns 1 (nothing enqueued but the bounds) when the merged voxel pass declines: the caller falls back.
static int multiscale_group(pcr_context *ctx, pcr_pair_ex *const *px, int G, const double *voxels, const double *given_dists /* n_scales, radius_rule 0 */, int radius_rule,
                            int n_scales, int sor_k, double sor_std, int normal_k, const pcr_gicp_params *params) {
    if (G < 1 || G > 32 || n_scales < 2 || n_scales > 8 || sor_k > 32 || normal_k > 32) return 1;
    const int C = 2 * G;                                  // cloud c = 2 g + which (0 source, 1 target)
    std::vector<const float *> xyz((size_t)C), nrm((size_t)C); std::vector<int64_t> n((size_t)C);
    bool any_nrm = false;
    size_t total = 0;
    for (int g = 0; g < G; g++) {
        const pcr_pair &p = px[g]->base;
        if (p.n_src <= 0 || p.n_tgt <= 0 || !p.src_xyz || !p.tgt_xyz || !p.records) return 1;
        PCR_TRY(check_T(ctx, p.init_T));
        xyz[2 * g] = p.src_xyz; xyz[2 * g + 1] = p.tgt_xyz; nrm[2 * g] = p.src_normals; nrm[2 * g + 1] = p.tgt_normals;
        n[2 * g] = p.n_src; n[2 * g + 1] = p.n_tgt;
        any_nrm = any_nrm || p.src_normals || p.tgt_normals;
    }
    for (int c = 0; c < C; c++) total += (size_t)n_scales * ((size_t)n[c] + 512) * 1100 + pcr_sort_temp_bytes((size_t)n[c] * n_scales) + (size_t)n[c] * n_scales * 40 + (8u << 20);
    PCR_TRY(pcr_arena_reserve(ctx, total + (16u << 20)));
    std::vector<double> b6((size_t)C * 6);
    PCR_TRY(pcr_dev_bounds_batch(ctx, C, xyz.data(), n.data(), b6.data()));
    std::vector<double> dists((size_t)G * n_scales);
    for (int g = 0; g < G; g++) {
        if (radius_rule == 1) {              // ALL_FUNCTIONS.py:277-278 + 1092-1101: radius_from_cloud_pair * 2^-scale from the two AABBs
            const double *bs = &b6[12 * g], *bt = &b6[12 * g + 6];
            const double r1 = std::pow((bs[3] - bs[0]) * (bs[4] - bs[1]) * (bs[5] - bs[2]), 1.0 / 3.0), r2 = std::pow((bt[3] - bt[0]) * (bt[4] - bt[1]) * (bt[5] - bt[2]), 1.0 / 3.0);
            for (int s = 0; s < n_scales; s++) dists[(size_t)g * n_scales + s] = (r1 + r2) / 2 * std::pow(2.0, -(double)s);
        } else for (int s = 0; s < n_scales; s++) dists[(size_t)g * n_scales + s] = given_dists[s];
        for (int s = 0; s < 8; s++) px[g]->max_distances[s] = s < n_scales ? dists[(size_t)g * n_scales + s] : 0.0;
        for (int s = 0; s < n_scales; s++) if (!(dists[(size_t)g * n_scales + s] > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    }
    int *cnt = arena<int>(ctx, (size_t)C * n_scales * 2);          // [c][s]: voxel count, clean count
    if (!cnt) return PCR_ENOMEM;
    std::vector<DevCloud> vox((size_t)C * n_scales), clean((size_t)C * n_scales), tmp((size_t)C * n_scales);
    std::vector<const float4 *> priors((size_t)C * n_scales); std::vector<uint8_t *> todos((size_t)C * n_scales);
    std::vector<int *> tcs((size_t)C * n_scales), cin((size_t)C * n_scales), ckept((size_t)C * n_scales);
    for (int c = 0; c < C; c++) {
        const bool need_tree = (c & 1) == 1;
        for (int s = 0; s < n_scales; s++) {
            const size_t k = (size_t)c * n_scales + s;
            PCR_TRY(pcr_alloc_cloud(ctx, &vox[k], (int)n[c], nrm[c] != nullptr, true));
            PCR_TRY(pcr_alloc_cloud(ctx, &clean[k], (int)n[c], true, need_tree));
            float4 *prior = nrm[c] ? arena<float4>(ctx, n[c]) : nullptr;
            uint8_t *todo = arena<uint8_t>(ctx, n[c]);
            int *tc = arena<int>(ctx, 2);
            if (!todo || !tc || (nrm[c] && !prior)) return PCR_ENOMEM;
            tmp[k] = clean[k]; tmp[k].nrm = prior; tmp[k].nrm_final = clean[k].nrm;
            priors[k] = prior; todos[k] = todo; tcs[k] = tc + 1; cin[k] = cnt + 2 * k; ckept[k] = cnt + 2 * k + 1;
        }
    }
    bool merged = false;
    PCR_TRY(pcr_dev_voxel_multi_batch(ctx, C, xyz.data(), any_nrm ? nrm.data() : nullptr, n.data(), b6.data(), voxels, n_scales, vox.data(), &merged));
    if (!merged) return 1;
    {
        std::vector<DevCloud *> vp((size_t)C * n_scales);
        for (size_t k = 0; k < vp.size(); k++) vp[k] = &vox[k];
        PCR_TRY(pcr_dev_build_bvh_batch(ctx, vp.data(), (int)vp.size()));
    }
    PcrGateToken heavy_token(ctx->heavy_gate);         // (held to the end of the group: the function returns after its last read-back)
    {   // the filter chains of ALL clouds and scales of the group as one batch (round 5: sources and targets were two batches -- twice the launches, and
        // twice the tails of the list-driven searches); incomplete normal lists are searched over the voxel trees (piece lists of the filter pass)
        std::vector<const DevCloud *> ins; std::vector<DevCloud *> outs; std::vector<const float4 *> pr; std::vector<uint8_t *> td; std::vector<int *> tc, ci, ck;
        for (size_t k = 0; k < (size_t)C * n_scales; k++) {
            ins.push_back(&vox[k]); outs.push_back(&tmp[k]); pr.push_back(priors[k]); td.push_back(todos[k]); tc.push_back(tcs[k]); ci.push_back(cin[k]); ck.push_back(ckept[k]);
        }
        PCR_TRY(pcr_dev_sor_batch(ctx, ins.data(), outs.data(), (int)ins.size(), sor_k, sor_std, normal_k, pr.data(), td.data(), tc.data(), ci.data(), ck.data(), true));
    }
    for (size_t k = 0; k < clean.size(); k++) { for (int d = 0; d < 3; d++) { clean[k].key_org[d] = tmp[k].key_org[d]; clean[k].key_unit[d] = tmp[k].key_unit[d]; } clean[k].voxel_lattice = tmp[k].voxel_lattice; }
    {
        std::vector<DevCloud *> trees; std::vector<const float4 *> pr; std::vector<float4 *> no; std::vector<const uint8_t *> td;
        for (int g = 0; g < G; g++)
            for (int s = 0; s < n_scales; s++) {
                const size_t k = (size_t)(2 * g + 1) * n_scales + s;
                trees.push_back(&clean[k]); pr.push_back(priors[k]); no.push_back(clean[k].nrm); td.push_back(todos[k]);
            }
        PCR_TRY(pcr_dev_build_bvh_batch(ctx, trees.data(), (int)trees.size()));
    }
    // ---- the GICP loops, one lockstep loop per scale
    std::vector<double> T((size_t)G * 16), md((size_t)G);
    std::vector<pcr_result> res((size_t)G);
    std::vector<int32_t *> match((size_t)G);
    for (int g = 0; g < G; g++) {
        memcpy(&T[16 * g], px[g]->base.init_T, 16 * sizeof(double));
        match[g] = arena<int32_t>(ctx, n[2 * g]);
        if (!match[g]) return PCR_ENOMEM;
    }
    int rc_all = 1;
    if (n_scales > 1 && pcr_options().icp_scales.load(std::memory_order_relaxed)) {      // every pair goes on to its next scale by itself (pcr_dev_gicp_group_scales)
        std::vector<const DevCloud *> ss((size_t)G * n_scales), tt((size_t)G * n_scales);
        std::vector<double> mds((size_t)G * n_scales); std::vector<pcr_result> rall((size_t)G * n_scales);
        for (int g = 0; g < G; g++)
            for (int s = 0; s < n_scales; s++) {
                ss[(size_t)g * n_scales + s] = &clean[(size_t)(2 * g) * n_scales + s]; tt[(size_t)g * n_scales + s] = &clean[(size_t)(2 * g + 1) * n_scales + s];
                mds[(size_t)g * n_scales + s] = dists[(size_t)g * n_scales + s];
            }
        rc_all = pcr_dev_gicp_group_scales(ctx, G, n_scales, ss.data(), tt.data(), mds.data(), T.data(), params, rall.data(), match.data());
        if (rc_all != PCR_OK && rc_all != 1) return rc_all;
        if (rc_all == PCR_OK)
            for (int g = 0; g < G; g++) for (int s = 0; s < n_scales; s++) px[g]->base.records[s].icp = rall[(size_t)g * n_scales + s];
    }
    for (int s = 0; s < n_scales && rc_all == 1; s++) {
        std::vector<const DevCloud *> ss((size_t)G), tt((size_t)G);
        for (int g = 0; g < G; g++) { ss[g] = &clean[(size_t)(2 * g) * n_scales + s]; tt[g] = &clean[(size_t)(2 * g + 1) * n_scales + s]; md[g] = dists[(size_t)g * n_scales + s]; }
        PCR_TRY(pcr_dev_gicp_group(ctx, G, ss.data(), tt.data(), md.data(), T.data(), params, res.data(), match.data()));
        for (int g = 0; g < G; g++) { px[g]->base.records[s].icp = res[g]; memcpy(&T[16 * g], res[g].transformation, 16 * sizeof(double)); }
    }
    {   // correspondence sets of the group's pairs in three batched launches and no read-back (round 5: three launches and a host wait PER PAIR cost the
        // script-2 stage on the NCLT scans a fifth of its rate)
        std::vector<const int32_t *> mm; std::vector<const int *> nn; std::vector<int> cc; std::vector<int32_t *> oo;
        for (int g = 0; g < G; g++)
            if (px[g]->base.correspondences) {
                const DevCloud &last = clean[(size_t)(2 * g) * n_scales + n_scales - 1];
                mm.push_back(match[g]); nn.push_back(last.n); cc.push_back(last.cap); oo.push_back((int32_t *)px[g]->base.correspondences);
            }
        if (!mm.empty()) PCR_TRY(pcr_dev_compact_matches_batch(ctx, (int)mm.size(), mm.data(), nn.data(), cc.data(), oo.data()));
    }
    std::vector<int> h((size_t)C * n_scales * 2);
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), cnt, sizeof(int) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int g = 0; g < G; g++)
        for (int s = 0; s < n_scales; s++) {
            pcr_scale_record &r = px[g]->base.records[s];
            for (int w = 0; w < 2; w++) { r.n_voxel[w] = h[2 * ((size_t)(2 * g + w) * n_scales + s)]; r.n_clean[w] = h[2 * ((size_t)(2 * g + w) * n_scales + s) + 1]; }
        }
    return PCR_OK;
}

static int multiscale_gicp_impl(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src,
                                const float *tgt_xyz, const float *tgt_normals, int64_t n_tgt, const double *voxels,
                                const double *dists, int n_scales, int sor_k, double sor_std, int normal_k,
                                const double *init_T, const pcr_gicp_params *params, pcr_scale_record *records,
                                int32_t *correspondences) {
    if (!params || !records || !voxels || !dists || n_scales < 1 || n_src < 0 || n_tgt < 0) return PCR_EINVAL;
    if ((n_src > 0 && !src_xyz) || (n_tgt > 0 && !tgt_xyz)) return PCR_EINVAL;
    PCR_TRY(check_T(ctx, init_T));
    for (int s = 0; s < n_scales; s++) {
        if (!(voxels[s] > 0.0)) { ctx->err = "voxel_size <= 0"; return PCR_EINVAL; }
        if (!(dists[s] > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    }
    if (sor_k < 1 || !(sor_std > 0.0) || normal_k < 1) { ctx->err = "nb_neighbors < 1, std_ratio <= 0 or knn < 1"; return PCR_EINVAL; }
    const size_t blk_s = 2 * pcr_scratch_bytes_for(n_src), blk_t = 2 * pcr_scratch_bytes_for(n_tgt);
    static const int ahead_env = getenv("PCR_PIPELINE") ? atoi(getenv("PCR_PIPELINE")) : 3;
    constexpr int MAX_RING = 4;
    const int ring = ahead_env < 1 ? 1 : (ahead_env > MAX_RING ? MAX_RING : ahead_env);      // scales prepared ahead (+ the one in use)
    // the voxel stage of all scales in one pass per cloud (13 launches instead of 13 per scale); PCR_VOXEL_MERGED=0: one by one
    static const bool merged_env = !(getenv("PCR_VOXEL_MERGED") && atoi(getenv("PCR_VOXEL_MERGED")) == 0);
    const bool try_merged = merged_env && n_scales >= 2 && n_scales <= 8;
    const size_t vox_bytes = try_merged ? (size_t)n_scales * ((size_t)(n_src > 0 ? n_src : 1) + (size_t)(n_tgt > 0 ? n_tgt : 1) + 512) * 136 + (1u << 16) : 0;   // points, keys, normals, tree
    PCR_TRY(pcr_arena_reserve(ctx, (size_t)ring * (blk_s + blk_t) + pcr_scratch_bytes_for(n_src) + vox_bytes));
    double bs[6], bt[6];
    PCR_TRY(pcr_dev_bounds(ctx, src_xyz, n_src, bs));
    PCR_TRY(pcr_dev_bounds(ctx, tgt_xyz, n_tgt, bt));
    // all scales of a cloud in one set of batched launches (PCR_SOR_BATCH=0: scale by scale, overlapped with the GICP loops)
    static const bool batch_env = !(getenv("PCR_SOR_BATCH") && atoi(getenv("PCR_SOR_BATCH")) == 0);
    if (batch_env && try_merged) {
        const int rc = multiscale_batched(ctx, src_xyz, src_normals, n_src, tgt_xyz, tgt_normals, n_tgt, voxels, dists, n_scales, sor_k, sor_std, normal_k, init_T, params,
                                          records, correspondences, bs, bt);
        if (rc != 1) return rc;
        PCR_TRY(pcr_arena_reserve(ctx, (size_t)ring * (blk_s + blk_t) + pcr_scratch_bytes_for(n_src) + vox_bytes));      // declined: the path below
    }
    double T[16];
    memcpy(T, init_T, sizeof T);
    // Preprocessing never depends on the pose, so it runs AHEAD of the GICP loop, each cloud on its own lane (stream +
    // private block of the arena): the many small latency-bound launches of the coming scales (sort passes, octree
    // levels, scans) and their VALU-heavy k-NN kernels fill the machine while the latency-bound iterations of scale s
    // run on the caller's stream.  Blocks form a ring of `ring` scales; a block is recycled only after the GICP of its
    // scale has been waited for on the host.
    char *blocks[MAX_RING][2];
    for (int r = 0; r < ring; r++) {
        blocks[r][0] = (char *)pcr_arena_alloc(ctx, blk_s); blocks[r][1] = (char *)pcr_arena_alloc(ctx, blk_t);
        if (!blocks[r][0] || !blocks[r][1]) return PCR_ENOMEM;
    }
    static const int n_lanes = getenv("PCR_LANES") ? atoi(getenv("PCR_LANES")) : 2;
    PCR_TRY(pcr_ensure_lanes(ctx, n_lanes));
    hipStream_t lane_t = ctx->side_stream, lane_s = n_lanes > 1 ? ctx->side_stream2 : ctx->side_stream;
    struct LaneGuard {            // an early error return must not leave lane work running over a recycled arena
        hipStream_t a, b;
        ~LaneGuard() { (void)hipStreamSynchronize(a); if (b != a) (void)hipStreamSynchronize(b); }
    } guard{lane_t, lane_s};
    PCR_HIP_CHECK(ctx, hipEventRecord(ctx->side_ev[0], ctx->stream));       // inputs are ready once the caller's stream gets here
    PCR_HIP_CHECK(ctx, hipStreamWaitEvent(lane_t, ctx->side_ev[0], 0));
    if (lane_s != lane_t) PCR_HIP_CHECK(ctx, hipStreamWaitEvent(lane_s, ctx->side_ev[0], 0));
    DevCloud cs[MAX_RING], ct[MAX_RING];
    int *cnt4 = arena<int>(ctx, 4 * MAX_RING);        // per ring slot: voxel counts (source, target), clean counts (source, target)
    if (!cnt4) return PCR_ENOMEM;
    DevCloud vs[8], vt[8];                            // voxel clouds of all scales (points, keys, count; no tree) when the merged pass runs
    bool merged_s = false, merged_t = false;
    if (try_merged) {
        for (int s = 0; s < n_scales; s++) {
            PCR_TRY(pcr_alloc_cloud(ctx, &vt[s], (int)n_tgt, tgt_normals != nullptr, true));
            PCR_TRY(pcr_alloc_cloud(ctx, &vs[s], (int)n_src, src_normals != nullptr, true));
        }
        DevCloud *pt[8], *ps[8];
        for (int s = 0; s < n_scales; s++) { pt[s] = &vt[s]; ps[s] = &vs[s]; }
        {   // scratch of the passes = the lane's first ring block, which the same lane reuses afterwards (stream order)
            SideLane lane(ctx, blocks[0][1], blk_t, lane_t);
            PCR_TRY(pcr_dev_voxel_multi(ctx, tgt_xyz, tgt_normals, n_tgt, bt, voxels, n_scales, vt, &merged_t));
            if (merged_t) PCR_TRY(pcr_dev_build_bvh_batch(ctx, pt, n_scales));      // the trees of all scales: 6 launches
        }
        {
            SideLane lane(ctx, blocks[0][0], blk_s, lane_s);
            PCR_TRY(pcr_dev_voxel_multi(ctx, src_xyz, src_normals, n_src, bs, voxels, n_scales, vs, &merged_s));
            if (merged_s) PCR_TRY(pcr_dev_build_bvh_batch(ctx, ps, n_scales));
        }
    }
    auto enqueue_prep = [&](int s) -> int {
        const int r = s % ring;
        {
            SideLane lane(ctx, blocks[r][1], blk_t, lane_t);
            PCR_TRY(prep_scale(ctx, tgt_xyz, tgt_normals, n_tgt, bt, voxels[s], sor_k, sor_std, normal_k, &ct[r], cnt4 + 4 * r + 1, cnt4 + 4 * r + 3, true, merged_t ? &vt[s] : nullptr));
            PCR_HIP_CHECK(ctx, hipEventRecord(ctx->lane_ev[2 * r + 1], ctx->stream));
        }
        {
            SideLane lane(ctx, blocks[r][0], blk_s, lane_s);
            PCR_TRY(prep_scale(ctx, src_xyz, src_normals, n_src, bs, voxels[s], sor_k, sor_std, normal_k, &cs[r], cnt4 + 4 * r, cnt4 + 4 * r + 2, false, merged_s ? &vs[s] : nullptr));
            PCR_HIP_CHECK(ctx, hipEventRecord(ctx->lane_ev[2 * r], ctx->stream));
        }
        return PCR_OK;
    };
    int prepared = 0;
    for (; prepared < n_scales && prepared < ring; prepared++) PCR_TRY(enqueue_prep(prepared));
    for (int s = 0; s < n_scales; s++) {
        const int r = s % ring;
        ArenaMark mark(ctx);
        PCR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_ev[2 * r], 0));
        PCR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_ev[2 * r + 1], 0));
        int32_t *match = arena<int32_t>(ctx, n_src > 0 ? n_src : 1);
        if (!match) return PCR_ENOMEM;
        // measurement only (option "fence_prep" with profiling on: bench.py's solo roofline pass of config 5): no later scale's preprocessing shares
        // the chip with this scale's loop, so the HIP-event time per launch is the iteration kernels' own
        if (ctx->profiling && pcr_options().fence_prep.load(std::memory_order_relaxed)) {
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(lane_t)); PCR_HIP_CHECK(ctx, hipStreamSynchronize(lane_s));
        }
        PCR_TRY(pcr_dev_gicp(ctx, &cs[r], &ct[r], dists[s], T, params, &records[s].icp, match));
        int h[4];
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(h, cnt4 + 4 * r, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        records[s].n_voxel[0] = h[0]; records[s].n_voxel[1] = h[1]; records[s].n_clean[0] = h[2]; records[s].n_clean[1] = h[3];
        memcpy(T, records[s].icp.transformation, sizeof T);
        if (s == n_scales - 1 && correspondences) {
                PCR_TRY(pcr_dev_compact_matches(ctx, match, cs[r].n, cs[r].cap, nullptr, nullptr, correspondences, nullptr));      // (the count is the result's n_correspondences: no read-back)
        }
        // the block of scale s is free again (its GICP has been waited for): prepare the next scale not yet enqueued
        if (prepared < n_scales) { PCR_TRY(enqueue_prep(prepared)); prepared++; }
    }
    return PCR_OK;
}
extern "C" int pcr_multiscale_gicp(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src,
                                   const float *tgt_xyz, const float *tgt_normals, int64_t n_tgt, const double *voxels,
                                   const double *dists, int n_scales, int sor_k, double sor_std, int normal_k,
                                   const double *init_T, const pcr_gicp_params *params, pcr_scale_record *records,
                                   int32_t *correspondences) {
    return pcr_api_call(ctx, [&]() -> int {
        return multiscale_gicp_impl(ctx, src_xyz, src_normals, n_src, tgt_xyz, tgt_normals, n_tgt, voxels, dists, n_scales, sor_k, sor_std, normal_k,
                                    init_T, params, records, correspondences);
    });
}

// ---------------------------------------------------------------------------------- many pairs per call
// Worker contexts live in a process-wide pool (arena, streams and cached graphs survive between calls); a call borrows
// `inflight` of them, spawns as many threads and lets them pull pair indices from a shared counter.
namespace {
std::mutex g_pool_mutex;
std::vector<std::pair<int, pcr_context *>> g_pool;        // (device, idle context)
pcr_context *pool_take(int device) {
    {
        // the pooled context with the LARGEST arena: a call with fewer workers than the pool holds then reuses the same grown
        // contexts every time (taking them in turn made every call re-grow arenas after a call with more workers had filled the pool)
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t best = g_pool.size();
        for (size_t k = 0; k < g_pool.size(); k++)
            if (g_pool[k].first == device && (best == g_pool.size() || g_pool[k].second->arena_cap > g_pool[best].second->arena_cap)) best = k;
        if (best < g_pool.size()) { pcr_context *c = g_pool[best].second; g_pool.erase(g_pool.begin() + best); return c; }
    }
    pcr_context *c = nullptr;
    return pcr_create(device, &c) == PCR_OK ? c : nullptr;
}
void pool_give(int device, pcr_context *c) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool.emplace_back(device, c);
}
}  // namespace

extern "C" int pcr_pool_profile(int device, int enable, double *out16, int reset) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (out16) for (int i = 0; i < 16; i++) out16[i] = 0.0;
    for (auto &e : g_pool) {
        if (e.first != device) continue;
        if (enable >= 0) e.second->profiling = enable ? 1 : 0;
        for (int i = 0; i < 16; i++) { if (out16) out16[i] += e.second->prof[i]; if (reset) e.second->prof[i] = 0; }
    }
    return PCR_OK;
}

static int information_matrix_impl(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz, int64_t n_tgt,
                                   double max_dist, const double *T, double *info36);
// one pair of a plan on a worker context (one pcr_enter / pcr_leave around everything: launch errors are the pair's)
static int run_pair(pcr_context *ctx, pcr_pair_ex &px, int index, const pcr_pairs_plan &plan) {
    return pcr_api_call(ctx, [&]() -> int {
        pcr_pair &p = px.base;
        const bool do_fgr = plan.stage == PCR_STAGE_FGR || plan.stage == PCR_STAGE_FGR_GICP, do_gicp = plan.stage == PCR_STAGE_GICP || plan.stage == PCR_STAGE_FGR_GICP;
        if (!do_fgr && !do_gicp) { ctx->err = "unknown stage"; return PCR_EINVAL; }
        if (do_fgr && !plan.fgr) { ctx->err = "plan.fgr == NULL"; return PCR_EINVAL; }
        if (do_gicp && (!plan.voxel_sizes || !plan.gicp || plan.n_scales < 1 || plan.n_scales > 8 || !p.records || (plan.radius_rule == 0 && !plan.max_distances))) {
            ctx->err = "plan: missing scale tables / parameters / records (1..8 scales)"; return PCR_EINVAL;
        }
        const float *sn = p.src_normals, *tn = p.tgt_normals;
        double T0[16];
        memcpy(T0, p.init_T, sizeof T0);
        if (do_fgr) {
            float *sno = px.src_normals_out, *tno = px.tgt_normals_out;
            if (do_gicp && plan.gicp_prior_from_fgr && (!sno || !tno)) {
                // the normals registro_FGR leaves on the clouds outlive its arena: context-owned side buffer
                const size_t need = (size_t)(p.n_src + p.n_tgt + 2) * 3 * sizeof(float);
                if (need > ctx->aux_cap) {
                    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                    if (ctx->aux) { PCR_HIP_CHECK(ctx, hipFree(ctx->aux)); ctx->aux = nullptr; ctx->aux_cap = 0; }
                    if (hipMalloc((void **)&ctx->aux, need + need / 4) != hipSuccess) { ctx->err = "hipMalloc(aux)"; return PCR_ENOMEM; }
                    ctx->aux_cap = need + need / 4;
                }
                if (!sno) sno = (float *)ctx->aux;
                if (!tno) tno = (float *)ctx->aux + (size_t)(p.n_src + 1) * 3;
            }
            pcr_fgr_params fp = *plan.fgr;
            fp.option.seed = plan.fgr->option.seed + (uint64_t)index;
            PCR_TRY(pcr_registro_fgr_impl(ctx, p.src_xyz, p.src_normals, p.n_src, p.tgt_xyz, p.tgt_normals, p.n_tgt, &fp, sno, tno, &px.fgr,
                                          do_gicp ? nullptr : p.correspondences));
            memcpy(T0, px.fgr.transformation, sizeof T0);
            if (do_gicp && plan.gicp_prior_from_fgr) { sn = sno; tn = tno; }
        }
        const double *Tfinal = T0;
        if (do_gicp) {
            double dists[8];
            if (plan.radius_rule == 1) {              // ALL_FUNCTIONS.py:277-278 + 1092-1101
                double bs[6], bt[6];
                PCR_TRY(pcr_arena_reserve(ctx, 1 << 20));
                PCR_TRY(pcr_dev_bounds(ctx, p.src_xyz, p.n_src, bs));
                PCR_TRY(pcr_dev_bounds(ctx, p.tgt_xyz, p.n_tgt, bt));
                const double r1 = std::pow((bs[3] - bs[0]) * (bs[4] - bs[1]) * (bs[5] - bs[2]), 1.0 / 3.0), r2 = std::pow((bt[3] - bt[0]) * (bt[4] - bt[1]) * (bt[5] - bt[2]), 1.0 / 3.0);
                const double r = (r1 + r2) / 2;
                for (int i = 0; i < plan.n_scales; i++) dists[i] = r * std::pow(2.0, -(double)i);
            } else for (int i = 0; i < plan.n_scales; i++) dists[i] = plan.max_distances[i];
            for (int i = 0; i < 8; i++) px.max_distances[i] = i < plan.n_scales ? dists[i] : 0.0;
            PCR_TRY(multiscale_gicp_impl(ctx, p.src_xyz, sn, p.n_src, p.tgt_xyz, tn, p.n_tgt, plan.voxel_sizes, dists, plan.n_scales, plan.sor_k, plan.sor_std,
                                         plan.normal_k, T0, plan.gicp, p.records, p.correspondences));
            Tfinal = p.records[plan.n_scales - 1].icp.transformation;
        }
        if (plan.info_max_dist > 0.0) PCR_TRY(information_matrix_impl(ctx, p.src_xyz, p.n_src, p.tgt_xyz, p.n_tgt, plan.info_max_dist, Tfinal, px.info36));
        return PCR_OK;
    });
}

extern "C" int pcr_register_pairs_plan(int device, pcr_pair_ex *pairs, int n_pairs, const pcr_pairs_plan *plan, void *after_stream) {
    if (n_pairs < 0 || (n_pairs > 0 && !pairs) || !plan) return PCR_EINVAL;
    if (n_pairs == 0) return PCR_OK;
    if (hipSetDevice(device) != hipSuccess) return PCR_EHIP;
    const int inflight = plan->inflight;
    // lockstep groups: `group` consecutive pairs go through the same GICP launches, `inflight` groups in flight (stage FGR + GICP: the
    // worker runs registro_FGR pair by pair and then the group's GICP in lockstep from the FGR poses)
    const int fgr_group = plan->fgr_group > 1 ? (plan->fgr_group > 64 ? 64 : plan->fgr_group) : 1;
    const int group = plan->stage == PCR_STAGE_FGR ? fgr_group
                    : (((plan->stage == PCR_STAGE_GICP || plan->stage == PCR_STAGE_FGR_GICP) && plan->group > 1) ? (plan->group > 24 ? 24 : plan->group)      /* (32 in lockstep were measured at 310 pairs/s against 1150 with 24 on NCLT-size pairs: capped) */ : 1);
    const int units = (n_pairs + group - 1) / group;
    int workers = inflight < 1 ? 1 : (inflight > units ? units : (inflight > 16 ? 16 : inflight));
    // (option "plan_prefetch": twice the workers, of which `workers` at a time are past the gate of multiscale_group)
    PcrGate gate;
    const bool prefetch = pcr_options().plan_prefetch.load(std::memory_order_relaxed) != 0 && plan->stage == PCR_STAGE_GICP && group > 1 && units > workers;
    if (prefetch) { gate.free_slots = workers; workers = 2 * workers > units ? units : 2 * workers; if (workers > 16) workers = 16; }
    // the workers wait for everything already enqueued on `after_stream` (NULL = the legacy default stream, which is what torch's
    // default stream is): the producers of the clouds, normals and initial poses
    hipEvent_t ready = nullptr;
    if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return PCR_EHIP;
    if (hipEventRecord(ready, (hipStream_t)after_stream) != hipSuccess) { (void)hipEventDestroy(ready); return PCR_EHIP; }
    for (int i = 0; i < n_pairs; i++) { pairs[i].base.status = PCR_EHIP; snprintf(pairs[i].base.error, sizeof pairs[i].base.error, "not processed (no worker context)"); }
    std::atomic<int> next(0), failed(0), started(0);
    const int stagger_us = pcr_options().plan_stagger_us.load(std::memory_order_relaxed);      // measurement only: worker w starts w x this later
    auto work = [&]() {
        const int wid = started.fetch_add(1);
        if (stagger_us > 0 && wid > 0) std::this_thread::sleep_for(std::chrono::microseconds((long long)wid * stagger_us));
        (void)hipSetDevice(device);
        pcr_context *ctx = pool_take(device);
        if (!ctx) { failed++; return; }
        use_private_stream(ctx);
        ctx->heavy_gate = prefetch ? &gate : nullptr;
        if (ensure_stream(ctx) == PCR_OK) (void)hipStreamWaitEvent(ctx->stream, ready, 0);
        const bool gicp_stage = plan->stage == PCR_STAGE_GICP || plan->stage == PCR_STAGE_FGR_GICP;
        auto group_unit = [&](int i, int cnt) -> int {
            int rc_group = 1;
            {
                // `cnt` consecutive pairs through the same launches (lockstep group; a ragged last group of ONE pair as well, so that its
                // arithmetic does not depend on how the batch was cut); rc 1: declined, pair by pair below (still with the group forms)
                std::vector<pcr_pair_ex *> gp((size_t)cnt);
                std::vector<pcr_pair_ex> staged;                   // stage FGR + GICP: copies that carry the FGR pose and normals into the group's GICP
                for (int k = 0; k < cnt; k++) { gp[k] = &pairs[i + k]; pairs[i + k].base.error[0] = 0; }
                rc_group = PCR_OK;
                if (plan->stage == PCR_STAGE_FGR_GICP) {
                    if (!plan->fgr) rc_group = 1;
                    size_t need = 0;
                    std::vector<size_t> off((size_t)cnt + 1, 0);
                    for (int k = 0; k < cnt && rc_group == PCR_OK; k++) {
                        const pcr_pair &p = pairs[i + k].base;
                        if (plan->gicp_prior_from_fgr && !(pairs[i + k].src_normals_out && pairs[i + k].tgt_normals_out)) need += (size_t)(p.n_src + p.n_tgt + 2) * 3;
                        off[k + 1] = need;
                    }
                    if (rc_group == PCR_OK && need * sizeof(float) > ctx->aux_cap)
                        rc_group = pcr_api_call(ctx, [&]() -> int {
                            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                            if (ctx->aux) { PCR_HIP_CHECK(ctx, hipFree(ctx->aux)); ctx->aux = nullptr; ctx->aux_cap = 0; }
                            if (hipMalloc((void **)&ctx->aux, need * sizeof(float) * 5 / 4) != hipSuccess) { ctx->err = "hipMalloc(aux)"; return PCR_ENOMEM; }
                            ctx->aux_cap = need * sizeof(float) * 5 / 4;
                            return PCR_OK;
                        });
                    staged.resize((size_t)cnt);
                    std::vector<char> fgr_done((size_t)cnt, 0);
                    std::vector<float *> snos((size_t)cnt), tnos((size_t)cnt);
                    for (int k = 0; k < cnt; k++) {
                        pcr_pair_ex &px = pairs[i + k];
                        snos[k] = px.src_normals_out; tnos[k] = px.tgt_normals_out;
                        if (rc_group == PCR_OK && plan->gicp_prior_from_fgr && !(snos[k] && tnos[k])) {
                            float *base = (float *)ctx->aux + off[k];
                            if (!snos[k]) snos[k] = base;
                            if (!tnos[k]) tnos[k] = base + (size_t)(px.base.n_src + 1) * 3;
                        }
                    }
                    if (rc_group == PCR_OK && fgr_group > 1 && cnt > 1) {          // registro_FGR of the unit's pairs in lockstep chunks; what the group form does not take runs below
                        std::vector<int> take;
                        for (int k = 0; k < cnt; k++) if (pcr_fgr_group_takes(pairs[i + k].base.n_src, pairs[i + k].base.n_tgt)) take.push_back(k);
                        for (size_t t0 = 0; t0 < take.size() && take.size() >= 2; t0 += (size_t)fgr_group) {
                            const size_t m = t0 + (size_t)fgr_group <= take.size() ? (size_t)fgr_group : take.size() - t0;
                            std::vector<pcr_fgr_group_pair> q(m);
                            for (size_t t = 0; t < m; t++) {
                                const int k = take[t0 + t];
                                pcr_pair_ex &px = pairs[i + k];
                                q[t] = pcr_fgr_group_pair{px.base.src_xyz, px.base.src_normals, px.base.n_src, px.base.tgt_xyz, px.base.tgt_normals, px.base.n_tgt, *plan->fgr,
                                                          snos[k], tnos[k], &px.fgr, nullptr, PCR_OK};
                                q[t].p.option.seed = plan->fgr->option.seed + (uint64_t)(i + k);
                            }
                            const int rc = pcr_api_call(ctx, [&]() -> int { return pcr_registro_fgr_group(ctx, q.data(), (int)m); });
                            if (rc == PCR_EHIP) { rc_group = rc; break; }
                            if (rc == PCR_OK) for (size_t t = 0; t < m; t++) { fgr_done[take[t0 + t]] = q[t].status == PCR_OK; if (q[t].status == 1) pcr_counters().fgr_group_pairs_redone_alone++; }
                        }
                    }
                    for (int k = 0; k < cnt && rc_group == PCR_OK; k++) {
                        pcr_pair_ex &px = pairs[i + k];
                        float *sno = snos[k], *tno = tnos[k];
                        const int rc = fgr_done[k] ? PCR_OK : pcr_api_call(ctx, [&]() -> int {
                            pcr_fgr_params fp = *plan->fgr;
                            fp.option.seed = plan->fgr->option.seed + (uint64_t)(i + k);
                            return pcr_registro_fgr_impl(ctx, px.base.src_xyz, px.base.src_normals, px.base.n_src, px.base.tgt_xyz, px.base.tgt_normals, px.base.n_tgt, &fp, sno, tno, &px.fgr, nullptr);
                        });
                        if (rc != PCR_OK) { rc_group = rc == PCR_EHIP ? rc : 1; break; }       // an argument error: pair by pair, so that it lands on its pair
                        staged[k] = px;
                        memcpy(staged[k].base.init_T, px.fgr.transformation, sizeof staged[k].base.init_T);
                        if (plan->gicp_prior_from_fgr) { staged[k].base.src_normals = sno; staged[k].base.tgt_normals = tno; }
                        gp[k] = &staged[k];
                    }
                }
                if (rc_group == PCR_OK)
                rc_group = pcr_api_call(ctx, [&]() -> int {
                    return multiscale_group(ctx, gp.data(), cnt, plan->voxel_sizes, plan->max_distances, plan->radius_rule, plan->n_scales, plan->sor_k, plan->sor_std, plan->normal_k, plan->gicp);
                });
                if (rc_group == PCR_OK && plan->stage == PCR_STAGE_FGR_GICP)
                    for (int k = 0; k < cnt; k++) memcpy(pairs[i + k].max_distances, staged[k].max_distances, sizeof staged[k].max_distances);
                if (rc_group == PCR_OK && plan->info_max_dist > 0.0)
                    for (int k = 0; k < cnt && rc_group == PCR_OK; k++) {
                        pcr_pair_ex &px = pairs[i + k];
                        rc_group = pcr_api_call(ctx, [&]() -> int {
                            return information_matrix_impl(ctx, px.base.src_xyz, px.base.n_src, px.base.tgt_xyz, px.base.n_tgt, plan->info_max_dist,
                                                           px.base.records[plan->n_scales - 1].icp.transformation, px.info36);
                        });
                    }
                if (rc_group == PCR_OK) {
                    for (int k = 0; k < cnt; k++) pairs[i + k].base.status = PCR_OK;
                } else if (rc_group == PCR_EHIP) {          // a device error is not one pair's fault
                    for (int k = 0; k < cnt; k++) { pairs[i + k].base.status = rc_group; failed++; snprintf(pairs[i + k].base.error, sizeof pairs[i + k].base.error, "%s", ctx->err.c_str()); }
                } else rc_group = 1;                        // an argument / capacity error: pair by pair, so that it lands on the pair that has it
            }
            return rc_group;
        };
        // stage FGR: `cnt` pairs through registro_FGR in lockstep (pcr_registro_fgr_group); pairs it leaves (status 1) and groups it declines run one by one
        auto fgr_unit = [&](int i, int cnt) -> int {
            std::vector<int> take;                                   // the pairs of the unit the group form takes (sizes); the others run alone below
            for (int k = 0; k < cnt; k++) { pairs[i + k].base.error[0] = 0; if (pcr_fgr_group_takes(pairs[i + k].base.n_src, pairs[i + k].base.n_tgt)) take.push_back(k); }
            std::vector<char> done((size_t)cnt, 0);
            if (take.size() >= 2) {
                std::vector<pcr_fgr_group_pair> q(take.size());
                for (size_t t = 0; t < take.size(); t++) {
                    pcr_pair_ex &px = pairs[i + take[t]];
                    q[t] = pcr_fgr_group_pair{px.base.src_xyz, px.base.src_normals, px.base.n_src, px.base.tgt_xyz, px.base.tgt_normals, px.base.n_tgt, *plan->fgr,
                                              px.src_normals_out, px.tgt_normals_out, &px.fgr, px.base.correspondences, PCR_OK};
                    q[t].p.option.seed = plan->fgr->option.seed + (uint64_t)(i + take[t]);
                }
                const int rc = pcr_api_call(ctx, [&]() -> int { return pcr_registro_fgr_group(ctx, q.data(), (int)q.size()); });
                if (rc == PCR_EHIP) {
                    for (int k = 0; k < cnt; k++) { pairs[i + k].base.status = rc; failed++; snprintf(pairs[i + k].base.error, sizeof pairs[i + k].base.error, "%s", ctx->err.c_str()); }
                    return rc;
                }
                if (rc == PCR_OK) for (size_t t = 0; t < take.size(); t++) { if (q[t].status == PCR_OK) { done[take[t]] = 1; pairs[i + take[t]].base.status = PCR_OK; } else if (q[t].status == 1) pcr_counters().fgr_group_pairs_redone_alone++; }
            }
            for (int k = 0; k < cnt; k++) {
                if (done[k]) continue;
                pcr_pair &p = pairs[i + k].base;
                p.status = run_pair(ctx, pairs[i + k], i + k, *plan);
                if (p.status != PCR_OK) { failed++; snprintf(p.error, sizeof p.error, "%s", ctx->err.c_str()); }
            }
            return PCR_OK;
        };
        auto solo_unit = [&](int i, int cnt) -> int {
            for (int k = 0; k < cnt; k++) {
                pcr_pair &p = pairs[i + k].base;
                p.error[0] = 0;
                p.status = run_pair(ctx, pairs[i + k], i + k, *plan);
                if (p.status != PCR_OK) { failed++; snprintf(p.error, sizeof p.error, "%s", ctx->err.c_str()); }
            }
            return PCR_OK;
        };
        for (;;) {
            const int i = next.fetch_add(group);
            if (i >= n_pairs) break;
            const int cnt = i + group <= n_pairs ? group : n_pairs - i;
            if (plan->stage == PCR_STAGE_FGR) {
                if (group > 1 && cnt > 1 && plan->fgr) fgr_unit(i, cnt); else solo_unit(i, cnt);
                continue;
            }
            // kernel forms (ctx->group_forms: wavefront k-NN, 1024-point iteration tiles).  plan->pair_forms == 0: by the plan -- every unit
            // of a plan with group > 1 takes the group forms, a ragged last group of ONE pair included, so that a pair's arithmetic does not
            // depend on how the batch was cut.  pair_forms != 0: by the PAIR alone -- group forms iff both clouds are under
            // PCR_GROUP_FORMS_MAX_POINTS -- whatever `group` says, so that shards of any world size produce the single-GPU bits.
            auto small_pair = [&](int k) { const pcr_pair &p = pairs[k].base; return (p.n_src > p.n_tgt ? p.n_src : p.n_tgt) < PCR_GROUP_FORMS_MAX_POINTS; };
            bool all_small = true;
            for (int k = 0; k < cnt; k++) all_small = all_small && small_pair(i + k);
            if (plan->pair_forms && gicp_stage && !all_small) {
                for (int k = 0; k < cnt; k++) {
                    ctx->group_forms = small_pair(i + k);
                    int rc1 = ctx->group_forms ? group_unit(i + k, 1) : 1;
                    if (rc1 == 1) rc1 = solo_unit(i + k, 1);
                }
                continue;
            }
            ctx->group_forms = gicp_stage && (plan->pair_forms ? true : group > 1);
            int rc_group = ctx->group_forms ? group_unit(i, cnt) : 1;
            if (rc_group == 1) solo_unit(i, cnt);
        }
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        ctx->group_forms = false;
        ctx->heavy_gate = nullptr;
        pool_give(device, ctx);
    };
    std::vector<std::thread> threads;
    for (int w = 1; w < workers; w++) threads.emplace_back(work);
    work();                                                                   // the calling thread is worker 0
    for (auto &t : threads) t.join();
    (void)hipEventDestroy(ready);
    return failed.load() ? PCR_EHIP : PCR_OK;
}

// the GICP-only form (2_MGICP...py:187-214): pcr_register_pairs_plan with stage PCR_STAGE_GICP and the given radii
extern "C" int pcr_register_pairs(int device, pcr_pair *pairs, int n_pairs, const double *voxels, const double *dists, int n_scales,
                                  int sor_k, double sor_std, int normal_k, const pcr_gicp_params *params, int inflight, void *after_stream) {
    if (n_pairs < 0 || (n_pairs > 0 && !pairs) || !voxels || !dists || !params || n_scales < 1) return PCR_EINVAL;
    if (n_pairs == 0) return PCR_OK;
    std::vector<pcr_pair_ex> ex((size_t)n_pairs);
    for (int i = 0; i < n_pairs; i++) { memset(&ex[i], 0, sizeof(pcr_pair_ex)); ex[i].base = pairs[i]; }
    pcr_pairs_plan plan; memset(&plan, 0, sizeof plan);
    plan.stage = PCR_STAGE_GICP; plan.voxel_sizes = voxels; plan.max_distances = dists; plan.n_scales = n_scales; plan.sor_k = sor_k; plan.sor_std = sor_std;
    plan.normal_k = normal_k; plan.gicp = params; plan.inflight = inflight;
    const int rc = pcr_register_pairs_plan(device, ex.data(), n_pairs, &plan, after_stream);
    for (int i = 0; i < n_pairs; i++) pairs[i] = ex[i].base;
    return rc;
}

int pcr_evaluate_registration_impl(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz, int64_t n_tgt,
                                   double max_dist, const double *T, pcr_result *result, int32_t *correspondences) {
    if (!result || n_src < 0 || n_tgt < 0) return PCR_EINVAL;
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    PCR_TRY(check_T(ctx, T));
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n_src) + pcr_scratch_bytes_for(n_tgt)));
    DevCloud s, t; uint32_t *sperm = nullptr, *tperm = nullptr;
    PCR_TRY(pcr_import_cloud(ctx, src_xyz, nullptr, n_src, &s, &sperm, false));
    PCR_TRY(pcr_import_cloud(ctx, tgt_xyz, nullptr, n_tgt, &t, &tperm, false));
    int32_t *match = arena<int32_t>(ctx, n_src > 0 ? n_src : 1);
    if (!match) return PCR_ENOMEM;
    PCR_TRY(pcr_dev_evaluate(ctx, &s, &t, max_dist, T, result, match, nullptr));
    for (int k = 0; k < 16; k++) result->transformation[k] = T[k];
    if (correspondences) {
        int64_t nc = 0;
        PCR_TRY(pcr_dev_compact_matches(ctx, match, s.n, s.cap, sperm, tperm, correspondences, &nc));
    }
    return PCR_OK;
}
extern "C" int pcr_evaluate_registration(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz, int64_t n_tgt,
                                         double max_dist, const double *T, pcr_result *result, int32_t *correspondences) {
    return pcr_api_call(ctx, [&]() -> int { return pcr_evaluate_registration_impl(ctx, src_xyz, n_src, tgt_xyz, n_tgt, max_dist, T, result, correspondences); });
}

static int information_matrix_impl(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz, int64_t n_tgt,
                                   double max_dist, const double *T, double *info36) {
    if (!info36 || n_src < 0 || n_tgt < 0) return PCR_EINVAL;
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    PCR_TRY(check_T(ctx, T));
    PCR_TRY(pcr_arena_reserve(ctx, pcr_scratch_bytes_for(n_src) + pcr_scratch_bytes_for(n_tgt)));
    DevCloud s, t;
    PCR_TRY(pcr_import_cloud(ctx, src_xyz, nullptr, n_src, &s, nullptr, false));
    PCR_TRY(pcr_import_cloud(ctx, tgt_xyz, nullptr, n_tgt, &t, nullptr, false));
    pcr_result r;
    return pcr_dev_evaluate(ctx, &s, &t, max_dist, T, &r, nullptr, info36);
}
extern "C" int pcr_information_matrix(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz, int64_t n_tgt,
                                      double max_dist, const double *T, double *info36) {
    return pcr_api_call(ctx, [&]() -> int { return information_matrix_impl(ctx, src_xyz, n_src, tgt_xyz, n_tgt, max_dist, T, info36); });
}
