// pcr_internal.h -- host-side internals of libpcr_hip.so (context, scratch arena, device cloud records).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <mutex>
#include <condition_variable>
#include <cstring>
#include <vector>
#include <atomic>
#include "../../include/pcr_hip.h"

// process-wide switches (pcr_set_option, include/pcr_hip.h): latched from the environment once, atomics afterwards
struct PcrOptions { std::atomic<int> knn_wave{-1}, knnw_budget{80}, fence_prep{0}, icp_phase{0}, icp_verify{0}, debug_stamps{0}, debug_visits{0}, spfh_float64{0}, radius_list_select{1}, arena_poison{0}, featnn_mutual{1}, plan_stagger_us{0}, plan_prefetch{0}, icp_scales{1}; };
PcrOptions &pcr_options();
// process-wide event counters (pcr_counter, include/pcr_hip.h): how often a slow fall-back was taken -- invisible in the results, which are the same bits
struct PcrCounters { std::atomic<long long> fgr_group_barrier_timeouts{0}, fgr_group_pool_overflows{0}, fgr_group_pairs_redone_alone{0}; };
PcrCounters &pcr_counters();
#define PCR_GROUP_FORMS_MAX_POINTS 400000      // pcr_pairs_plan.pair_forms: pairs with both clouds under this take the group forms of the kernels


// A counting gate of a plan call.  A lockstep group starts with ~5 ms of small dependent launches (bounds, voxel keys, sort, means, trees) during
// which the chip is nearly idle when all groups in flight are in that part together.  With twice the workers, the groups beyond `inflight`
// run that part early -- next to the other groups' searches and iteration loops -- and wait here before their own chip-filling part, so
// that at most `inflight` groups are in the searches and loops at a time (more of THOSE in flight is slower: DESIGN.md section 4.2).
struct PcrGate {
    std::mutex m; std::condition_variable cv; int free_slots = 0;
    void acquire() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return free_slots > 0; }); free_slots--; }
    void release() { { std::lock_guard<std::mutex> l(m); free_slots++; } cv.notify_one(); }
};
struct PcrGateToken {
    PcrGate *g;
    explicit PcrGateToken(PcrGate *gate) : g(gate) { if (g) g->acquire(); }
    ~PcrGateToken() { if (g) g->release(); }
    PcrGateToken(const PcrGateToken &) = delete; PcrGateToken &operator=(const PcrGateToken &) = delete;
};

// One captured chunk of GICP launches.  Solo calls key it by everything the launches bake in; lockstep groups key it by the launch
// form only and patch grid widths / by-value arguments into `exec` through `nodes` (kept in launch order, owned by `graph`).
struct IcpGraph {
    std::string key;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    std::vector<hipGraphNode_t> nodes;
    std::string baked;
};

struct pcr_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t side_stream = nullptr;   // preprocessing lanes: the two clouds of a scale are prepared concurrently on them,
    hipStream_t side_stream2 = nullptr;  //   one scale ahead of the GICP loop that runs on `stream`
    hipEvent_t side_ev[2] = {nullptr, nullptr};
    hipEvent_t lane_ev[8] = {};   // [ring slot][cloud]: prepared cloud ready
    char *arena = nullptr;
    size_t arena_cap = 0, arena_off = 0;
    char *aux = nullptr;           // side buffer that outlives the arena within one pair of a plan (normals left by registro_FGR)
    size_t aux_cap = 0;
    char *desc_host = nullptr, *desc_dev = nullptr;   // argument structs of large batches: pinned staging ring and its device image
    size_t desc_cap = 0, desc_off = 0;
    char *icp_group_dev = nullptr, *icp_group_host = nullptr;     // argument structs + start poses of a lockstep GICP group (fixed address: graphs read it)
    char *pinned = nullptr;        // host-pinned read-back window
    size_t pinned_cap = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int profiling = 0;             // bench instrumentation (pcr_profile_*)
    double prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [0..7] GICP loop (pcr_hip.h), [8..10] feature matching: ms, flops, launches, [11] GICP queries searched again
    std::vector<hipEvent_t> prof_events;
    std::vector<IcpGraph> icp_graphs;   // captured launch chunks of the GICP loop
    std::string err;
    // The caller's stream is the legacy default stream (pcr_set_stream(ctx, NULL), torch's default stream): the work runs on the
    // context's own stream (graph capture is not allowed on the legacy stream) and every call is fenced against the
    // default stream on both sides: own stream waits for what the caller has enqueued, the default stream waits for the call.
    bool fence_default = false;
    hipEvent_t fence_ev[2] = {nullptr, nullptr};
    // first failed kernel launch of the current call (hipGetLastError after every launch); reported by pcr_leave
    hipError_t launch_err = hipSuccess;
    const char *launch_file = nullptr; int launch_line = 0;
    // Inside a lockstep-group plan every unit takes the GROUP forms of the kernels (wavefront k-NN, 1024-point iteration tiles) whatever
    // its size: a ragged last group of one pair, or a one-pair shard of another world size, is then the same arithmetic as the pair
    // inside a full group (gathered poses are the single-GPU bits)
    struct PcrGate *heavy_gate = nullptr;   // pcr_register_pairs_plan with more workers than groups in flight: taken before the chip-filling part of a group (see PcrGate)
    bool group_forms = false;
    bool octet_only = false;     // lockstep FGR groups: every batched search runs the octet kernel, as registro_FGR's one-cloud searches do
};

// every extern "C" entry point runs its body between these two (pcr_api.hip): device, stream, fences, launch errors
int pcr_enter(pcr_context *ctx);
int pcr_leave(pcr_context *ctx, int rc);
template <class F> static inline int pcr_api_call(pcr_context *ctx, F &&body) {
    const int rc = pcr_enter(ctx);
    if (rc != PCR_OK) return rc;
    return pcr_leave(ctx, body());
}

// kernel launch + hipGetLastError: a failed launch (bad grid, missing code object, out of resources) is remembered in the
// context and turns the call's status into PCR_EHIP instead of surfacing later as a wrong result or a timeout
template <class... KArgs, class... Args>
static inline void pcr_launch(pcr_context *ctx, const char *file, int line, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem,
                              hipStream_t stream, Args &&...args) {
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, static_cast<KArgs>(args)...);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && ctx->launch_err == hipSuccess) { ctx->launch_err = e; ctx->launch_file = file; ctx->launch_line = line; }
}
#define PCR_LAUNCH(ctx, ...) pcr_launch(ctx, __FILE__, __LINE__, __VA_ARGS__)

#define PCR_HIP_CHECK(ctx, expr)                                                                   \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                        \
            return PCR_EHIP;                                                                       \
        }                                                                                          \
    } while (0)

#define PCR_TRY(expr)                                                                              \
    do {                                                                                           \
        int _rc = (expr);                                                                          \
        if (_rc != PCR_OK) return _rc;                                                             \
    } while (0)

// ---- descriptor buffer: argument structs of batches too large for the kernel arguments.  A ring in pinned host memory is the
// staging area of small asynchronous copies into a device ring of the same layout; when the ring wraps, the streams that may still
// be copying from it are drained first.
template <class A> static inline const A *pcr_desc_upload(pcr_context *ctx, const A *host, int count) {
    const size_t bytes = (sizeof(A) * (size_t)count + 255) & ~(size_t)255;
    if (!ctx->desc_host) {
        ctx->desc_cap = 8u << 20;
        if (hipHostMalloc((void **)&ctx->desc_host, ctx->desc_cap, hipHostMallocDefault) != hipSuccess) { ctx->desc_host = nullptr; ctx->err = "hipHostMalloc(descriptors)"; return nullptr; }
        if (hipMalloc((void **)&ctx->desc_dev, ctx->desc_cap) != hipSuccess) { ctx->desc_dev = nullptr; ctx->err = "hipMalloc(descriptors)"; return nullptr; }
        ctx->desc_off = 0;
    }
    if (bytes > ctx->desc_cap) { ctx->err = "descriptor batch too large"; return nullptr; }
    if (ctx->desc_off + bytes > ctx->desc_cap) {
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
        if (ctx->side_stream2) (void)hipStreamSynchronize(ctx->side_stream2);
        if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
        ctx->desc_off = 0;
    }
    char *h = ctx->desc_host + ctx->desc_off, *d = ctx->desc_dev + ctx->desc_off;
    memcpy(h, host, sizeof(A) * (size_t)count);
    if (hipMemcpyAsync(d, h, sizeof(A) * (size_t)count, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "descriptor upload failed"; return nullptr; }
    ctx->desc_off += bytes;
    return (const A *)d;
}

int pcr_ensure_lanes(pcr_context *ctx, int lanes);           // create the side stream(s) (pcr_api.hip)

// ---- arena ------------------------------------------------------------------------------------
int pcr_arena_reserve(pcr_context *ctx, size_t bytes);       // grow (sync + realloc) if needed, reset offset
void *pcr_arena_alloc(pcr_context *ctx, size_t bytes);       // bump, 256-B aligned; nullptr if exhausted
template <class T> static inline T *arena(pcr_context *ctx, size_t count) {
    return (T *)pcr_arena_alloc(ctx, count * sizeof(T));
}
// Run a scope on the side lane: a private sub-arena carved out of the main one + the side stream, so that two
// independent pipelines can be enqueued without sharing (and prematurely recycling) scratch memory.
struct SideLane {
    pcr_context *ctx; char *arena; size_t cap, off; hipStream_t stream;
    SideLane(pcr_context *c, char *base, size_t bytes, hipStream_t lane = nullptr) : ctx(c), arena(c->arena), cap(c->arena_cap), off(c->arena_off), stream(c->stream) {
        c->arena = base; c->arena_cap = bytes; c->arena_off = 0; c->stream = lane ? lane : c->side_stream;
    }
    ~SideLane() { ctx->arena = arena; ctx->arena_cap = cap; ctx->arena_off = off; ctx->stream = stream; }
};
struct ArenaMark { pcr_context *ctx; size_t off; ArenaMark(pcr_context *c) : ctx(c), off(c->arena_off) {} ~ArenaMark() { ctx->arena_off = off; } };

// ---- device-resident cloud: Morton-ordered float4 points (+ optional normals) with a device-side count ----
struct DevCloud {
    float4 *pts = nullptr;       // xyz, w = bit pattern of the original index (or 0)
    float4 *nrm = nullptr;       // optional
    float *cov6 = nullptr;       // optional raw covariances (xx,xy,xz,yy,yz,zz) in Morton order
    float4 *nrm_final = nullptr; // (pcr_dev_sor) where fused normals of the cleaned cloud go
    int *n = nullptr;            // device count
    int cap = 0;                 // host upper bound of *n
    uint64_t *keys = nullptr;    // sorted Morton keys of the points (voxel index / quantised position)
    int *oct_child = nullptr;    // linear octree (pcr_octree.h): build-time child links,
    float4 *oct_nodes = nullptr; //   packed node records (box + first/count),
    int4 *oct_up = nullptr;      //   (parent, first sibling, sibling count) per node
    struct OctMeta *oct_meta = nullptr;
    int *leaf_of = nullptr;
    int4 *pinfo = nullptr;       //   point -> (leaf, first point of the leaf, point count)
    int2 *oct_l1 = nullptr;      //   level-1 node -> (first point, point count): the "fat leaf" as a point range
    float key_org[3] = {0, 0, 0};   // lattice of the Morton keys: coordinate i <-> [org + i*unit, org + (i+1)*unit)
    float key_unit[3] = {1, 1, 1};
    bool voxel_lattice = false;     // the keys are voxel indices of a voxel grid (at most one point per lattice cell): cell hashes make sense
};

static inline size_t oct_node_capacity(int cap_points) { return (size_t)cap_points + 256; }   // entries of child[] / box pairs

// ---- sort (pcr_sort.hip; hand-written stable LSD radix sort) ---------------------------------------
size_t pcr_sort_temp_bytes(size_t n);
int pcr_sort_pairs(pcr_context *ctx, void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const uint32_t *vals_in, uint32_t *vals_out, size_t n, int end_bit);

int pcr_sort_pairs_batch(pcr_context *ctx, int count, void *const *temps, const uint64_t *const *keys_in, uint64_t *const *keys_out,
                         const uint32_t *const *vals_in, uint32_t *const *vals_out, const size_t *n, int end_bit);

// ---- cloud ops (pcr_cloud.hip) -------------------------------------------------------------------
size_t pcr_scratch_bytes_for(int64_t n);                       // generous per-cloud scratch estimate
int pcr_dev_bounds(pcr_context *ctx, const float *xyz, int64_t n, double *bounds6_host);
// voxel grid mean in Morton order of the voxel index; out.cap must be >= n
int pcr_dev_voxel(pcr_context *ctx, const float *xyz, const float *nrm_in, int64_t n, const double *bounds6,
                  double voxel, DevCloud *out);
// voxel stage of up to 8 scales in one pass (outs[s]: pts / keys / n / optional nrm allocated by the caller, cap >= n); *done = false
// (and nothing enqueued) when the scales cannot share one sort key or the scratch does not fit: run them one by one then
int pcr_dev_voxel_multi(pcr_context *ctx, const float *xyz, const float *nrm_in, int64_t n, const double *bounds6, const double *voxels,
                        int n_scales, DevCloud *outs, bool *done);
int pcr_dev_voxel_multi_batch(pcr_context *ctx, int count, const float *const *xyz, const float *const *nrm_in, const int64_t *n, const double *b6, const double *voxels,
                              int n_scales, DevCloud *outs /* count x n_scales */, bool *done);
int pcr_dev_bounds_batch(pcr_context *ctx, int count, const float *const *xyz, const int64_t *n, double *b6 /* count x 6, host */);
// Morton-sort a raw packed cloud (for kNN on un-voxelised input); perm[i] = original index of sorted point i
int pcr_dev_sort_cloud(pcr_context *ctx, const float *xyz, int64_t n, const double *bounds6, DevCloud *out,
                       uint32_t *perm);
int pcr_dev_build_bvh(pcr_context *ctx, DevCloud *c);
int pcr_dev_build_bvh_batch(pcr_context *ctx, DevCloud *const *cs, int count);     // up to 8 trees per launch (blockIdx.y)
// cell hash of level L over a voxel-lattice cloud (the GICP correspondence search with a radius of a few voxels); tables from the arena
struct GridView;
int pcr_dev_build_grid_batch(pcr_context *ctx, const DevCloud *const *cs, const int *levels, int count, GridView *views /* count; codes == nullptr: none */);
int pcr_grid_level_for(const DevCloud *c, double search_radius);      // -1: no grid for this radius on this cloud
// SOR: keep flags + compaction into `out` (out.cap >= in.cap); returns nothing to the host
int pcr_dev_sor(pcr_context *ctx, const DevCloud *in, int nb_neighbors, double std_ratio, DevCloud *out,
                uint8_t *keep_sorted /*optional device, in.cap*/, double *avg_sorted /*optional*/,
                int normal_k = 0, const float4 *prior_out = nullptr, uint8_t *todo_out = nullptr, int *todo_count = nullptr,
                int *cnt_in_out = nullptr, int *cnt_kept_out = nullptr /* optional device ints: input and kept counts */,
                bool fallback_here = false /* search the incomplete lists over `in`'s tree restricted to kept points */);
int pcr_dev_sor_batch(pcr_context *ctx, const DevCloud *const *ins, DevCloud *const *outs, int count, int nb_neighbors, double std_ratio, int normal_k,
                      const float4 *const *priors, uint8_t *const *todos, int *const *todo_counts, int *const *cnt_in, int *const *cnt_kept, bool fallback_here);
int pcr_dev_normals_batch(pcr_context *ctx, DevCloud *const *cs, int count, int search_kind, int knn, double radius, const float4 *const *priors, float4 *const *normals_out,
                          const uint8_t *const *todos);
int pcr_dev_knn_lists_batch(pcr_context *ctx, const DevCloud *const *cs, int count, int k, double radius, int32_t *const *idx, float *const *d2);
// Hybrid(r, k) neighbour lists of `count` clouds: rows of k sorted-cloud indices, cnt = entries per row (-1: scan all k slots, -1 = empty slot)
int pcr_dev_radius_lists_batch(pcr_context *ctx, const DevCloud *const *cs, int count, int k, double radius, int32_t *const *idx, int32_t *const *cnt);
int pcr_dev_flag_scan_batch(pcr_context *ctx, int count, uint8_t *const *flags, const int *const *n_ptr, const int *n_cap, int *const *pos, int *const *total_dev);
// `count` caller clouds Morton-sorted with their octrees in one pass (cs[c] allocated with a tree by the caller; perms[c]: sorted -> caller)
int pcr_import_clouds_batch(pcr_context *ctx, int count, const float *const *xyz, const float *const *nrm, const int64_t *n, DevCloud *cs, uint32_t **perms);
// normals (and optionally covariances) by k-NN / hybrid / radius neighbourhoods over the BVH
int pcr_dev_normals(pcr_context *ctx, DevCloud *c, int search_kind, int knn, double radius, const float4 *prior,
                    float4 *normals_out, float *cov6_out /*optional, sorted order, 6 per point*/, const uint8_t *todo = nullptr);
int pcr_dev_knn_debug(pcr_context *ctx, const DevCloud *c, int k, double radius, int32_t *idx, float *d2, int32_t *counts);
int pcr_read_count(pcr_context *ctx, const int *dev_n, int64_t *out);
// pos[i] = number of set flags before i, *total_dev = number of set flags (n from device pointer or n_cap)
int pcr_dev_flag_scan(pcr_context *ctx, const uint8_t *flags, const int *n_ptr, int n_cap, int *pos, int *total_dev);
int pcr_dev_compact_matches_batch(pcr_context *ctx, int count, const int32_t *const *match, const int *const *n, const int *cap, int32_t *const *corr_out,
                                  const uint32_t *const *src_perm = nullptr, const uint32_t *const *tgt_perm = nullptr);
// gather helpers between caller order and Morton order
int pcr_dev_scatter_rows_f4_to_f3(pcr_context *ctx, const float4 *src_sorted, const uint32_t *perm, const int *n, int cap, float *dst_packed);
int pcr_dev_pack_f4_to_f3(pcr_context *ctx, const float4 *src, const int *n, int cap, float *dst_packed);
int pcr_dev_unpack_f3_to_f4(pcr_context *ctx, const float *src, int64_t n, float4 *dst);
int pcr_dev_gather_f3_to_f4(pcr_context *ctx, const float *src_packed, const uint32_t *perm, int64_t n, float4 *dst);

// ---- shared by the API layer and the FGR stage (pcr_api.hip)
int pcr_alloc_cloud(pcr_context *ctx, DevCloud *c, int cap, bool with_nrm, bool with_tree);
// Morton-sorted copy of a caller cloud (+ optional normals) with its octree; perm maps sorted -> caller index
int pcr_import_cloud(pcr_context *ctx, const float *xyz, const float *nrm, int64_t n, DevCloud *c, uint32_t **perm_out, bool force_nrm);

int pcr_registro_fgr_impl(pcr_context *ctx, const float *src_xyz, const float *src_prior, int64_t ns, const float *tgt_xyz, const float *tgt_prior, int64_t nt,
                          const pcr_fgr_params *p, float *src_normals_out, float *tgt_normals_out, pcr_result *result, int32_t *correspondences);
// registro_FGR of G pairs through the same launches (pcr_fgr.hip); 1 = declined (sizes / parameters): run the pairs one by one
struct pcr_fgr_group_pair {                 // one pair of a lockstep FGR group (what pcr_registro_fgr_impl takes, per pair)
    const float *src_xyz, *src_prior; int64_t ns; const float *tgt_xyz, *tgt_prior; int64_t nt;
    pcr_fgr_params p; float *src_normals_out, *tgt_normals_out; pcr_result *result; int32_t *correspondences;
    int status;                              // out: PCR_OK, or 1 = this pair must be redone alone (record pool overflow, optimiser fallback)
};
int pcr_registro_fgr_group(pcr_context *ctx, pcr_fgr_group_pair *pairs, int G);
// whether the group form takes a pair of these sizes (both clouds of at least 64 points, under 5e9 feature-row pairs: no tile pruning)
static inline bool pcr_fgr_group_takes(int64_t ns, int64_t nt) { return ns >= 64 && nt >= 64 && (double)ns * (double)nt < 5.0e9; }
int pcr_evaluate_registration_impl(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz, int64_t n_tgt,
                                   double max_dist, const double *T, pcr_result *result, int32_t *correspondences);

// ---- feature matching (pcr_featnn.hip): exact nearest feature rows in both directions; PCR_ECAPACITY = values outside the f16 range
size_t pcr_feature_nn_scratch_bytes(int64_t n0, int64_t n1, int prune_mode = -1);      // prune_mode as pcr_feature_nn_mutual's (1: forced on)
int pcr_feature_nn_mutual(pcr_context *ctx, const float *f0, int n0, const float *f1, int n1, int32_t *out_1to0, int32_t *out_0to1,
                          int prune_mode /* -1: by size (PCR_FEATNN_PRUNE overrides), 0: off, 1: on */,
                          int mutual_only = 0 /* 1: out_0to1 is only needed where the cross check can keep it (-1 elsewhere): second direction seeded by the first */);

int pcr_feature_nn_mutual_batch(pcr_context *ctx, int G, const float *const *f0, const int *n0, const float *const *f1, const int *n1, int32_t *const *out_1to0,
                                int32_t *const *out_0to1, const int **overflow_dev, int mutual_only = 0);

// ---- gicp (pcr_gicp.hip) --------------------------------------------------------------------------
struct IcpOutputs { pcr_result res; };
int pcr_dev_gicp(pcr_context *ctx, const DevCloud *src, const DevCloud *tgt, double max_dist, const double *T0,
                 const pcr_gicp_params *p, pcr_result *out, int32_t *match_dev /*optional src.cap*/);
int pcr_dev_gicp_group_scales(pcr_context *ctx, int G, int S, const DevCloud *const *src /* G x S */, const DevCloud *const *tgt, const double *max_dists /* G x S */, const double *T0 /* G x 16 */,
                              const pcr_gicp_params *p, pcr_result *out /* G x S */, int32_t *const *match_dev /* G, optional */);       // all scales of a lockstep group in one loop; 1: declined
int pcr_dev_gicp_group(pcr_context *ctx, int G, const DevCloud *const *src, const DevCloud *const *tgt, const double *max_dists /* G */, const double *T0 /* G x 16 */,
                       const pcr_gicp_params *p, pcr_result *out /* G */, int32_t *const *match_dev /* optional, G */);
int pcr_dev_linearize_once(pcr_context *ctx, const DevCloud *src, const DevCloud *tgt, double max_dist, const double *T,
                           const pcr_gicp_params *p, double *JTJ36, double *JTr6, double *stats3, int32_t *match_dev);
int pcr_dev_evaluate(pcr_context *ctx, const DevCloud *src, const DevCloud *tgt, double max_dist, const double *T,
                     pcr_result *out, int32_t *match_dev, double *info36 /*optional*/);
// evaluate_registration of G pairs in one launch pair (per pair the arithmetic of pcr_dev_evaluate); results on the host after ONE read-back
int pcr_dev_evaluate_group(pcr_context *ctx, int G, const DevCloud *const *src, const DevCloud *const *tgt, double max_dist, const double *T /* G x 16 */,
                           pcr_result *out /* G */, int32_t *const *match_dev /* G, required */);
int pcr_dev_compact_matches(pcr_context *ctx, const int32_t *match, const int *n, int cap, const uint32_t *src_perm,
                            const uint32_t *tgt_perm, int32_t *corr_out, int64_t *n_corr);
