/*
 * pcr_hip.h -- C ABI of libpcr_hip.so, the MI355X (gfx950) pairwise-registration hot path.
 *
 * The reference has no FFI of its own: its hot path is reached through Open3D's pybind
 * module from ALL_FUNCTIONS.py / scripts 1-2.  Each entry point below
 * therefore names the Open3D binding call it replaces and the reference line that makes
 * that call.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every cloud/normal/feature pointer is a DEVICE pointer (HIP, current device of the
 *     context); `T`, option structs and result structs are HOST pointers;
 *   - clouds are packed float32 xyz rows (N x 3), normals likewise, FPFH is N x 33 float32;
 *   - poses are row-major 4x4 float64, source -> target;
 *   - caller owns every buffer; the library owns a per-context scratch arena;
 *   - return 0 on success, negative pcr_status otherwise; degenerate-but-valid results
 *     (no correspondences: fitness 0, rmse 0, T = init) are NOT errors (Open3D behaviour);
 *   - calls are ordered on the context's stream (pcr_set_stream).  Entry points with host outputs (counts, poses, results)
 *     return after those are on the host; entry points whose outputs are all device buffers (pcr_estimate_normals,
 *     pcr_estimate_covariances, pcr_compute_fpfh_feature, pcr_debug_knn) only enqueue and return;
 *   - stream NULL = the legacy default stream: the library works on a stream of its own and fences every call against the
 *     default stream on both sides (the call sees everything enqueued there before it; work enqueued there after the call
 *     sees its results);
 *   - every kernel launch is followed by hipGetLastError(); a failed launch turns the call into PCR_EHIP with file:line
 *     in pcr_last_error().
 */
#ifndef PCR_HIP_H
#define PCR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PCR_OK = 0,
    PCR_EINVAL = -1,      /* Open3D would raise (voxel_size<=0, max_dist<=0, nb_neighbors<1, std_ratio<=0 ...) */
    PCR_ENOMEM = -2,
    PCR_EHIP = -3,        /* a HIP runtime call failed; see pcr_last_error() */
    PCR_ENUMERIC = -4,    /* non-finite pose */
    PCR_ECAPACITY = -5    /* caller buffer too small */
} pcr_status;

/* == o3d.geometry.KDTreeSearchParam{KNN,Radius,Hybrid}  (ALL_FUNCTIONS.py:181,185,213,301) */
typedef enum { PCR_SEARCH_KNN = 0, PCR_SEARCH_RADIUS = 1, PCR_SEARCH_HYBRID = 2 } pcr_search_kind;
/* == o3d.pipelines.registration.{L2Loss,L1Loss,GMLoss} (ALL_FUNCTIONS.py:219,284) */
typedef enum { PCR_LOSS_L2 = 0, PCR_LOSS_L1 = 1, PCR_LOSS_GM = 2 } pcr_loss_kind;

typedef struct pcr_context pcr_context;

/* == o3d.pipelines.registration.RegistrationResult (fields read at ALL_FUNCTIONS.py:312,323,369) */
typedef struct {
    double transformation[16];
    double fitness;
    double inlier_rmse;
    int64_t n_correspondences;
    int32_t iterations;       /* pose updates applied */
    int32_t converged;
} pcr_result;

/* == TransformationEstimationForGeneralizedICP(loss) + ICPConvergenceCriteria(...)
 *    (ALL_FUNCTIONS.py:308-311, 2_MGICP...py:159-162)                                   */
typedef struct {
    int32_t loss;             /* pcr_loss_kind */
    double loss_k;            /* GMLoss k */
    double epsilon;           /* GICP covariance regulariser, Open3D default 1e-3 */
    double relative_fitness;
    double relative_rmse;
    int32_t max_iteration;
} pcr_gicp_params;

/* per-scale record of pcr_multiscale_gicp (what the roofline byte model needs) */
typedef struct {
    int64_t n_voxel[2];       /* D_k: source, target after voxel_down_sample           */
    int64_t n_clean[2];       /* C_k: after remove_statistical_outlier                 */
    pcr_result icp;
} pcr_scale_record;

/* == FastGlobalRegistrationOption (ALL_FUNCTIONS.py:189-196) */
typedef struct {
    double division_factor;
    int32_t use_absolute_scale;
    int32_t decrease_mu;
    double maximum_correspondence_distance;
    int32_t iteration_number;
    double tuple_scale;
    int32_t maximum_tuple_count;   /* pcr_registro_fgr / plans: < 0 = the reference's rule per pair, int(0.2 * int((n_src + n_tgt) / 2)) (ALL_FUNCTIONS.py:179,196) */
    int32_t tuple_test;
    uint64_t seed;            /* Open3D seeds from std::random_device; here explicit */
} pcr_fgr_option;

/* ---- context ---------------------------------------------------------------------- */
int pcr_create(int device, pcr_context **out);
int pcr_destroy(pcr_context *ctx);
int pcr_set_stream(pcr_context *ctx, void *hip_stream);       /* NULL = the legacy default stream (fenced, see above) */
const char *pcr_last_error(const pcr_context *ctx);
int pcr_version(void);

/* ---- geometry: PointCloud methods ------------------------------------------------- */
/* == PointCloud.get_min_bound/get_max_bound (ALL_FUNCTIONS.py:1093-1097); bounds6 host = min xyz, max xyz */
int pcr_bounds(pcr_context *ctx, const float *xyz, int64_t n, double *bounds6);

/* == PointCloud.voxel_down_sample (ALL_FUNCTIONS.py:293-294; 2_MGICP...py:146-147).
 * out capacity n rows. normals_in/out optional (mean, not re-normalised). Output is in Morton
 * order of the voxel index (Open3D's order is unspecified hash order).                      */
int pcr_voxel_down_sample(pcr_context *ctx, const float *xyz, const float *normals_in, int64_t n, double voxel_size,
                          float *out_xyz, float *out_normals, int64_t *out_n);

/* == PointCloud.remove_statistical_outlier (ALL_FUNCTIONS.py:297-298). keep_mask: n bytes (device),
 * out_xyz optional compacted cloud (capacity n), out_index optional int64 indices (device).      */
int pcr_remove_statistical_outlier(pcr_context *ctx, const float *xyz, int64_t n, int nb_neighbors, double std_ratio,
                                   uint8_t *keep_mask, float *out_xyz, int64_t *out_index, int64_t *out_n);

/* == PointCloud.estimate_normals(search_param) (ALL_FUNCTIONS.py:182-183, 214-215, 301-302).
 * prior_normals optional: new normal flipped to agree with it (Open3D when has_normals).         */
int pcr_estimate_normals(pcr_context *ctx, const float *xyz, int64_t n, int search_kind, int knn, double radius,
                         const float *prior_normals, float *normals);

/* == PointCloud.estimate_covariances(search_param) (ALL_FUNCTIONS.py:216-217); cov6 = xx,xy,xz,yy,yz,zz */
int pcr_estimate_covariances(pcr_context *ctx, const float *xyz, int64_t n, int search_kind, int knn, double radius,
                             float *cov6);

/* ---- registration ------------------------------------------------------------------ */
/* == registration_generalized_icp (ALL_FUNCTIONS.py:304-311; 2_MGICP...py:155-162).
 * correspondences optional device int32 [n_src x 2]; filled with n_correspondences rows.       */
int pcr_registration_generalized_icp(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src,
                                     const float *tgt_xyz, const float *tgt_normals, int64_t n_tgt,
                                     double max_correspondence_distance, const double *init_T,
                                     const pcr_gicp_params *params, pcr_result *result, int32_t *correspondences);

/* == registration_icp(..., TransformationEstimationForGeneralizedICP(loss), ...) on clouds that already carry
 *    covariances (GICP_robusto, ALL_FUNCTIONS.py:216-226): the given covariances are used untouched.
 *    cov6 = xx,xy,xz,yy,yz,zz per point, float32, device.                                                   */
int pcr_registration_generalized_icp_cov(pcr_context *ctx, const float *src_xyz, const float *src_cov6, int64_t n_src,
                                         const float *tgt_xyz, const float *tgt_cov6, int64_t n_tgt,
                                         double max_correspondence_distance, const double *init_T,
                                         const pcr_gicp_params *params, pcr_result *result, int32_t *correspondences);

/* == the whole body of Multiscale_GICP (ALL_FUNCTIONS.py:286-312 / 2_MGICP...py:140-163), device resident:
 * per scale voxel_down_sample -> remove_statistical_outlier(sor_k, sor_std) -> estimate_normals(KNN normal_k)
 * -> registration_generalized_icp, chained.  src/tgt_normals optional (AF flow orientation prior).
 * records: n_scales entries (host). correspondences: optional device int32 [n_src x 2] of the last scale. */
int pcr_multiscale_gicp(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src,
                        const float *tgt_xyz, const float *tgt_normals, int64_t n_tgt, const double *voxel_sizes,
                        const double *max_distances, int n_scales, int sor_k, double sor_std, int normal_k,
                        const double *init_T, const pcr_gicp_params *params, pcr_scale_record *records,
                        int32_t *correspondences);

/* == the per-pair loops of the reference (1_FGR...py:134-147 is the FGR one; 2_MGICP...py:187-214 and
 *    ALL_FUNCTIONS.py:349-392 the GICP ones): MANY independent pairs in one call.  The library keeps `inflight` pairs in
 *    flight on `device` (one worker thread + context + stream each, taken from a process-wide pool), pair i runs exactly
 *    pcr_multiscale_gicp on pairs[i] with the shared scale tables, and the call returns when all pairs are done.
 *    `after_stream` is the HIP stream whose already-enqueued work produces the input clouds (NULL = the legacy default
 *    stream); the workers always wait for it.  Per-pair status and error text come back in the descriptor; the return value is PCR_OK iff every pair is. */
typedef struct {
    const float *src_xyz, *src_normals; int64_t n_src;      /* device; normals optional */
    const float *tgt_xyz, *tgt_normals; int64_t n_tgt;
    double init_T[16];
    pcr_scale_record *records;                              /* host, n_scales entries */
    int32_t *correspondences;                               /* optional device int32 [n_src x 2] of the last scale */
    int32_t status;                                         /* out */
    char error[120];                                        /* out */
} pcr_pair;
int pcr_register_pairs(int device, pcr_pair *pairs, int n_pairs, const double *voxel_sizes, const double *max_distances,
                       int n_scales, int sor_k, double sor_std, int normal_k, const pcr_gicp_params *params, int inflight,
                       void *after_stream);

/* == registro_FGR as ONE call (ALL_FUNCTIONS.py:178-203 / 1_FGR...py:41-66): estimate_normals(Hybrid(normal_radius, normal_max_nn))
 *    on both clouds -> compute_fpfh_feature(Hybrid(feature_radius, feature_max_nn)) on both -> FGR with `option` ->
 *    evaluate_registration.  Each cloud is Morton-sorted and indexed ONCE for all four uses.  The reference's side effect (both
 *    inputs gain normals) is returned through src/tgt_normals_out (optional, device, caller order); src/tgt_prior are the
 *    normals the clouds already carry, if any (Open3D flips the new normal to agree with them).                                */
typedef struct {
    double normal_radius;  int32_t normal_max_nn;      /* 2 * voxel_size, 20    (ALL_FUNCTIONS.py:181) */
    double feature_radius; int32_t feature_max_nn;     /* 10 * voxel_size, 200  (ALL_FUNCTIONS.py:185) */
    pcr_fgr_option option;                             /* ALL_FUNCTIONS.py:189-196 */
} pcr_fgr_params;
int pcr_registro_fgr(pcr_context *ctx, const float *src_xyz, const float *src_prior, int64_t n_src, const float *tgt_xyz,
                     const float *tgt_prior, int64_t n_tgt, const pcr_fgr_params *params, float *src_normals_out,
                     float *tgt_normals_out, pcr_result *result, int32_t *correspondences);

/* == the reference's per-pair loops with a choice of what runs per pair:
 *    PCR_STAGE_FGR       script 1 (1_FGR...py:134-147):                 registro_FGR                      -> pairs[i].fgr
 *    PCR_STAGE_GICP      script 2 (2_MGICP...py:187-214):               Multiscale_GICP from init_T       -> pairs[i].records
 *    PCR_STAGE_FGR_GICP  Coarse_to_fine_FGR_M_GICP (ALL_FUNCTIONS.py:317-332, full_registration :349-392): registro_FGR, then
 *                        Multiscale_GICP from its pose (init_T ignored)                                   -> both
 *    radius_rule 0: max_distances as given (script 2 table); 1: ALL_FUNCTIONS.py:277-278, radius_from_cloud_pair(source, target)
 *    * 2^-scale computed per pair from the two AABBs (max_distances ignored).  gicp_prior_from_fgr: the normals registro_FGR
 *    left on the clouds are the orientation prior of every scale (the ALL_FUNCTIONS flow; the scripts reload the clouds).
 *    info_max_dist > 0: pairs[i].info36 <- get_information_matrix_from_point_clouds(source, target, info_max_dist, final pose)
 *    (ALL_FUNCTIONS.py:327-331).  The library keeps `inflight` pairs (or groups, see `group`) in flight exactly as pcr_register_pairs does. */
typedef enum { PCR_STAGE_GICP = 1, PCR_STAGE_FGR = 2, PCR_STAGE_FGR_GICP = 3 } pcr_stage;
typedef struct {
    int32_t stage;
    const pcr_fgr_params *fgr;                          /* stages with FGR; option.seed + pair index seeds pair i */
    const double *voxel_sizes, *max_distances; int32_t n_scales;
    int32_t radius_rule;
    int32_t sor_k; double sor_std; int32_t normal_k;
    const pcr_gicp_params *gicp;
    int32_t gicp_prior_from_fgr;
    double info_max_dist;
    int32_t inflight;
    int32_t group;                                      /* > 1 (stages GICP and FGR + GICP, whose FGR part stays pair by pair): `group` consecutive pairs run in LOCKSTEP through the same
                                                           launches (blockIdx.y = pair: preprocessing batched over clouds and scales, one GICP loop per
                                                           scale for the whole group); `inflight` then counts groups.  Same per-pair arithmetic as the
                                                           pair-by-pair path; at most 24 (larger values are clamped).  Every unit of such a plan -- a ragged last group of ONE pair too --
                                                           runs the GROUP forms of the kernels (one-query-per-lane k-NN, 1024-point iteration tiles),
                                                           so a pair's bits do not depend on how the batch was cut */
    int32_t pair_forms;                                 /* != 0: the kernel forms are chosen by the PAIR alone (group forms iff both clouds hold fewer than
                                                           400 000 points; larger pairs run one by one with the single-pair forms) whatever `group` is:
                                                           a pair's pose bits are then the same in every batch, group size and shard (gathered
                                                           multi-GPU poses = the single-GPU run).  What registration.register_pairs_plan(group=None) sets. */
    int32_t fgr_group;                                  /* > 1 (stages with FGR): that many consecutive pairs go through registro_FGR in LOCKSTEP -- imports, sorts,
                                                           trees, hybrid normals, FPFH lists and histograms, the mutual feature search, cross check, tuple
                                                           test, GNC optimiser and evaluation of all of them in the same launches, six host waits per
                                                           GROUP instead of eight per pair (1_FGR...py:134-147 is ~125 small dependent launches per NCLT-size
                                                           pair).  Same bits per pair as pair by pair.  Pairs the group form does not take (from ~70k points:
                                                           the tile-pruned feature search) run one by one.  Stage FGR: `inflight` counts these groups. */
} pcr_pairs_plan;
typedef struct {
    pcr_pair base;                                      /* inputs, records (stages with GICP), correspondences of the LAST stage run, status */
    pcr_result fgr;                                     /* out, stages with FGR */
    float *src_normals_out, *tgt_normals_out;           /* optional device buffers (n x 3): the normals registro_FGR leaves on the clouds */
    double max_distances[8];                            /* out: the per-scale search radii actually used (radius_rule 1) */
    double info36[36];                                  /* out when info_max_dist > 0 */
} pcr_pair_ex;
int pcr_register_pairs_plan(int device, pcr_pair_ex *pairs, int n_pairs, const pcr_pairs_plan *plan, void *after_stream);

/* measurement hook for the worker contexts pcr_register_pairs keeps in its pool (all idle between calls): enable >= 0 switches
 * their instrumentation on/off, out16 (optional) receives the SUM of their pcr_profile_read counters, reset clears them. */
int pcr_pool_profile(int device, int enable, double *out16, int reset);

/* == evaluate_registration (ALL_FUNCTIONS.py:809-820) */
int pcr_evaluate_registration(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz,
                              int64_t n_tgt, double max_correspondence_distance, const double *T, pcr_result *result,
                              int32_t *correspondences);

/* == get_information_matrix_from_point_clouds (ALL_FUNCTIONS.py:327-331); info36 host */
int pcr_information_matrix(pcr_context *ctx, const float *src_xyz, int64_t n_src, const float *tgt_xyz, int64_t n_tgt,
                           double max_correspondence_distance, const double *T, double *info36);

/* == compute_fpfh_feature (ALL_FUNCTIONS.py:186-187); feat33: n x 33 float32 (device) */
int pcr_compute_fpfh_feature(pcr_context *ctx, const float *xyz, const float *normals, int64_t n, int search_kind,
                             int knn, double radius, float *feat33);

/* == registration_fgr_based_on_feature_matching (ALL_FUNCTIONS.py:198-202) */
int pcr_registration_fgr(pcr_context *ctx, const float *src_xyz, const float *src_feat33, int64_t n_src,
                         const float *tgt_xyz, const float *tgt_feat33, int64_t n_tgt, const pcr_fgr_option *option,
                         pcr_result *result, int32_t *correspondences);

/* ---- measurement hooks (bench.py): no reference counterpart -------------------------- */
/* While enabled, pcr_multiscale_gicp / pcr_registration_generalized_icp bracket every chunk of GICP-iteration
 * launches with HIP events on the context stream and the kernel stamps itself with s_memrealtime.
 * out16 = { [0] ms of HIP-event time over chunks whose launches were all live, [1] launches in those chunks,
 *           [2] us of in-kernel time summed over live launches, [3] live launches,
 *           [4] algorithmic bytes of the live launches (48 B x source points), [5] launches issued, [6], [7] diagnostics,
 *           [8] ms of HIP-event time over the feature-matching kernels of registro_FGR, [9] their algorithmic flops
 *           (2 * 33 * Ns * Nt per direction), [10] launches, [11] GICP queries whose skip certificate did not hold (searched again),
 *           summed over the launches after the first of every scale, rest 0 } */
int pcr_profile_enable(pcr_context *ctx, int on);
int pcr_profile_read(pcr_context *ctx, double *out16, int reset);

/* ---- test hooks (exercised by tests/ only) ----------------------------------------- */
/* exact k nearest neighbours of every point of a cloud (self included), through the same index the
 * pipeline uses. idx: n x k int32, d2: n x k float32 (device), rows sorted ascending.              */
int pcr_debug_knn(pcr_context *ctx, const float *xyz, int64_t n, int k, double radius, int32_t *idx, float *d2,
                  int32_t *counts);
/* one GICP linearisation at pose T: search + A.6 sums. JTJ36, JTr6, stats3 = {count, sum d^2, sum r^2} (host) */
int pcr_debug_gicp_linearize(pcr_context *ctx, const float *src_xyz, const float *src_normals, int64_t n_src,
                             const float *tgt_xyz, const float *tgt_normals, int64_t n_tgt, double max_dist,
                             const double *T, const pcr_gicp_params *params, double *JTJ36, double *JTr6,
                             double *stats3, int32_t *match /* device n_src, optional */);

/* the mutual nearest-feature search of registro_FGR alone (33-D float32 rows, device): out_1to0[j] = row of f0 nearest to row j of f1,
 * out_0to1 likewise.  mode 0: f16-split MFMA screen + exact float64 re-check (production; with tile pruning from ~70k rows per side),
 * 1: all-pairs float64 MFMA, 2: float32 brute force, 3 / 4: the screen with tile pruning forced on / off */
int pcr_debug_feature_nn(pcr_context *ctx, const float *f0, int64_t n0, const float *f1, int64_t n1, int32_t *out_1to0, int32_t *out_0to1, int mode);

/* test / diagnostic switches of the process (no reference equivalent).  Each one is latched from the environment variable of the same
 * name in upper case with the PCR_ prefix when the library first needs it; this call overrides it afterwards without touching the
 * environment (worker threads read an atomic, never getenv).  "knn_wave": -1 by size and call form (default), 0 the octet k-NN kernel,
 * 1 the one-query-per-lane kernel for every search that fits it; "knnw_budget": candidate batches a wavefront of that kernel takes before
 * it hands its queries over; "fence_prep": measurement only -- with profiling on, every scale's GICP loop of the pipelined multiscale path
 * (clouds the batched preprocessing declines: config 5) starts after ALL preprocessing enqueued so far has finished, so that HIP-event times
 * per launch are the iteration kernels' own; "icp_phase", "icp_verify", "debug_stamps", "debug_visits": diagnostics of the GICP loop and the searches
 * (phase stamps of the fused iteration kernel, re-search of certified queries, per-call stamp print-outs, visit counts instead of results).
 * Switches between two forms of the FGR half that give the same bits (tests compare them): "spfh_float64" (0: pair features of FPFH decided in
 * float where float can and in float64 otherwise, for clouds from 60 000 points; 1: all in float64; 2: both, a disagreement is an error; 4: the
 * float pass whatever the size; 3: as 4 with a 16-entry queue, the overflow path), "radius_list_select" (1: overfull Hybrid(r, max_nn) balls finished by threshold selection; 0: by the k-best kernel),
 * "featnn_mutual" (1: the second direction of the feature search inside FGR runs only for the rows the first direction points at, under the
 * bound it found; 0: both directions in full), "icp_scales" (1: in lockstep groups of small clouds every pair goes through its scales by itself;
 * 0: one lockstep loop per scale).  "plan_stagger_us", "plan_prefetch": measurement only (delayed worker starts; twice the workers
 * behind a gate in front of a group's chip-filling part) -- both cost throughput, README.md quotes the numbers.  "arena_poison": the scratch arena is filled with this byte before every call (a read of
 * scratch nobody wrote then follows the pattern).  Returns PCR_EINVAL for an unknown name. */
int pcr_set_option(const char *name, long long value);
/* Process-wide event counters (value, or -1 for an unknown name; reset != 0 clears it): how often a lockstep registro_FGR group fell back to
 * the one-pair path for one of its pairs -- "fgr_group_barrier_timeouts" (the co-resident optimiser workgroups of the group did not all get a
 * slot in time), "fgr_group_pool_overflows" (record pool of the feature screen), "fgr_group_pairs_redone_alone" (all causes).  The results are
 * the same bits either way; the throughput is not (bench.py prints them next to the NCLT stages). */
long long pcr_counter(const char *name, int reset);

#ifdef __cplusplus
}
#endif
#endif
