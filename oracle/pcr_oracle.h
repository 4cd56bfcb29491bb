/*
 * pcr_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A float64 restatement, in plain C, of the Open3D CPU algorithms that the
 * reference repository reaches through its Python call sites
 *   ALL_FUNCTIONS.py:178-203 (registro_FGR), :211-227 (GICP_robusto),
 *   :272-313 (Multiscale_GICP), :317-332 (Coarse_to_fine_FGR_M_GICP),
 *   2_MGICP_refinement_in_NCLT_dataset.py:128-164 (script Multiscale_GICP),
 *   1_FGR_pairwise_registration_in_NCLT_dataset.py:41-66 (script registro_FGR).
 * The arithmetic itself lives in the third-party dependency `open3d`
 * (isl-org/Open3D, imported at ALL_FUNCTIONS.py:4; NO version pinned by the
 * reference; API evidence puts it at 0.15-0.18).  Open3D is absent from
 * /root/reference and from this image, so this file restates its published
 * algorithms and parity is anchored on the reference's
 * own shipped outputs:  relative_poses_FGR_GICP/NCLT/pose_{i+1}_{i}.txt given
 * relative_poses_FGR/NCLT/pose_{i+1}_{i}.txt as the initial pose
 * (tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libpcr_hip.so) never links or calls it.
 *
 * All pointers are HOST pointers, all reals are float64, matrices row-major.
 */
#ifndef PCR_ORACLE_H
#define PCR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_OK = 0, ORC_EINVAL = -1, ORC_ENOMEM = -2, ORC_ENUMERIC = -3 };

/* neighbourhood selectors == o3d.geometry.KDTreeSearchParam{KNN,Radius,Hybrid} */
enum { ORC_SEARCH_KNN = 0, ORC_SEARCH_RADIUS = 1, ORC_SEARCH_HYBRID = 2 };

/* robust kernels == o3d.pipelines.registration.{L2Loss,L1Loss,GMLoss} */
enum { ORC_LOSS_L2 = 0, ORC_LOSS_L1 = 1, ORC_LOSS_GM = 2 };

typedef struct {
    double T[16];          /* source -> target, row-major 4x4            */
    double fitness;        /* |corr| / |source|                          */
    double inlier_rmse;    /* sqrt(sum d^2 / |corr|), Euclidean           */
    int64_t n_corr;
    int32_t iterations;    /* number of pose updates applied             */
    int32_t converged;     /* 1: stopped on the two-sided tolerance test */
} orc_result;

typedef struct {
    int64_t n_voxel[2];    /* D_k  (source, target)  */
    int64_t n_clean[2];    /* C_k  (source, target)  */
    orc_result icp;        /* result of this scale   */
} orc_scale_stats;

int orc_set_num_threads(int n);            /* returns previous max threads */
int orc_get_num_threads(void);

/* exact k-NN / hybrid / radius-capped search (nanoflann semantics).
 * idx/d2 are nq x k, padded with -1 / inf; counts[i] = neighbours returned.
 * radius <= 0 means "no radius cap".                                         */
int orc_knn(const double *pts, int64_t n, int dim, const double *queries,
            int64_t nq, int k, double radius, int64_t *idx, double *d2,
            int32_t *counts);

/* PointCloud::VoxelDownSample (A.1). out_xyz capacity n*3. Output ordered by
 * (ix,iy,iz) voxel key.  normals_in/normals_out may be NULL (mean of normals,
 * NOT re-normalised, when present).                                          */
int orc_voxel_down_sample(const double *xyz, int64_t n, double voxel,
                          double *out_xyz, int64_t *out_n,
                          const double *normals_in, double *normals_out);

/* PointCloud::RemoveStatisticalOutliers (A.2): keep[i] in {0,1}.            */
int orc_remove_statistical_outlier(const double *xyz, int64_t n, int nb_neighbors,
                                   double std_ratio, uint8_t *keep,
                                   double *avg_dist /*n or NULL*/,
                                   double *mean_out, double *std_out);

/* EstimatePerPointCovariances (A.3): cov9 is n x 9 row-major.               */
int orc_estimate_covariances(const double *xyz, int64_t n, int mode, int knn,
                             double radius, double *cov9);

/* EstimateNormals (A.4). prior may be NULL; if given the new normal is
 * flipped to agree with it.  cov9_in may be NULL (then A.3 is run).          */
int orc_estimate_normals(const double *xyz, int64_t n, int mode, int knn,
                         double radius, const double *prior,
                         const double *cov9_in, double *normals);

/* smallest-eigenvalue eigenvector by the analytic 3x3 solver (A.4) */
void orc_fast_eigen3x3(const double cov[9], double normal[3]);

/* covariances from normals, GeneralizedICP.cpp InitializeCovariances (A.5.1) */
int orc_covariances_from_normals(const double *normals, int64_t n, double eps,
                                 double *cov9);

/* One linearisation of TransformationEstimationForGeneralizedICP (A.6) on a
 * given correspondence set.  src_* must already be in the target frame.
 * JTJ: 36, JTr: 6, r2: 1.                                                     */
/* test knob: size of the fixed summation chunks of orc_gicp_linearize (default 256); returns the previous value */
int orc_set_sum_chunk(int chunk);
int orc_gicp_linearize(const double *src_xyz, const double *src_cov9,
                       const double *tgt_xyz, const double *tgt_cov9,
                       const int32_t *corr, int64_t n_corr, int loss, double loss_k,
                       double *JTJ, double *JTr, double *r2);

/* GetRegistrationResultAndCorrespondences: 1-NN within max_dist (A.5.2).
 * corr is ns x 2 capacity.                                                    */
int orc_find_correspondences(const double *src_xyz, int64_t ns,
                             const double *tgt_xyz, int64_t nt, double max_dist,
                             int32_t *corr, int64_t *n_corr, double *fitness,
                             double *rmse);

/* solve JTJ x = -JTr (LDLT) and build Rz*Ry*Rx | t  (A.6)                     */
int orc_solve_update(const double *JTJ, const double *JTr, double *T16);

/* registration_generalized_icp / registration_icp with the GICP estimator (A.5).
 * Exactly one of {src_nrm, src_cov9} (and tgt likewise) must be non-NULL:
 * normals -> covariances by A.5.1; cov9 used untouched (GICP_robusto path).
 * corr may be NULL; trace (max_it+1) x 2 [fitness, rmse] may be NULL.        */
int orc_registration_gicp(const double *src_xyz, const double *src_nrm,
                          const double *src_cov9, int64_t ns,
                          const double *tgt_xyz, const double *tgt_nrm,
                          const double *tgt_cov9, int64_t nt, double max_dist,
                          const double *T0, int loss, double loss_k, double eps,
                          double rel_fitness, double rel_rmse, int max_it,
                          orc_result *out, int32_t *corr, double *trace);

/* Multiscale_GICP body (ALL_FUNCTIONS.py:286-312 / script 2:140-163): per scale
 * voxel -> SOR -> normals(KNN normal_k) -> GICP, chained.  src_nrm/tgt_nrm may be
 * NULL; when given (AF flow after registro_FGR mutated the inputs) they are
 * voxel-averaged and used as orientation prior (A.4).                         */
int orc_multiscale_gicp(const double *src_xyz, const double *src_nrm, int64_t ns,
                        const double *tgt_xyz, const double *tgt_nrm, int64_t nt,
                        const double *voxels, const double *dists, int n_scales,
                        int sor_k, double sor_std, int normal_k, const double *T0,
                        int loss, double loss_k, double eps, double rel_fitness,
                        double rel_rmse, int max_it, orc_scale_stats *stats,
                        int32_t *corr_last /* C_s x 2 capacity ns*2, or NULL */);

/* compute_fpfh_feature (A.7): feat is n x 33 (row per point).                */
int orc_compute_fpfh(const double *xyz, const double *nrm, int64_t n, int mode,
                     int knn, double radius, double *feat33);

typedef struct {
    double division_factor;
    int32_t use_absolute_scale;
    int32_t decrease_mu;
    double maximum_correspondence_distance;
    int32_t iteration_number;
    double tuple_scale;
    int32_t maximum_tuple_count;
    int32_t tuple_test;
    uint64_t seed;
} orc_fgr_option;

/* registration_fgr_based_on_feature_matching (A.8). corr capacity ns*2 or NULL */
int orc_registration_fgr(const double *src_xyz, const double *src_feat, int64_t ns,
                         const double *tgt_xyz, const double *tgt_feat, int64_t nt,
                         const orc_fgr_option *opt, orc_result *out,
                         int64_t *n_cross, int64_t *n_tuple_corr);

/* evaluate_registration / get_information_matrix_from_point_clouds            */
int orc_evaluate_registration(const double *src_xyz, int64_t ns, const double *tgt_xyz,
                              int64_t nt, double max_dist, const double *T,
                              orc_result *out, int32_t *corr);
int orc_information_matrix(const double *src_xyz, int64_t ns, const double *tgt_xyz,
                           int64_t nt, double max_dist, const double *T,
                           double *info36);

#ifdef __cplusplus
}
#endif
#endif
