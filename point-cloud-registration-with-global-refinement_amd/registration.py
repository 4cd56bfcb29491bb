"""Stand-ins for ``o3d.pipelines.registration`` as used on the reference's hot path.

``registration_generalized_icp`` ALL_FUNCTIONS.py:304-311 / 2_MGICP...py:155-162; ``registration_icp`` with
``TransformationEstimationForGeneralizedICP`` :220-226; ``L1Loss`` :284, ``GMLoss`` :219;
``ICPConvergenceCriteria`` :309-311; ``evaluate_registration`` :809-820;
``get_information_matrix_from_point_clouds`` :327-331; ``compute_fpfh_feature`` :186-187;
``FastGlobalRegistrationOption`` :189-196; ``registration_fgr_based_on_feature_matching`` :198-202.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .geometry import PointCloud, _ptr, _torch


class RobustKernel:
    kind = _lib.LOSS_L2
    k = 1.0


class L2Loss(RobustKernel):
    pass


class L1Loss(RobustKernel):
    kind = _lib.LOSS_L1


class GMLoss(RobustKernel):
    kind = _lib.LOSS_GM

    def __init__(self, k: float = 1.0):
        self.k = float(k)


class TransformationEstimationForGeneralizedICP:
    def __init__(self, kernel: RobustKernel | None = None, epsilon: float = 1e-3):
        self.kernel = kernel if kernel is not None else L2Loss()
        self.epsilon = float(epsilon)


class ICPConvergenceCriteria:
    def __init__(self, relative_fitness: float = 1e-6, relative_rmse: float = 1e-6, max_iteration: int = 30):
        self.relative_fitness = float(relative_fitness)
        self.relative_rmse = float(relative_rmse)
        self.max_iteration = int(max_iteration)


class FastGlobalRegistrationOption:
    def __init__(self, division_factor=1.4, use_absolute_scale=False, decrease_mu=False,
                 maximum_correspondence_distance=0.025, iteration_number=64, tuple_scale=0.95,
                 maximum_tuple_count=1000, tuple_test=True, seed=None):
        self.division_factor = float(division_factor)
        self.use_absolute_scale = bool(use_absolute_scale)
        self.decrease_mu = bool(decrease_mu)
        self.maximum_correspondence_distance = float(maximum_correspondence_distance)
        self.iteration_number = int(iteration_number)
        self.tuple_scale = float(tuple_scale)
        self.maximum_tuple_count = int(maximum_tuple_count)
        self.tuple_test = bool(tuple_test)
        self.seed = seed


class RegistrationResult:
    """``transformation`` (4x4 float64, source->target), ``fitness``, ``inlier_rmse``, ``correspondence_set``."""

    def __init__(self, transformation=None, fitness=0.0, inlier_rmse=0.0, correspondence_set=None, iterations=0,
                 converged=False, scales=None):
        self.transformation = np.eye(4) if transformation is None else np.array(transformation, dtype=np.float64).reshape(4, 4)
        self.fitness = float(fitness)
        self.inlier_rmse = float(inlier_rmse)
        self._corr = correspondence_set           # torch int32 (K,2) on device, materialised lazily
        self.iterations = int(iterations)
        self.converged = bool(converged)
        self.scales = scales or []

    @property
    def correspondence_set(self):
        if self._corr is None:
            return np.zeros((0, 2), np.int32)
        if not isinstance(self._corr, np.ndarray):
            self._corr = self._corr.cpu().numpy()
        return self._corr

    def __repr__(self):
        return (f"RegistrationResult with fitness={self.fitness:e}, inlier_rmse={self.inlier_rmse:e}, "
                f"and correspondence_set size of {len(self.correspondence_set)}")


class Feature:
    """``o3d.pipelines.registration.Feature``: ``data`` is (dimension, N) float64 like Open3D's."""

    def __init__(self, dev):
        self._dev = dev            # torch (N,33) float32 cuda

    @property
    def data(self):
        return self._dev.detach().cpu().numpy().astype(np.float64).T

    def dimension(self):
        return int(self._dev.shape[1])

    def num(self):
        return int(self._dev.shape[0])


def _params(estimation, criteria) -> _lib.PcrGicpParams:
    return _lib.PcrGicpParams(int(estimation.kernel.kind), float(estimation.kernel.k), float(estimation.epsilon),
                              float(criteria.relative_fitness), float(criteria.relative_rmse), int(criteria.max_iteration))


def _result(r: _lib.PcrResult, corr=None, scales=None) -> RegistrationResult:
    if corr is not None:
        corr = corr[: r.n_correspondences]
    return RegistrationResult(np.array(r.transformation, dtype=np.float64).reshape(4, 4), r.fitness, r.inlier_rmse, corr,
                              r.iterations, bool(r.converged), scales)


def _T(init):
    T = np.ascontiguousarray(np.asarray(init, dtype=np.float64).reshape(4, 4))
    return T, T.ctypes.data_as(C.POINTER(C.c_double))


def _ensure_gicp_normals(pc: PointCloud):
    """Open3D InitializePointCloudForGeneralizedICP: clouds without covariances and without normals get
    ``estimate_normals()`` with the default search (KNN 30) first."""
    if not pc.has_normals():
        pc = pc.__deepcopy__({})
        pc.estimate_normals()
    return pc


def registration_generalized_icp(source: PointCloud, target: PointCloud, max_correspondence_distance: float,
                                 init=np.eye(4), estimation_method=None, criteria=None) -> RegistrationResult:
    estimation = estimation_method or TransformationEstimationForGeneralizedICP()
    criteria = criteria or ICPConvergenceCriteria()
    ctx = _lib.Context.current()
    torch = _torch()
    if max_correspondence_distance <= 0:
        raise RuntimeError("Invalid max_correspondence_distance.")
    ns, nt = len(source), len(target)
    corr = torch.empty((max(ns, 1), 2), dtype=torch.int32, device="cuda")
    res = _lib.PcrResult()
    T, Tp = _T(init)
    p = _params(estimation, criteria)
    if source.has_covariances() and target.has_covariances():
        # Open3D InitializePointCloudForGeneralizedICP: clouds that already carry covariances keep them untouched
        ctx.check(ctx.lib.pcr_registration_generalized_icp_cov(
            ctx.handle, _ptr(source.device_xyz()), _ptr(source._cov), C.c_int64(ns), _ptr(target.device_xyz()), _ptr(target._cov),
            C.c_int64(nt), C.c_double(max_correspondence_distance), Tp, C.byref(p), C.byref(res), _ptr(corr)),
            "registration_generalized_icp")
        return _result(res, corr)
    if source.has_covariances() != target.has_covariances():
        raise RuntimeError("registration_generalized_icp: either both clouds carry covariances or neither (mixed case not on the MI355X path)")
    source = _ensure_gicp_normals(source); target = _ensure_gicp_normals(target)
    ctx.check(ctx.lib.pcr_registration_generalized_icp(
        ctx.handle, _ptr(source.device_xyz()), _ptr(source.device_normals()), C.c_int64(ns), _ptr(target.device_xyz()),
        _ptr(target.device_normals()), C.c_int64(nt), C.c_double(max_correspondence_distance), Tp, C.byref(p), C.byref(res),
        _ptr(corr)), "registration_generalized_icp")
    return _result(res, corr)


def registration_icp(source, target, max_correspondence_distance, init=np.eye(4), estimation_method=None, criteria=None):
    """Only the GICP estimator is on the hot path (ALL_FUNCTIONS.py:220-226)."""
    if not isinstance(estimation_method, TransformationEstimationForGeneralizedICP):
        raise RuntimeError("registration_icp: only TransformationEstimationForGeneralizedICP is implemented on the MI355X path")
    return registration_generalized_icp(source, target, max_correspondence_distance, init, estimation_method, criteria)


def multiscale_gicp(source: PointCloud, target: PointCloud, voxel_sizes, max_correspondence_distances, init=np.eye(4),
                    estimation_method=None, criteria=None, nb_neighbors: int = 30, std_ratio: float = 1.0,
                    normal_knn: int = 20) -> RegistrationResult:
    """Device-resident body of the reference's ``Multiscale_GICP`` loops (one C-ABI call for all scales)."""
    estimation = estimation_method or TransformationEstimationForGeneralizedICP()
    criteria = criteria or ICPConvergenceCriteria()
    ctx = _lib.Context.current()
    torch = _torch()
    vox = np.ascontiguousarray(np.asarray(voxel_sizes, dtype=np.float64).reshape(-1))
    dst = np.ascontiguousarray(np.asarray(max_correspondence_distances, dtype=np.float64).reshape(-1))
    if vox.size != dst.size or vox.size < 1:
        raise RuntimeError("multiscale_gicp: voxel_sizes and max_correspondence_distances must have the same length >= 1")
    ns, nt = len(source), len(target)
    recs = (_lib.PcrScaleRecord * vox.size)()
    corr = torch.empty((max(ns, 1), 2), dtype=torch.int32, device="cuda")
    T, Tp = _T(init)
    p = _params(estimation, criteria)
    sn = source.device_normals() if source.has_normals() else None
    tn = target.device_normals() if target.has_normals() else None
    ctx.check(ctx.lib.pcr_multiscale_gicp(
        ctx.handle, _ptr(source.device_xyz()), _ptr(sn), C.c_int64(ns), _ptr(target.device_xyz()), _ptr(tn), C.c_int64(nt),
        vox.ctypes.data_as(C.POINTER(C.c_double)), dst.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(vox.size),
        C.c_int(nb_neighbors), C.c_double(std_ratio), C.c_int(normal_knn), Tp, C.byref(p), recs, _ptr(corr)),
        "multiscale_gicp")
    scales = _scale_dicts(recs, vox, dst)
    return _result(recs[vox.size - 1].icp, corr, scales)


_fgr_seed_counter = [0x9E3779B97F4A7C15]


def _scale_dicts(recs, vox, dst):
    return [dict(voxel=float(vox[i]), max_dist=float(dst[i]), n_voxel=tuple(recs[i].n_voxel), n_clean=tuple(recs[i].n_clean),
                 iterations=int(recs[i].icp.iterations), fitness=recs[i].icp.fitness, inlier_rmse=recs[i].icp.inlier_rmse,
                 converged=bool(recs[i].icp.converged), n_corr=int(recs[i].icp.n_correspondences),
                 T=np.array(recs[i].icp.transformation).reshape(4, 4)) for i in range(len(vox))]


def _fgr_params(voxel_size: float, use_absolute_scale: bool, n_pontos: int, seed) -> _lib.PcrFgrParams:
    """The constants of ``registro_FGR`` (ALL_FUNCTIONS.py:181-196 / 1_FGR...py:44-59) for one voxel size."""
    if seed is None:                       # Open3D draws from std::random_device; here a process-local counter
        _fgr_seed_counter[0] = (_fgr_seed_counter[0] * 6364136223846793005 + 1442695040888963407) & (2 ** 64 - 1)
        seed = _fgr_seed_counter[0]
    opt = _lib.PcrFgrOption(1.4, int(bool(use_absolute_scale)), 1, 2 * voxel_size, 300, 0.95, -1 if n_pontos is None else int(n_pontos * 0.2), 1, int(seed) & (2 ** 64 - 1))
    return _lib.PcrFgrParams(2 * voxel_size, 20, 10 * voxel_size, 200, opt)


def registro_fgr(source: PointCloud, target: PointCloud, voxel_size: float, use_absolute_scale: bool = True, seed=None) -> RegistrationResult:
    """``registro_FGR`` (ALL_FUNCTIONS.py:178-203; script 1:41-66 with ``use_absolute_scale=False``) as ONE library call:
    hybrid normals, FPFH, feature matching, tuple test, GNC optimisation and the final evaluation, every cloud sorted and
    indexed once.  Like the reference it leaves normals on ``source`` and ``target``."""
    ctx = _lib.Context.current()
    torch = _torch()
    ns, nt = len(source), len(target)
    fp = _fgr_params(voxel_size, use_absolute_scale, int((ns + nt) / 2), seed)
    corr = torch.empty((max(ns, 1), 2), dtype=torch.int32, device="cuda")
    sno = torch.empty((max(ns, 1), 3), dtype=torch.float32, device="cuda")
    tno = torch.empty((max(nt, 1), 3), dtype=torch.float32, device="cuda")
    res = _lib.PcrResult()
    ctx.check(ctx.lib.pcr_registro_fgr(ctx.handle, _ptr(source.device_xyz()), _ptr(source.device_normals() if source.has_normals() else None), C.c_int64(ns),
                                       _ptr(target.device_xyz()), _ptr(target.device_normals() if target.has_normals() else None), C.c_int64(nt),
                                       C.byref(fp), _ptr(sno), _ptr(tno), C.byref(res), _ptr(corr)), "registro_FGR")
    if ns:
        source._nrm = sno[:ns]
    if nt:
        target._nrm = tno[:nt]
    return _result(res, corr)


def default_group(points_per_cloud: float) -> int:
    """Pairs per lockstep GICP group for clouds of this size: about 1.2M points per group, at most 8 (the by-value argument batch
    of ``k_icp_fused_b``; larger groups take the by-pointer kernel) -- 24 up to 40k points per cloud (round 5, below).  Measured on one
    MI355X, 4 groups in flight, 3-scale GICP stage, pair by pair -> groups: 200k-point pairs 340 -> 530 pairs/s (groups of 6), 100k
    440 -> 850 (8), 50k 540 -> 1360 (8) (round 3/4 numbers; HISTORY.md)."""
    if points_per_cloud >= 400_000:           # PCR_GROUP_FORMS_MAX_POINTS: such pairs run one by one with the single-pair kernel forms
        return 1
    g = int(round(1_200_000 / max(float(points_per_cloud), 1.0)))
    # Round 5: up to 24 pairs (the library's limit) for clouds up to 40k points -- the reference's NCLT scans.  The by-pointer kernel of groups beyond 8
    # holds 117 VGPRs now (146 before: one workgroup per CU, which is why 8 was the rule) and such pairs take 512-point tiles: script-2 stage on the
    # shipped-size scans, tiled and in three random orders, 8 / 12 / 16 / 24 pairs x 4 groups in flight = 1970 / 2530 / 2390 / 2920 pairs/s
    # (tools/gicp_nclt_sweep.py).  From 50k points the group size makes no difference (8 / 12 / 24: 1892 / 1874 / 1918 pairs/s at 50k, 1197 / 1199 / 1196 at 100k).
    return max(1, min(24 if points_per_cloud <= 40_000 else 8, g))


def default_fgr_group(points_per_cloud: float) -> int:
    """Pairs per lockstep ``registro_FGR`` group (``pcr_pairs_plan.fgr_group``): NCLT-size pairs are ~125 small dependent launches and 8
    host waits each, which a group shares (24 pairs up to 30k points per cloud -- measured on the shipped-size NCLT scans with the K = 64
    feature screen: 8 / 12 / 16 / 24 pairs x 4 groups in flight = 1762 / 1775 / 1841 / 1885 pairs/s; 16 at 20k-40k by the older rule); from ~70k
    points the tile-pruned feature search takes over and pairs run one by one."""
    if points_per_cloud >= 70_000:
        return 1
    if points_per_cloud <= 30_000:
        return 24
    return max(1, min(16, int(round(320_000 / max(float(points_per_cloud), 1.0)))))


def balanced_group(group: int, n_pairs: int, inflight: int) -> int:
    """Group size near ``group`` for which the groups of a batch of ``n_pairs`` fill whole rounds of ``inflight`` workers: 96 pairs in
    groups of 16 are 6 groups -- a round of 4 and a round of 2 -- in groups of 12 they are two full rounds (NCLT-size pairs, 5-scale
    GICP stage: 1930 -> 2138 pairs/s)."""
    if group <= 1 or inflight <= 1 or n_pairs <= group * inflight:
        return max(1, group)
    n_groups = -(-n_pairs // group)
    n_groups = -(-n_groups // inflight) * inflight
    return max(1, -(-n_pairs // n_groups))


def register_pairs_plan(pairs, stage: str = "gicp", voxel_sizes=None, max_correspondence_distances=None, estimation_method=None, criteria=None,
                        nb_neighbors: int = 30, std_ratio: float = 1.0, normal_knn: int = 20, inflight: int = 3, with_correspondences: bool = True,
                        fgr_voxel_size: float = 0.1, fgr_use_absolute_scale: bool = True, fgr_seed=None, radius_rule: str = "given",
                        prior_from_fgr: bool = False, info_max_dist: float = 0.0, keep_fgr_normals: bool = False, group=1, pair_forms=None, fgr_group=None) -> list:
    """The per-pair loops of the reference as ONE library call (``pcr_register_pairs_plan``): `pairs` = [(source PointCloud,
    target PointCloud, initial 4x4 or None), ...].

    stage "fgr"       = script 1 (1_FGR...py:134-147): ``registro_FGR`` per pair;
    stage "gicp"      = script 2 (2_MGICP...py:187-214): ``Multiscale_GICP`` from the given initial pose;
    stage "fgr+gicp"  = ``Coarse_to_fine_FGR_M_GICP`` / ``full_registration`` (ALL_FUNCTIONS.py:317-332, 349-392).
    radius_rule "af"  = search radii ``radius_from_cloud_pair * 2**-i`` per pair (ALL_FUNCTIONS.py:277-278) instead of the given list.
    ``group`` > 1 (stages "gicp" and "fgr+gicp"): that many consecutive pairs run in LOCKSTEP through the same GICP launches (preprocessing
    batched over clouds and scales, one GICP loop per scale for the whole group; same per-pair arithmetic; with "fgr+gicp" the worker runs
    registro_FGR pair by pair first); ``inflight`` counts groups.
    ``group=None`` picks by cloud size (``default_group``).
    ``fgr_group`` (stages with FGR; None = ``default_fgr_group`` of the mean cloud size, 1 = pair by pair): that many consecutive pairs go
    through ``registro_FGR`` in lockstep (``pcr_pairs_plan.fgr_group``); the same bits per pair either way.
    ``pair_forms`` (default: on exactly when ``group`` is None): the kernel forms of the GICP stage go by the PAIR alone
    (``pcr_pairs_plan.pair_forms``), so a pair's pose bits are the same in every batch, group size and shard of a multi-GPU run.
    The library keeps ``inflight`` pairs in flight on the current device.  Returns RegistrationResults in input order (for
    stages with FGR the FGR result is attached as ``.fgr``; ``.information`` when ``info_max_dist > 0``)."""
    estimation = estimation_method or TransformationEstimationForGeneralizedICP()
    criteria = criteria or ICPConvergenceCriteria()
    torch = _torch()
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible to torch: the MI355X registration path cannot run (no CPU fallback)")
    stage_id = {"gicp": _lib.STAGE_GICP, "fgr": _lib.STAGE_FGR, "fgr+gicp": _lib.STAGE_FGR_GICP}[stage]
    do_fgr, do_gicp = stage_id != _lib.STAGE_GICP, stage_id != _lib.STAGE_FGR
    vox = np.ascontiguousarray(np.asarray(voxel_sizes if voxel_sizes is not None else [], dtype=np.float64).reshape(-1))
    rule = {"given": 0, "af": 1}[radius_rule]
    dst = np.ascontiguousarray(np.asarray(max_correspondence_distances if max_correspondence_distances is not None else np.zeros(vox.size), dtype=np.float64).reshape(-1))
    if do_gicp and (vox.size < 1 or vox.size > 8 or dst.size != vox.size):
        raise RuntimeError("register_pairs: voxel_sizes and max_correspondence_distances must have the same length (1..8)")
    n = len(pairs)
    if n == 0:
        return []
    if pair_forms is None:
        pair_forms = group is None
    if fgr_group is None:
        fgr_group = default_fgr_group(float(np.mean([len(s_) + len(t_) for s_, t_, _ in pairs])) / 2) if do_fgr else 1
        if stage == "fgr":
            fgr_group = balanced_group(fgr_group, n, int(inflight))
    if group is None:
        mean_pts = float(np.mean([len(s_) + len(t_) for s_, t_, _ in pairs])) / 2
        group = balanced_group(default_group(mean_pts), n, int(inflight))
    # FGR groups and GICP groups of the same size (NCLT-size clouds: 24 and 24): ONE library call, every worker runs registro_FGR and the GICP of
    # its group back to back -- the same poses, 1170 -> 1220 pairs/s on the shipped scans (PCR_PLAN_ONE_CALL=0: two passes as below)
    import os as _os
    one_call = (fgr_group == group and _os.environ.get("PCR_PLAN_ONE_CALL", "1") != "0") or _os.environ.get("PCR_PLAN_ONE_CALL") == "2"
    if stage == "fgr+gicp" and group > 1 and n > 1 and not one_call:
        # Two passes over the batch instead of FGR -> GICP pair by pair: registro_FGR is a chain of ~150 small launches with a few host
        # waits and wants MANY pairs in flight (20k-point pairs: 230 / 590 / 700 pairs/s with 1 / 4 / 8), the GICP wants lockstep groups.
        # Same arithmetic as the single call (same seeds per pair, the FGR normals as the orientation prior, the same radius rule).
        def view(pc):                         # a cloud object of our own on the caller's tensors: the FGR normals land here, not on the caller's cloud
            v = PointCloud(); v._xyz = pc.device_xyz(); v._nrm = pc.device_normals() if pc.has_normals() else None
            return v
        views = [(view(s_), view(t_), None) for s_, t_, _ in pairs]
        fg = register_pairs_plan(views, "fgr", None, None, estimation, criteria, nb_neighbors, std_ratio, normal_knn, max(int(inflight), 8) if fgr_group <= 1 else max(int(inflight), 4), False,
                                 fgr_voxel_size, fgr_use_absolute_scale, fgr_seed, "given", False, 0.0, True, 1, fgr_group=fgr_group)
        second = [((vs if prior_from_fgr else s_), (vt if prior_from_fgr else t_), f.transformation) for (vs, vt, _), (s_, t_, _), f in zip(views, pairs, fg)]
        out = register_pairs_plan(second, "gicp", voxel_sizes, max_correspondence_distances, estimation, criteria, nb_neighbors, std_ratio, normal_knn, inflight,
                                  with_correspondences, radius_rule=radius_rule, info_max_dist=info_max_dist, group=group, pair_forms=pair_forms)
        for r, f, (vs, vt, _), (s_, t_, _) in zip(out, fg, views, pairs):
            r.fgr = f
            if keep_fgr_normals:              # the reference's side effect: both inputs gain normals
                if len(s_):
                    s_._nrm = vs._nrm
                if len(t_):
                    t_._nrm = vt._nrm
        return out
    arr = (_lib.PcrPairEx * n)()
    keep = []                                   # device tensors and record arrays must outlive the call
    # the correspondence sets of the batch in ONE allocation (a view per pair): 48 allocator calls per bench step were a third of the ~1 ms a step
    # spent outside the library call
    corr_rows = [max(len(src), 1) for src, _, _ in pairs]
    corr_all = torch.empty((sum(corr_rows), 2), dtype=torch.int32, device="cuda") if with_correspondences else None
    corr_off = 0
    for k, (src, tgt, init) in enumerate(pairs):
        recs = (_lib.PcrScaleRecord * max(vox.size, 1))()
        corr = corr_all[corr_off: corr_off + corr_rows[k]] if with_correspondences else None
        corr_off += corr_rows[k]
        sx, tx = src.device_xyz(), tgt.device_xyz()
        sn = src.device_normals() if src.has_normals() else None
        tn = tgt.device_normals() if tgt.has_normals() else None
        sno = torch.empty((max(len(src), 1), 3), dtype=torch.float32, device="cuda") if (do_fgr and keep_fgr_normals) else None
        tno = torch.empty((max(len(tgt), 1), 3), dtype=torch.float32, device="cuda") if (do_fgr and keep_fgr_normals) else None
        keep.append((recs, corr, sx, tx, sn, tn, sno, tno))
        a = arr[k].base
        a.src_xyz = sx.data_ptr(); a.src_normals = sn.data_ptr() if sn is not None else None; a.n_src = len(src)
        a.tgt_xyz = tx.data_ptr(); a.tgt_normals = tn.data_ptr() if tn is not None else None; a.n_tgt = len(tgt)
        a.init_T[:] = list(np.asarray(np.eye(4) if init is None else init, dtype=np.float64).reshape(16))
        a.records = C.cast(recs, C.POINTER(_lib.PcrScaleRecord)); a.correspondences = corr.data_ptr() if corr is not None else None
        arr[k].src_normals_out = sno.data_ptr() if sno is not None else None
        arr[k].tgt_normals_out = tno.data_ptr() if tno is not None else None
    p = _params(estimation, criteria)
    plan = _lib.PcrPairsPlan()
    plan.stage = stage_id
    fp = None
    if do_fgr:
        # maximum_tuple_count = int(0.2 * n_pontos) depends on the pair's sizes (ALL_FUNCTIONS.py:179,196): the library applies the rule per pair
        n_pontos = None
        fp = _fgr_params(fgr_voxel_size, fgr_use_absolute_scale, n_pontos, fgr_seed)
        plan.fgr = C.pointer(fp)
    plan.voxel_sizes = vox.ctypes.data_as(C.POINTER(C.c_double)); plan.max_distances = dst.ctypes.data_as(C.POINTER(C.c_double))
    plan.n_scales = int(vox.size); plan.radius_rule = rule
    plan.sor_k = int(nb_neighbors); plan.sor_std = float(std_ratio); plan.normal_k = int(normal_knn)
    plan.gicp = C.pointer(p); plan.gicp_prior_from_fgr = int(bool(prior_from_fgr)); plan.info_max_dist = float(info_max_dist); plan.inflight = int(inflight); plan.group = int(group); plan.pair_forms = int(pair_forms); plan.fgr_group = int(fgr_group)
    dev = torch.cuda.current_device()
    rc = lib.pcr_register_pairs_plan(C.c_int(dev), arr, C.c_int(n), C.byref(plan), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    out = []
    empty = torch.empty((0, 2), dtype=torch.int32, device="cuda")
    for k in range(n):
        b = arr[k].base
        if b.status != 0:
            raise RuntimeError(f"register_pairs: pair {k} failed with code {b.status}: {b.error.decode(errors='replace')}")
        recs, corr, sno, tno = keep[k][0], keep[k][1], keep[k][6], keep[k][7]
        if do_gicp:
            used = np.array(arr[k].max_distances[: vox.size]) if rule == 1 else dst
            r = _result(recs[vox.size - 1].icp, corr if corr is not None else empty, _scale_dicts(recs, vox, used))
        else:
            r = _result(arr[k].fgr, corr if corr is not None else empty)
        if do_fgr:
            r.fgr = _result(arr[k].fgr, None)
            if keep_fgr_normals:              # the reference's side effect: both inputs gain normals
                src, tgt, _ = pairs[k]
                if len(src):
                    src._nrm = sno[: len(src)]
                if len(tgt):
                    tgt._nrm = tno[: len(tgt)]
        if info_max_dist > 0:
            r.information = np.array(arr[k].info36, dtype=np.float64).reshape(6, 6)
        out.append(r)
    if rc != 0:
        raise RuntimeError(f"register_pairs failed with code {rc}")
    return out


def register_pairs(pairs, voxel_sizes, max_correspondence_distances, estimation_method=None, criteria=None, nb_neighbors: int = 30,
                   std_ratio: float = 1.0, normal_knn: int = 20, inflight: int = 3, with_correspondences: bool = True, group=1) -> list:
    """The per-pair loop of script 2 (2_MGICP...py:187-214) as ONE library call: every pair gets the body of ``multiscale_gicp``
    from its initial pose; see ``register_pairs_plan``."""
    return register_pairs_plan(pairs, "gicp", voxel_sizes, max_correspondence_distances, estimation_method, criteria, nb_neighbors, std_ratio,
                               normal_knn, inflight, with_correspondences, group=group)


def evaluate_registration(source: PointCloud, target: PointCloud, max_correspondence_distance: float,
                          transformation=np.eye(4)) -> RegistrationResult:
    ctx = _lib.Context.current()
    torch = _torch()
    if max_correspondence_distance <= 0:
        raise RuntimeError("Invalid max_correspondence_distance.")
    ns, nt = len(source), len(target)
    corr = torch.empty((max(ns, 1), 2), dtype=torch.int32, device="cuda")
    res = _lib.PcrResult()
    T, Tp = _T(transformation)
    ctx.check(ctx.lib.pcr_evaluate_registration(ctx.handle, _ptr(source.device_xyz()), C.c_int64(ns), _ptr(target.device_xyz()),
                                                C.c_int64(nt), C.c_double(max_correspondence_distance), Tp, C.byref(res),
                                                _ptr(corr)), "evaluate_registration")
    return _result(res, corr)


def get_information_matrix_from_point_clouds(source: PointCloud, target: PointCloud, max_correspondence_distance: float,
                                             transformation) -> np.ndarray:
    ctx = _lib.Context.current()
    info = np.zeros(36, dtype=np.float64)
    T, Tp = _T(transformation)
    ctx.check(ctx.lib.pcr_information_matrix(ctx.handle, _ptr(source.device_xyz()), C.c_int64(len(source)),
                                             _ptr(target.device_xyz()), C.c_int64(len(target)),
                                             C.c_double(max_correspondence_distance), Tp,
                                             info.ctypes.data_as(C.POINTER(C.c_double))), "get_information_matrix_from_point_clouds")
    return info.reshape(6, 6)


def compute_fpfh_feature(cloud: PointCloud, search_param) -> Feature:
    ctx = _lib.Context.current()
    torch = _torch()
    if not cloud.has_normals():
        raise RuntimeError("Failed because input point cloud has no normal.")
    kind, knn, radius = search_param._spec()
    n = len(cloud)
    feat = torch.zeros((max(n, 1), 33), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.pcr_compute_fpfh_feature(ctx.handle, _ptr(cloud.device_xyz()), _ptr(cloud.device_normals()), C.c_int64(n),
                                               C.c_int(kind), C.c_int(knn), C.c_double(radius), _ptr(feat)), "compute_fpfh_feature")
    return Feature(feat[:n].contiguous())


def registration_fgr_based_on_feature_matching(source: PointCloud, target: PointCloud, source_feature: Feature,
                                               target_feature: Feature, option: FastGlobalRegistrationOption | None = None):
    ctx = _lib.Context.current()
    torch = _torch()
    option = option or FastGlobalRegistrationOption()
    seed = option.seed
    if seed is None:                       # Open3D draws from std::random_device; here a process-local counter
        _fgr_seed_counter[0] = (_fgr_seed_counter[0] * 6364136223846793005 + 1442695040888963407) & (2 ** 64 - 1)
        seed = _fgr_seed_counter[0]
    o = _lib.PcrFgrOption(option.division_factor, int(option.use_absolute_scale), int(option.decrease_mu),
                          option.maximum_correspondence_distance, option.iteration_number, option.tuple_scale,
                          option.maximum_tuple_count, int(option.tuple_test), int(seed))
    ns, nt = len(source), len(target)
    corr = torch.empty((max(ns, 1), 2), dtype=torch.int32, device="cuda")
    res = _lib.PcrResult()
    ctx.check(ctx.lib.pcr_registration_fgr(ctx.handle, _ptr(source.device_xyz()), _ptr(source_feature._dev), C.c_int64(ns),
                                           _ptr(target.device_xyz()), _ptr(target_feature._dev), C.c_int64(nt), C.byref(o),
                                           C.byref(res), _ptr(corr)), "registration_fgr_based_on_feature_matching")
    return _result(res, corr)


# pose-graph slice of o3d.pipelines.registration (3_Global_Optimizations...py:292-358; host side, posegraph.py)
from .posegraph import (GlobalOptimizationConvergenceCriteria, GlobalOptimizationLevenbergMarquardt,  # noqa: E402,F401
                        GlobalOptimizationOption, PoseGraph, PoseGraphEdge, PoseGraphNode, global_optimization)
