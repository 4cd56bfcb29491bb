"""Host-side mirror of the reference's function surface for the pairwise-registration hot path:
same names, positional order and return objects as ``ALL_FUNCTIONS.py`` (library variants, this
module's top level) and as the private copies pasted into scripts 1 and 2 (``script1`` / ``script2``
namespaces below).  SURVEY.md §8b lists the surface; App. C lists the quirks preserved here.

Everything heavy runs in ``libpcr_hip.so`` on the MI355X.  The multiscale loop body is ONE C-ABI call
(``pcr_multiscale_gicp``) so that the clouds stay resident in HBM across all scales.  The reference's
call-by-call sequences are replayed through the PointCloud stand-ins by the tests (tests/test_gpu_fgr.py,
tests/test_gpu_gicp.py), not here: this module is signatures, constants and one library call each.
"""
from __future__ import annotations

import numpy as np

from . import geometry as _g
from . import registration as _r

KNN_FILTRO = 30          # ALL_FUNCTIONS.py:280, 2_MGICP...py:134
STD_FILTRO = 1.0         # ALL_FUNCTIONS.py:281, 2_MGICP...py:135
KNN_NORMAIS = 20         # ALL_FUNCTIONS.py:301, 2_MGICP...py:152
RETENTION_A, RETENTION_B = 1.18397758, 5.09388767     # ALL_FUNCTIONS.py:241-242: share of the points a voxel grid keeps ~ a * exp(-b * voxel)


# ------------------------------------------------------------------------------- ALL_FUNCTIONS.py variants
def registro_FGR(source, target, voxel_size, _use_absolute_scale=True, seed=None):
    """ALL_FUNCTIONS.py:178-203: normals Hybrid(2v, 20), FPFH Hybrid(10v, 200), FGR option (1.4, absolute scale, decrease_mu, 2v, 300,
    0.95, int(0.2 * n_pontos)), feature-matching FGR.  Mutates ``source``/``target`` (adds normals), like the reference.  One library
    call (``pcr_registro_fgr``: each cloud is sorted and indexed once)."""
    return _r.registro_fgr(source, target, voxel_size, _use_absolute_scale, seed)


def GICP_robusto(source, target, max_corres_dist, initial_T, iterations):
    """ALL_FUNCTIONS.py:211-227: radius-0.20 normals, raw 30-NN covariances (``estimate_covariances()`` default search),
    GICP with ``GMLoss(k=1.0)``.  Mutates ``source``/``target`` (normals and covariances), like the reference."""
    for cloud in (source, target):
        cloud.estimate_normals(_g.KDTreeSearchParamRadius(radius=0.20))
    for cloud in (source, target):
        cloud.estimate_covariances()
    return _r.registration_icp(source, target, max_corres_dist, initial_T, _r.TransformationEstimationForGeneralizedICP(_r.GMLoss(k=1.0)),
                               _r.ICPConvergenceCriteria(max_iteration=iterations))


def amostragem_multiescala_otimizada(nuvem, n_escalas, voxel_inicial, seed=None):
    """ALL_FUNCTIONS.py:233-254: one voxel grid at ``voxel_inicial`` and ``n_escalas - 1`` random subsets of it whose sizes imitate
    coarser voxel grids (retention model a*exp(-b*voxel), a = 1.18397758, b = 5.09388767, normalised as the reference does);
    returned as the reference returns them: the random subsets in reverse order of creation, the voxel cloud last.  ``seed``
    (not in the reference, whose draws come from Open3D's random device) makes the subsets repeatable."""
    base = nuvem.voxel_down_sample(voxel_inicial)
    voxels = voxel_inicial + voxel_inicial * np.arange(n_escalas)                    # the grids being imitated: v, 2v, 3v, ...
    share = RETENTION_A * np.exp(-RETENTION_B * voxels) * len(nuvem.points) / len(base.points)
    share = (share / np.linalg.norm(share))[1:10]                                    # the reference's normalisation; entry 0 is the voxel cloud itself
    subsets = [base.random_down_sample(share[k], seed=None if seed is None else seed + k) for k in range(n_escalas - 1)]
    return subsets[::-1] + [base]


def create_scales(n_scales):
    """ALL_FUNCTIONS.py:260-264: doubling voxel sizes [0.1, 0.2, 0.4, ...] (a product with a power of two is the repeated sum, bit for bit)."""
    return [0.1 * 2.0 ** k for k in range(n_scales)]


def radius_from_cloud_pair(source, target):
    """ALL_FUNCTIONS.py:1092-1101: mean of the cube roots of the two AABB volumes."""
    edge = [np.prod(c.get_max_bound() - c.get_min_bound()) ** (1 / 3) for c in (source, target)]
    return (edge[0] + edge[1]) / 2


def _multiscale(source, target, voxel_sizes, distances, itera_escala, T_ini):
    """The loop body of ALL_FUNCTIONS.py:286-312 / 2_MGICP...py:140-163 for all scales in one library call: per scale voxel grid ->
    outlier filter (30, 1.0) -> 20-NN normals -> GICP with L1Loss and criteria (1e-6, 1e-6, itera_escala), poses chained."""
    est = _r.TransformationEstimationForGeneralizedICP(_r.L1Loss())
    crit = _r.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=itera_escala)
    return _r.multiscale_gicp(source, target, voxel_sizes, distances, T_ini, est, crit, KNN_FILTRO, STD_FILTRO, KNN_NORMAIS)


def Multiscale_GICP(source, target, n_scales, itera_escala, T_ini):
    """ALL_FUNCTIONS.py:272-313: voxels 0.1*2^k coarse-to-fine, search radius = AABB radius * 2^-i."""
    radius = radius_from_cloud_pair(source, target)
    return _multiscale(source, target, create_scales(n_scales)[::-1], [radius * 2 ** -k for k in range(n_scales)], itera_escala, T_ini)


def Coarse_to_fine_FGR_M_GICP(source, target, voxel_size, seed=None):
    """ALL_FUNCTIONS.py:317-332 -> (RegistrationResult, 6x6 information matrix)."""
    coarse = registro_FGR(source, target, voxel_size, seed=seed)
    fine = Multiscale_GICP(source, target, 3, 100, coarse.transformation)                      # 3 scales, 100 iterations each (:320-321)
    return fine, _r.get_information_matrix_from_point_clouds(source, target, voxel_size, fine.transformation)


def calculate_RMSE_and_fitness(lista_nuvens, T_circuito, distancia):
    """ALL_FUNCTIONS.py:801-824: ``evaluate_registration`` of every pose of a circuit -- pose i moves cloud i+1 onto cloud i; with as
    many poses as clouds the last one closes the loop (cloud 0 onto the last cloud); any other count prints the reference's message
    and returns two empty lists."""
    n = len(lista_nuvens)
    if len(T_circuito) == n:
        pairs = [((i + 1) % n, i) for i in range(n)]
    elif len(T_circuito) == n - 1:
        pairs = [(i + 1, i) for i in range(n - 1)]
    else:
        print("The number of clouds and poses are inconsistent")
        return [], []
    results = [_r.evaluate_registration(lista_nuvens[s], lista_nuvens[t], distancia, T) for (s, t), T in zip(pairs, T_circuito)]
    return [r.inlier_rmse for r in results], [r.fitness for r in results]


# ------------------------------------------------------------------------------------ script variants
class script1:
    """Private copies in 1_FGR_pairwise_registration_in_NCLT_dataset.py."""

    @staticmethod
    def registro_FGR(source, target, voxel_size, seed=None):
        """Script 1:41-66: identical to the library version except ``use_absolute_scale=False`` (:54)."""
        return registro_FGR(source, target, voxel_size, _use_absolute_scale=False, seed=seed)


class script2:
    """Private copies in 2_MGICP_refinement_in_NCLT_dataset.py."""

    @staticmethod
    def create_scales(n_scales):
        """Script 2:102-106: linear voxel sizes, already coarse-to-fine."""
        return [0.1 + 0.1 * k for k in reversed(range(n_scales))]

    @staticmethod
    def max_correspondence_distances(scales):
        """Script 2:112-120 (defined for 3, 4 or 5 scales only; other counts raise like the reference's
        UnboundLocalError, here as a ValueError subclass-compatible NameError)."""
        factors = {3: (3, 2, 1), 4: (3, 2.5, 2, 1), 5: (3, 2.5, 2, 1.5, 1)}.get(len(scales))
        if factors is None:
            raise UnboundLocalError("local variable 'max_correspondence_distances' referenced before assignment")
        return [f * v for f, v in zip(factors, scales)]

    @staticmethod
    def Multiscale_GICP(source, target, n_scales, itera_escala, T_ini):
        """Script 2:128-164."""
        voxel_sizes = script2.create_scales(n_scales)
        search_distances = script2.max_correspondence_distances(voxel_sizes)
        return _multiscale(source, target, voxel_sizes, search_distances, itera_escala, T_ini)
