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


# ------------------------------------------------------------------------------- ALL_FUNCTIONS.py variants
def registro_FGR(source, target, voxel_size, _use_absolute_scale=True, seed=None):
    """ALL_FUNCTIONS.py:178-203: normals Hybrid(2v, 20), FPFH Hybrid(10v, 200), FGR option (1.4, absolute scale, decrease_mu, 2v, 300,
    0.95, int(0.2 * n_pontos)), feature-matching FGR.  Mutates ``source``/``target`` (adds normals), like the reference.  One library
    call (``pcr_registro_fgr``: each cloud is sorted and indexed once)."""
    return _r.registro_fgr(source, target, voxel_size, _use_absolute_scale, seed)


def GICP_robusto(source, target, max_corres_dist, initial_T, iterations):
    """ALL_FUNCTIONS.py:211-227: radius-0.20 normals, raw 30-NN covariances (``estimate_covariances()`` default search),
    GICP with ``GMLoss(k=1.0)``.  Mutates ``source``/``target`` (normals and covariances), like the reference."""
    kd_tree_normais = _g.KDTreeSearchParamRadius(radius=0.20)
    source.estimate_normals(kd_tree_normais)
    target.estimate_normals(kd_tree_normais)
    source.estimate_covariances()
    target.estimate_covariances()
    loss = _r.GMLoss(k=1.0)
    return _r.registration_icp(source, target, max_corres_dist, initial_T,
                               _r.TransformationEstimationForGeneralizedICP(loss),
                               _r.ICPConvergenceCriteria(max_iteration=iterations))


def amostragem_multiescala_otimizada(nuvem, n_escalas, voxel_inicial, seed=None):
    """ALL_FUNCTIONS.py:233-254: one voxel grid at ``voxel_inicial`` and ``n_escalas - 1`` random subsets of it whose sizes imitate
    coarser voxel grids (retention model a*exp(-b*voxel), a = 1.18397758, b = 5.09388767, normalised as the reference does);
    returned as the reference returns them: the random subsets in reverse order of creation, the voxel cloud last.  ``seed``
    (not in the reference, whose draws come from Open3D's random device) makes the subsets repeatable."""
    nuvem_amostrada_inicial = nuvem.voxel_down_sample(voxel_inicial)
    total_pts = len(np.asarray(nuvem.points))
    pts_inici = len(np.asarray(nuvem_amostrada_inicial.points))
    escalas = np.asarray([voxel_inicial + voxel_inicial * i for i in range(n_escalas)])
    a, b = 1.18397758, 5.09388767
    porcentagens = a * np.exp(-b * escalas)
    porcentagens_escalonadas = porcentagens * total_pts / pts_inici
    porcentagens_normalizadas = porcentagens_escalonadas / np.linalg.norm(porcentagens_escalonadas)
    porcentagens_normalizadas = porcentagens_normalizadas[1:10]
    lista_nuvens_amostradas = []
    for i in range(n_escalas - 1):
        lista_nuvens_amostradas.append(nuvem_amostrada_inicial.random_down_sample(porcentagens_normalizadas[i], seed=None if seed is None else seed + i))
    lista_nuvens_amostradas.insert(0, nuvem_amostrada_inicial)
    return list(reversed(lista_nuvens_amostradas))


def create_scales(n_scales):
    """ALL_FUNCTIONS.py:260-264: doubling voxel sizes [0.1, 0.2, 0.4, ...]."""
    voxel_radius = [0.1]
    for i in range(n_scales - 1):
        voxel_radius.append(voxel_radius[-1] + voxel_radius[-1])
    return voxel_radius


def radius_from_cloud_pair(source, target):
    """ALL_FUNCTIONS.py:1092-1101: mean of the cube roots of the two AABB volumes."""
    dif_1 = source.get_max_bound() - source.get_min_bound()
    dif_2 = target.get_max_bound() - target.get_min_bound()
    rad_1 = (dif_1[0] * dif_1[1] * dif_1[2]) ** (1 / 3)
    rad_2 = (dif_2[0] * dif_2[1] * dif_2[2]) ** (1 / 3)
    return (rad_1 + rad_2) / 2


def _multiscale(source, target, voxel_sizes, distances, itera_escala, T_ini):
    """The loop body of ALL_FUNCTIONS.py:286-312 / 2_MGICP...py:140-163 for all scales in one library call: per scale voxel grid ->
    outlier filter (30, 1.0) -> 20-NN normals -> GICP with L1Loss and criteria (1e-6, 1e-6, itera_escala), poses chained."""
    est = _r.TransformationEstimationForGeneralizedICP(_r.L1Loss())
    crit = _r.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=itera_escala)
    return _r.multiscale_gicp(source, target, voxel_sizes, distances, T_ini, est, crit, KNN_FILTRO, STD_FILTRO, KNN_NORMAIS)


def Multiscale_GICP(source, target, n_scales, itera_escala, T_ini):
    """ALL_FUNCTIONS.py:272-313: voxels 0.1*2^k coarse-to-fine, search radius = AABB radius * 2^-i."""
    voxel_sizes = create_scales(n_scales)
    voxel_sizes.reverse()
    max_correspondence_distance = radius_from_cloud_pair(source, target)
    max_correspondence_distances = [max_correspondence_distance * (2 ** (-i)) for i in range(n_scales)]
    return _multiscale(source, target, voxel_sizes, max_correspondence_distances, itera_escala, T_ini)


def Coarse_to_fine_FGR_M_GICP(source, target, voxel_size, seed=None):
    """ALL_FUNCTIONS.py:317-332 -> (RegistrationResult, 6x6 information matrix)."""
    result_FGR = registro_FGR(source, target, voxel_size, seed=seed)
    n_scales = 3
    itera_escala = 100
    T_ini = result_FGR.transformation
    result_M_GICP = Multiscale_GICP(source, target, n_scales, itera_escala, T_ini)
    information_matrix = _r.get_information_matrix_from_point_clouds(source, target, voxel_size, result_M_GICP.transformation)
    return result_M_GICP, information_matrix


def calculate_RMSE_and_fitness(lista_nuvens, T_circuito, distancia):
    """ALL_FUNCTIONS.py:801-824."""
    n_nuvens = len(lista_nuvens)
    n_T = len(T_circuito)
    lista_RMSE, lista_fitness = [], []
    if n_nuvens == n_T:
        for i in range(n_nuvens):
            src = lista_nuvens[i + 1] if i < n_nuvens - 1 else lista_nuvens[0]
            result = _r.evaluate_registration(src, lista_nuvens[i], distancia, T_circuito[i])
            lista_RMSE.append(result.inlier_rmse)
            lista_fitness.append(result.fitness)
    elif n_nuvens - 1 == n_T:
        for i in range(n_T):
            result = _r.evaluate_registration(lista_nuvens[i + 1], lista_nuvens[i], distancia, T_circuito[i])
            lista_RMSE.append(result.inlier_rmse)
            lista_fitness.append(result.fitness)
    else:
        print("The number of clouds and poses are inconsistent")
    return lista_RMSE, lista_fitness


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
        voxel_radius = 0.1
        voxel_radius = [voxel_radius + (0.1 * i) for i in range(n_scales)]
        voxel_radius.reverse()
        return voxel_radius

    @staticmethod
    def max_correspondence_distances(scales):
        """Script 2:112-120 (defined for 3, 4 or 5 scales only; other counts raise like the reference's
        UnboundLocalError, here as a ValueError subclass-compatible NameError)."""
        n_scales = len(scales)
        if n_scales == 3:
            return [3 * scales[0], 2 * scales[1], scales[2]]
        elif n_scales == 4:
            return [3 * scales[0], 2.5 * scales[1], 2 * scales[2], scales[3]]
        elif n_scales == 5:
            return [3 * scales[0], 2.5 * scales[1], 2 * scales[2], 1.5 * scales[3], scales[4]]
        raise UnboundLocalError("local variable 'max_correspondence_distances' referenced before assignment")

    @staticmethod
    def Multiscale_GICP(source, target, n_scales, itera_escala, T_ini):
        """Script 2:128-164."""
        voxel_sizes = script2.create_scales(n_scales)
        search_distances = script2.max_correspondence_distances(voxel_sizes)
        return _multiscale(source, target, voxel_sizes, search_distances, itera_escala, T_ini)
