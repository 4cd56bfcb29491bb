"""Pose-graph optimisation on the host (SURVEY.md §8 f-2, App. D "Pose-graph LM"): the slice of
`o3d.pipelines.registration` that `3_Global_Optimizations_in_NCLT_dataset.py:292-358` and `ALL_FUNCTIONS.py:342-394` use
-- `PoseGraph`, `PoseGraphNode`, `PoseGraphEdge`, `GlobalOptimizationOption`, `GlobalOptimizationConvergenceCriteria`,
`GlobalOptimizationLevenbergMarquardt`, `global_optimization` -- plus the two graph builders of the reference.

PARITY UNPINNED.  The arithmetic lives in Open3D (un-vendored, absent here) and the reference ships no output of this
stage, so nothing pins it: this is a restatement of the published method (Choi, Zhou, Koltun, "Robust reconstruction of
indoor scenes", CVPR 2015, as implemented by Open3D's GlobalOptimization): residual of edge (s, t, X) = linearised
6-vector of X^-1 Tt^-1 Ts weighted by the edge's information matrix, a line process l in [0,1] on every `uncertain`
edge with penalty mu (sqrt(l) - 1)^2, Levenberg-Marquardt on the stacked 6n pose increments with the reference node held
fixed, then edges whose line process fell below `edge_prune_threshold` are removed and the graph is optimised again.
Constants recalled from Open3D's defaults are marked [O3D ?].  Tests check properties only (tests/test_posegraph.py).
It stays on the host like the reference's: at 901 nodes it is a sparse 5406-unknown solve.
"""
from __future__ import annotations

import copy

import numpy as np


class PoseGraphNode:
    def __init__(self, pose=None):
        self.pose = np.identity(4) if pose is None else np.array(pose, dtype=np.float64).reshape(4, 4)


class PoseGraphEdge:
    def __init__(self, source_node_id=-1, target_node_id=-1, transformation=None, information=None, uncertain=False, confidence=1.0):
        self.source_node_id = int(source_node_id)
        self.target_node_id = int(target_node_id)
        self.transformation = np.identity(4) if transformation is None else np.array(transformation, dtype=np.float64).reshape(4, 4)
        self.information = np.identity(6) if information is None else np.array(information, dtype=np.float64).reshape(6, 6)
        self.uncertain = bool(uncertain)
        self.confidence = float(confidence)


class PoseGraph:
    def __init__(self):
        self.nodes = []
        self.edges = []


class GlobalOptimizationOption:
    def __init__(self, max_correspondence_distance=0.03, edge_prune_threshold=0.25, preference_loop_closure=1.0, reference_node=-1):
        self.max_correspondence_distance = float(max_correspondence_distance)
        self.edge_prune_threshold = float(edge_prune_threshold)
        self.preference_loop_closure = float(preference_loop_closure)
        self.reference_node = int(reference_node)


class GlobalOptimizationConvergenceCriteria:
    def __init__(self, max_iteration=100, min_relative_increment=1e-6, min_relative_residual_increment=1e-6, min_right_term=1e-6,
                 min_residual=1e-6, max_iteration_lm=20, upper_scale_factor=2.0 / 3.0, lower_scale_factor=1.0 / 3.0):   # [O3D ?]
        self.max_iteration = max_iteration
        self.min_relative_increment = min_relative_increment
        self.min_relative_residual_increment = min_relative_residual_increment
        self.min_right_term = min_right_term
        self.min_residual = min_residual
        self.max_iteration_lm = max_iteration_lm
        self.upper_scale_factor = upper_scale_factor
        self.lower_scale_factor = lower_scale_factor


def _vec6_to_T(v):
    """TransformVector6dToMatrix4d: R = Rz(v2) Ry(v1) Rx(v0), t = v[3:6] (the same convention as the ICP update)."""
    a, b, g = v[0], v[1], v[2]
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    T = np.identity(4)
    T[:3, :3] = [[cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa],
                 [sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa],
                 [-sb, cb * sa, cb * ca]]
    T[:3, 3] = v[3:6]
    return T


def _lin6(M):
    """Linearised 6-vector of a near-identity transform (Choi et al. Eq. 9)."""
    return np.array([(-M[1, 2] + M[2, 1]) / 2.0, (-M[2, 0] + M[0, 2]) / 2.0, (-M[0, 1] + M[1, 0]) / 2.0, M[0, 3], M[1, 3], M[2, 3]])


def _generators():
    G = np.zeros((6, 4, 4))
    G[0, 1, 2] = -1; G[0, 2, 1] = 1
    G[1, 2, 0] = -1; G[1, 0, 2] = 1
    G[2, 0, 1] = -1; G[2, 1, 0] = 1
    G[3, 0, 3] = 1; G[4, 1, 3] = 1; G[5, 2, 3] = 1
    return G


_GEN = _generators()


def _inv(T):
    Ti = np.identity(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


class GlobalOptimizationLevenbergMarquardt:
    """`OptimizePoseGraph` by Levenberg-Marquardt; the line-process values end up in `edge.confidence`."""

    def _zeta(self, nodes, edges):
        Z = np.empty((len(edges), 6))
        for k, e in enumerate(edges):
            Z[k] = _lin6(_inv(e.transformation) @ _inv(nodes[e.target_node_id]) @ nodes[e.source_node_id])
        return Z

    @staticmethod
    def _line_process(Z, edges, weight):
        lp = np.ones(len(edges))
        for k, e in enumerate(edges):
            if e.uncertain:
                r2 = float(Z[k] @ e.information @ Z[k])
                t = weight / (weight + r2)
                lp[k] = t * t
        return lp

    @staticmethod
    def _residual(Z, lp, edges, weight):
        r = 0.0
        for k, e in enumerate(edges):
            r += lp[k] * float(Z[k] @ e.information @ Z[k])
            if e.uncertain:
                r += weight * (np.sqrt(lp[k]) - 1.0) ** 2
        return r

    def _linear_system(self, nodes, edges, Z, lp):
        import scipy.sparse as sp
        n = len(nodes)
        rows, cols, vals = [], [], []
        b = np.zeros(6 * n)
        for k, e in enumerate(edges):
            s, t = e.source_node_id, e.target_node_id
            A = _inv(e.transformation) @ _inv(nodes[t])
            Js = np.stack([_lin6(A @ _GEN[i] @ nodes[s]) for i in range(6)], axis=1)
            Jt = -Js
            L = lp[k] * e.information
            blocks = ((s, s, Js.T @ L @ Js), (s, t, Js.T @ L @ Jt), (t, s, Jt.T @ L @ Js), (t, t, Jt.T @ L @ Jt))
            for bi, bj, M in blocks:
                r, c = np.meshgrid(np.arange(6) + 6 * bi, np.arange(6) + 6 * bj, indexing="ij")
                rows.append(r.ravel()); cols.append(c.ravel()); vals.append(M.ravel())
            b[6 * s:6 * s + 6] -= Js.T @ L @ Z[k]
            b[6 * t:6 * t + 6] -= Jt.T @ L @ Z[k]
        H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * n, 6 * n)).tocsr()
        return H, b

    @staticmethod
    def _fix_reference(H, b, ref):
        import scipy.sparse as sp
        if ref < 0:
            return H, b
        H = H.tolil()
        idx = np.arange(6 * ref, 6 * ref + 6)
        H[idx, :] = 0.0
        H[:, idx] = 0.0
        for i in idx:
            H[i, i] = 1.0
        b = b.copy(); b[idx] = 0.0
        return H.tocsr(), b

    def OptimizePoseGraph(self, pose_graph: PoseGraph, criteria: GlobalOptimizationConvergenceCriteria, option: GlobalOptimizationOption):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        edges = pose_graph.edges
        nodes = [nd.pose.copy() for nd in pose_graph.nodes]
        n = len(nodes)
        if n == 0 or not edges:
            return
        avg_corr = float(np.mean([e.information[5, 5] for e in edges]))
        weight = option.preference_loop_closure * option.max_correspondence_distance ** 2 * avg_corr     # ComputeLineProcessWeight [O3D ?]
        Z = self._zeta(nodes, edges)
        lp = np.ones(len(edges))
        cur = self._residual(Z, lp, edges, weight)
        H, b = self._fix_reference(*self._linear_system(nodes, edges, Z, lp), option.reference_node)
        lam = 1e-5 * H.diagonal().max()                                                               # tau = 1e-5 [O3D ?]
        ni, stop = 2.0, False
        eye = sp.identity(6 * n, format="csr")
        for _ in range(criteria.max_iteration):
            lm_count, rho = 0, 0.0
            while True:
                delta = spl.spsolve((H + lam * eye).tocsc(), b)
                x_norm = np.sqrt(sum(np.sum(T[:3, :] ** 2) for T in nodes))
                stop = stop or np.linalg.norm(delta) < criteria.min_relative_increment * (x_norm + criteria.min_relative_increment)
                if not stop:
                    new_nodes = [_vec6_to_T(delta[6 * i:6 * i + 6]) @ nodes[i] for i in range(n)]
                    Zn = self._zeta(new_nodes, edges)
                    lpn = self._line_process(Zn, edges, weight)
                    new = self._residual(Zn, lpn, edges, weight)
                    rho = (cur - new) / (float(delta @ (lam * delta + b)) + 1e-3)
                    if rho > 0:
                        stop = stop or (cur - new) < criteria.min_relative_residual_increment * cur
                        alpha = min(1.0 - (2.0 * rho - 1.0) ** 3, criteria.upper_scale_factor)
                        lam *= max(criteria.lower_scale_factor, alpha); ni = 2.0
                        cur, Z, lp, nodes = new, Zn, lpn, new_nodes
                        H, b = self._fix_reference(*self._linear_system(nodes, edges, Z, lp), option.reference_node)
                        stop = stop or np.max(np.abs(b)) < criteria.min_right_term or cur < criteria.min_residual
                    else:
                        lam *= ni; ni *= 2.0
                lm_count += 1
                stop = stop or lm_count > criteria.max_iteration_lm
                if rho > 0 or stop:
                    break
            if stop:
                break
        for nd, T in zip(pose_graph.nodes, nodes):
            nd.pose = T
        for e, l in zip(edges, lp):
            e.confidence = float(l)


def global_optimization(pose_graph: PoseGraph, method=None, criteria=None, option=None) -> None:
    """`o3d.pipelines.registration.global_optimization` (S3:351-354): optimise, drop the uncertain edges whose line process
    ended below `option.edge_prune_threshold`, optimise again; the graph is modified in place."""
    method = method or GlobalOptimizationLevenbergMarquardt()
    criteria = criteria or GlobalOptimizationConvergenceCriteria()
    option = option or GlobalOptimizationOption()
    n = len(pose_graph.nodes)
    for e in pose_graph.edges:
        if not (0 <= e.source_node_id < n and 0 <= e.target_node_id < n):
            raise RuntimeError("global_optimization: invalid pose graph (edge refers to a missing node)")
    work = copy.deepcopy(pose_graph)
    method.OptimizePoseGraph(work, criteria, option)
    work.edges = [e for e in work.edges if (not e.uncertain) or e.confidence > option.edge_prune_threshold]
    method.OptimizePoseGraph(work, criteria, option)
    pose_graph.nodes = work.nodes
    pose_graph.edges = work.edges


# ------------------------------------------------------------------------------------------- the reference's builders
def build_circuit_pose_graph(clouds, absolute_poses, relative_poses, voxel_size=0.1, information_fn=None) -> PoseGraph:
    """Script 3, steps 1-4 (`S3:292-340`): node i = absolute pose i (the identity is appended first and the absolute
    poses after it, exactly as the script does), edge i -> i+1 carries the INVERTED relative pose and the information
    matrix of (cloud i, cloud i+1); the last edge (n-1 -> 0) closes the loop and is the only `uncertain` one."""
    from .refinement import Transformar_de_volta
    if information_fn is None:
        from .registration import get_information_matrix_from_point_clouds as information_fn
    n = len(clouds)
    g = PoseGraph()
    g.nodes.append(PoseGraphNode(np.identity(4)))
    for i in range(n):
        inv_rel = Transformar_de_volta(relative_poses[i])
        if i < n - 1:
            g.nodes.append(PoseGraphNode(absolute_poses[i]))
            info = information_fn(clouds[i], clouds[i + 1], voxel_size, inv_rel)
            g.edges.append(PoseGraphEdge(i, i + 1, inv_rel, info, uncertain=False))
        else:
            info = information_fn(clouds[i], clouds[0], voxel_size, inv_rel)
            g.edges.append(PoseGraphEdge(i, 0, inv_rel, info, uncertain=True))
    return g


def full_registration(lista_nuvens, voxel_size, k, verbose=False) -> PoseGraph:
    """`ALL_FUNCTIONS.py:342-394`: every cloud registered onto its next k clouds with `Coarse_to_fine_FGR_M_GICP`;
    consecutive pairs are odometry edges (and accumulate the node poses), the others are uncertain loop closures."""
    from .functions import Coarse_to_fine_FGR_M_GICP
    g = PoseGraph()
    odometry = np.identity(4)
    g.nodes.append(PoseGraphNode(odometry))
    n = len(lista_nuvens)
    ok = 0
    for s in range(n):
        for t in range(s + 1, n):
            if t - s > k and t != s + 1:
                continue
            res, info = Coarse_to_fine_FGR_M_GICP(lista_nuvens[s], lista_nuvens[t], voxel_size)
            if t == s + 1:
                odometry = res.transformation @ odometry
                g.nodes.append(PoseGraphNode(np.linalg.inv(odometry)))
            g.edges.append(PoseGraphEdge(s, t, res.transformation, info, uncertain=(t != s + 1)))
            ok += 1 if res.fitness > 0.40 else 0
            if verbose:
                print(f"Registering cloud {s} in cloud {t}: {'Sucesso' if res.fitness > 0.40 else 'Falhou'}")
    # the reference only PRINTS its success count (fitness > 0.40, `AF:369,385,393`); kept on the graph object for callers and tests
    g.successes = ok
    g.attempted = len(g.edges)
    return g
