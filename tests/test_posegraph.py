"""Pose-graph LM (unpinned restatement): property tests on synthetic graphs."""
import copy

import numpy as np
import pytest

import pcr_amd
from pcr_amd import posegraph as pg
from conftest import pose_error


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _T(R, t):
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T


def _loop(n, rng):
    """Ground-truth node poses on a closed loop (node frame -> global)."""
    poses = []
    for i in range(n):
        a = 2 * np.pi * i / n
        poses.append(_T(_rot([0, 0, 1], a) @ _rot(rng.normal(size=3), 0.05), [20 * np.cos(a) - 20, 20 * np.sin(a), 0.3 * rng.normal()]))
    poses[0] = np.eye(4)
    return poses


def _graph(gt, rng, odo_noise=(0.0, 0.0), closure=True, info_scale=5000.0):
    n = len(gt)
    g = pg.PoseGraph()
    info = np.diag([info_scale * 30] * 3 + [info_scale] * 3)          # info[5,5] ~ number of correspondences
    cur = np.eye(4)
    g.nodes.append(pg.PoseGraphNode(cur))
    for i in range(n - 1):
        X = np.linalg.inv(gt[i + 1]) @ gt[i]                         # aligns node i to node i+1
        X = _T(_rot(rng.normal(size=3), odo_noise[0]) @ X[:3, :3], X[:3, 3] + odo_noise[1] * rng.normal(size=3))
        g.edges.append(pg.PoseGraphEdge(i, i + 1, X, info, uncertain=False))
        cur = cur @ np.linalg.inv(X)                                  # T_{i+1} = T_i X^-1
        g.nodes.append(pg.PoseGraphNode(cur))
    if closure:
        g.edges.append(pg.PoseGraphEdge(n - 1, 0, np.linalg.inv(gt[0]) @ gt[n - 1], info, uncertain=True))
    return g


OPT = dict(max_correspondence_distance=0.2, edge_prune_threshold=0.25, reference_node=0)


def test_consistent_graph_is_a_fixed_point():
    rng = np.random.default_rng(1)
    gt = _loop(12, rng)
    g = _graph(gt, rng)
    pg.global_optimization(g, pg.GlobalOptimizationLevenbergMarquardt(), pg.GlobalOptimizationConvergenceCriteria(), pg.GlobalOptimizationOption(**OPT))
    for nd, T in zip(g.nodes, gt):
        a, d = pose_error(nd.pose, T)
        assert a < 1e-7 and d < 1e-8            # arccos near 1 resolves ~2e-8 rad
    assert len(g.edges) == 12 and all(e.confidence > 0.999 for e in g.edges)


def test_loop_closure_pulls_drifted_odometry_back():
    rng = np.random.default_rng(2)
    gt = _loop(40, rng)
    g = _graph(gt, rng, odo_noise=(0.004, 0.02))
    before = [pose_error(nd.pose, T) for nd, T in zip(g.nodes, gt)]
    g0 = copy.deepcopy(g)
    pg.global_optimization(g, option=pg.GlobalOptimizationOption(**OPT))
    after = [pose_error(nd.pose, T) for nd, T in zip(g.nodes, gt)]
    assert np.allclose(g.nodes[0].pose, np.eye(4))                                   # reference node held
    assert np.mean([d for _, d in after]) < 0.75 * np.mean([d for _, d in before])      # one closure: the middle of the chain stays a random walk
    assert after[-1][1] < 0.3 * before[-1][1] and after[-1][0] < 0.5 * before[-1][0]   # the end of the chain gains most
    # the closing edge is now (nearly) satisfied
    e = g.edges[-1]
    M = np.linalg.inv(e.transformation) @ np.linalg.inv(g.nodes[e.target_node_id].pose) @ g.nodes[e.source_node_id].pose
    M0 = np.linalg.inv(e.transformation) @ np.linalg.inv(g0.nodes[e.target_node_id].pose) @ g0.nodes[e.source_node_id].pose
    assert np.linalg.norm(M[:3, 3]) < 0.2 * np.linalg.norm(M0[:3, 3])
    assert e.confidence > 0.25 and len(g.edges) == 40
    # inputs are only modified through the documented fields
    assert len(g0.nodes) == len(g.nodes)


def test_false_loop_closure_is_switched_off_and_pruned():
    rng = np.random.default_rng(3)
    gt = _loop(30, rng)
    g = _graph(gt, rng, odo_noise=(0.001, 0.005))
    bad = pg.PoseGraphEdge(5, 20, _T(_rot([0, 0, 1], 1.0), [4.0, -3.0, 0.5]), g.edges[0].information, uncertain=True)   # nonsense match
    g.edges.append(bad)
    ref = copy.deepcopy(g); ref.edges.pop()
    pg.global_optimization(g, option=pg.GlobalOptimizationOption(**OPT))
    pg.global_optimization(ref, option=pg.GlobalOptimizationOption(**OPT))
    assert len(g.edges) == 30 and not any(e.source_node_id == 5 and e.target_node_id == 20 for e in g.edges)
    for a, b in zip(g.nodes, ref.nodes):
        ang, d = pose_error(a.pose, b.pose)
        assert ang < 2e-3 and d < 2e-2
    with pytest.raises(RuntimeError):
        h = pg.PoseGraph(); h.nodes.append(pg.PoseGraphNode()); h.edges.append(pg.PoseGraphEdge(0, 3))
        pg.global_optimization(h)


def test_script3_graph_builder_conventions():
    """S3:292-340 with a stub information function: identity first, absolute poses after it, inverted relative poses on
    the edges, only the closing edge uncertain; the graph of a consistent circuit is already optimal."""
    rf = pcr_amd.refinement
    rng = np.random.default_rng(4)
    ab = [np.eye(4)]
    for _ in range(6):
        ab.append(_T(_rot(rng.normal(size=3), 0.2) @ ab[-1][:3, :3], ab[-1][:3, 3] + rng.normal(size=3)))
    rel = rf.poses_absolutas_para_relativas(ab + [np.eye(4)])
    calls = []
    g = pg.build_circuit_pose_graph(list(range(7)), ab, rel, 0.1, information_fn=lambda s, t, v, T: calls.append((s, t)) or np.eye(6) * 100)
    assert len(g.nodes) == 7 and len(g.edges) == 7 and calls[-1] == (6, 0) and calls[0] == (0, 1)
    assert [e.uncertain for e in g.edges] == [False] * 6 + [True]
    np.testing.assert_allclose(g.edges[2].transformation, rf.Transformar_de_volta(rel[2]))
    np.testing.assert_allclose(g.nodes[0].pose, np.eye(4)); np.testing.assert_allclose(g.nodes[3].pose, ab[2])     # the script's off-by-one


def test_full_registration_builds_the_reference_graph(monkeypatch):
    """`full_registration` (ALL_FUNCTIONS.py:342-394) with the pairwise registration replaced by a stub: which pairs are registered
    (every cloud onto its next k), which edges are odometry (`uncertain=False`, consecutive clouds) and which loop closures, what the
    nodes hold (inverse of the accumulated odometry, `AF:357-360`), and the fitness > 0.40 success count the reference prints."""
    from types import SimpleNamespace
    from pcr_amd import functions
    rng = np.random.default_rng(3)
    n, k = 6, 2
    T = {}
    calls = []

    def fake(source, target, voxel_size):
        s, t = source, target                                    # the "clouds" are just their indices here
        calls.append((s, t, voxel_size))
        T[(s, t)] = _T(_rot(rng.normal(size=3), 0.1), rng.normal(size=3))
        fit = 0.40 if (s, t) == (0, 1) else (0.41 if t == s + 1 else 0.39)          # exactly 0.40 is NOT a success (strict >)
        return SimpleNamespace(transformation=T[(s, t)], fitness=fit), np.eye(6) * (10 * s + t)
    monkeypatch.setattr(functions, "Coarse_to_fine_FGR_M_GICP", fake)
    g = pg.full_registration(list(range(n)), 0.1, k)
    want = [(s, t) for s in range(n) for t in range(s + 1, n) if t - s <= k]
    assert [(c[0], c[1]) for c in calls] == want and all(c[2] == 0.1 for c in calls)
    assert len(g.edges) == k * (n - k) + (k * k - k) // 2 == len(want)               # the count the reference prints (AF:343)
    for e, (s, t) in zip(g.edges, want):
        assert (e.source_node_id, e.target_node_id) == (s, t) and e.uncertain == (t != s + 1)
        assert np.array_equal(e.transformation, T[(s, t)]) and e.information[0, 0] == 10 * s + t
    assert len(g.nodes) == n and np.array_equal(g.nodes[0].pose, np.eye(4))
    odo = np.eye(4)
    for i in range(n - 1):
        odo = T[(i, i + 1)] @ odo
        assert np.allclose(g.nodes[i + 1].pose, np.linalg.inv(odo), atol=1e-12)
    assert g.attempted == len(want) and g.successes == n - 2                        # consecutive pairs but (0, 1): fitness 0.41 > 0.40


@pytest.mark.gpu
def test_full_registration_on_three_facade_scans():
    """The same function on real clouds (three consecutive scans of the shipped Facade loop, k = 2): three registrations through
    `Coarse_to_fine_FGR_M_GICP` on the device, two odometry edges and one loop closure, node poses consistent with the edges, and a
    graph the optimiser accepts."""
    import os
    from conftest import GOLDEN
    f = np.load(os.path.join(GOLDEN, "facade_loop.npz"))
    clouds = [pcr_amd.PointCloud(f[f"s{i}"]) for i in range(3)]
    g = pg.full_registration(clouds, 0.1, 2)
    assert [(e.source_node_id, e.target_node_id, e.uncertain) for e in g.edges] == [(0, 1, False), (0, 2, True), (1, 2, False)]
    assert g.attempted == 3 and 0 <= g.successes <= 3
    assert np.allclose(g.nodes[1].pose, np.linalg.inv(g.edges[0].transformation), atol=1e-12)
    assert np.allclose(g.nodes[2].pose, np.linalg.inv(g.edges[2].transformation @ g.edges[0].transformation), atol=1e-12)
    # the loop-closure edge (0 -> 2) agrees with the composition of the odometry edges within the registration accuracy on these scans
    ang, dt = pose_error(g.edges[1].transformation, g.edges[2].transformation @ g.edges[0].transformation)
    assert ang < 3e-2 and dt < 0.3, (ang, dt)
    for e in g.edges:
        assert e.information.shape == (6, 6) and e.information[5, 5] > 100           # number of correspondences within voxel_size
    pg.global_optimization(g, pg.GlobalOptimizationLevenbergMarquardt(), pg.GlobalOptimizationConvergenceCriteria(),
                           pg.GlobalOptimizationOption(max_correspondence_distance=0.2, edge_prune_threshold=0.25, reference_node=0))
    assert np.allclose(g.nodes[0].pose, np.eye(4), atol=1e-9)
