"""Known-answer tests of each oracle stage against brute-force numpy."""
import numpy as np
import pytest

from conftest import pose_error


def _rng(seed=0):
    return np.random.default_rng(seed)


def _surface_cloud(n, seed=0):
    r = _rng(seed)
    a = r.uniform(-10, 10, (n, 3))
    a[: n // 2, 2] = 0.05 * np.sin(a[: n // 2, 0])          # ground
    a[n // 2: 3 * n // 4, 0] = 4.0 + 0.01 * r.standard_normal(n // 4)   # wall
    return a


def test_knn_matches_bruteforce(oracle):
    r = _rng(1)
    for dim in (3, 33):
        pts = r.standard_normal((700, dim)); q = r.standard_normal((50, dim))
        idx, d2, cnt = oracle.knn(pts, q, 7)
        D = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
        ref = np.argsort(D, axis=1, kind="stable")[:, :7]
        assert np.array_equal(idx, ref)
        assert np.allclose(d2, np.take_along_axis(D, ref, 1), rtol=1e-13, atol=0)
        assert (cnt == 7).all()


def test_knn_hybrid_radius_cap_and_small_cloud(oracle):
    r = _rng(2)
    pts = r.uniform(0, 1, (300, 3))
    idx, d2, cnt = oracle.knn(pts, pts, 20, radius=0.15)
    D = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    for i in range(300):
        order = np.argsort(D[i], kind="stable")[:20]
        order = order[D[i, order] < 0.15 ** 2]
        assert cnt[i] == len(order)
        assert np.array_equal(idx[i, : cnt[i]], order)
        assert (idx[i, cnt[i]:] == -1).all()
    idx, d2, cnt = oracle.knn(pts[:5], pts[:5], 30)      # fewer points than k
    assert (cnt == 5).all()


def test_voxel_down_sample_matches_numpy(oracle):
    pts = _surface_cloud(5000, 3).astype(np.float32).astype(np.float64)
    v = 0.4
    out = oracle.voxel_down_sample(pts, v)
    org = pts.min(0) - v / 2
    key = np.floor((pts - org) / v).astype(np.int64)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    ref = np.zeros((len(uniq), 3))
    np.add.at(ref, inv.ravel(), pts)
    ref /= np.bincount(inv.ravel())[:, None]
    assert out.shape == ref.shape
    assert np.allclose(out, ref, rtol=0, atol=1e-12)        # np.unique order == (ix,iy,iz) lexicographic
    with pytest.raises(RuntimeError):
        oracle.voxel_down_sample(pts, 0.0)
    assert oracle.voxel_down_sample(np.zeros((0, 3)), 0.1).shape == (0, 3)


def test_sor_matches_numpy(oracle):
    pts = _surface_cloud(3000, 4)
    keep, avg, mu, sd = oracle.remove_statistical_outlier(pts, 30, 1.0)
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    avg_ref = np.sort(D, axis=1)[:, :30].mean(1)             # self (0) included
    assert np.allclose(avg, avg_ref, rtol=1e-12)
    mu_ref = avg_ref.mean(); sd_ref = np.sqrt(((avg_ref - mu_ref) ** 2).sum() / (len(pts) - 1))
    assert np.isclose(mu, mu_ref) and np.isclose(sd, sd_ref)
    assert np.array_equal(keep, (avg_ref > 0) & (avg_ref < mu_ref + sd_ref))
    with pytest.raises(RuntimeError):
        oracle.remove_statistical_outlier(pts, 0, 1.0)
    with pytest.raises(RuntimeError):
        oracle.remove_statistical_outlier(pts, 30, 0.0)


def test_covariance_and_normals_match_numpy(oracle):
    pts = _surface_cloud(2000, 5)
    cov = oracle.estimate_covariances(pts, oracle.SEARCH_KNN, 20)
    nrm = oracle.estimate_normals(pts, oracle.SEARCH_KNN, 20)
    D = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    nn = np.argsort(D, axis=1, kind="stable")[:, :20]
    for i in range(0, 2000, 37):
        P = pts[nn[i]]
        C = (P.T @ P) / 20 - np.outer(P.mean(0), P.mean(0))
        assert np.allclose(cov[i], C, atol=1e-10)
        w, V = np.linalg.eigh(C)
        if w[1] - w[0] > 1e-6 * w[2]:
            assert abs(abs(V[:, 0] @ nrm[i]) - 1) < 1e-8
        assert abs(np.linalg.norm(nrm[i]) - 1) < 1e-12
    # hybrid mode with too few neighbours -> identity covariance -> (0,0,1)
    far = np.array([[0, 0, 0], [100, 0, 0], [0, 100, 0], [0, 0, 100.0]])
    n2 = oracle.estimate_normals(far, oracle.SEARCH_HYBRID, 20, 0.2)
    assert np.array_equal(n2, np.tile([0, 0, 1.0], (4, 1)))
    # prior orientation flips the sign
    n3 = oracle.estimate_normals(pts, oracle.SEARCH_KNN, 20, prior=-nrm)
    assert np.allclose(n3, -nrm)


def test_fast_eigen_special_cases(oracle):
    assert np.array_equal(oracle.fast_eigen3x3(np.zeros((3, 3))), [0, 0, 0])
    assert np.array_equal(oracle.fast_eigen3x3(np.diag([3.0, 1.0, 2.0])), [0, 1, 0])
    assert np.array_equal(oracle.fast_eigen3x3(np.diag([1.0, 3.0, 2.0])), [1, 0, 0])
    assert np.array_equal(oracle.fast_eigen3x3(np.eye(3)), [0, 0, 1])
    r = _rng(6)
    for _ in range(200):
        A = r.standard_normal((3, 3)); C = A @ A.T
        w, V = np.linalg.eigh(C)
        n = oracle.fast_eigen3x3(C)
        assert abs(abs(V[:, 0] @ n) - 1) < 1e-6 * max(1.0, w[2] / max(w[1] - w[0], 1e-12))


def test_covariances_from_normals_closed_form(oracle):
    r = _rng(7)
    n = r.standard_normal((500, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[0] = [-1, 0, 0]; n[1] = [-0.995, np.sqrt(1 - 0.995 ** 2), 0]; n[2] = [1, 0, 0]
    C = oracle.covariances_from_normals(n, 1e-3)
    for i in range(500):
        m = np.array([1.0, 0, 0]) if n[i, 0] < -0.99 else n[i]
        assert np.allclose(C[i], np.eye(3) - (1 - 1e-3) * np.outer(m, m), atol=1e-12)


def _numpy_linearize(P, Cs, Q, Ct, corr, loss):
    JTJ = np.zeros((6, 6)); JTr = np.zeros(6); r2 = 0.0
    for s, t in corr:
        M = Cs[s] + Ct[t]
        w, V = np.linalg.eigh(M)
        W = (V / np.sqrt(w)) @ V.T
        vs = P[s]
        S = -np.array([[0, -vs[2], vs[1]], [vs[2], 0, -vs[0]], [-vs[1], vs[0], 0]])
        J = W @ np.hstack([S, np.eye(3)])
        r = W @ (P[s] - Q[t])
        for k in range(3):
            wt = 1.0 / abs(r[k]) if loss == "l1" else (1.0 if loss == "l2" else 1.0 / (1.0 + r[k] ** 2) ** 2)
            JTJ += wt * np.outer(J[k], J[k]); JTr += wt * J[k] * r[k]; r2 += r[k] ** 2
    return JTJ, JTr, r2


def test_gicp_linearize_matches_numpy(oracle):
    r = _rng(8)
    P = r.uniform(-5, 5, (300, 3)); Q = P + 0.05 * r.standard_normal((300, 3))
    ns = r.standard_normal((300, 3)); ns /= np.linalg.norm(ns, axis=1, keepdims=True)
    nt = r.standard_normal((300, 3)); nt /= np.linalg.norm(nt, axis=1, keepdims=True)
    Cs = oracle.covariances_from_normals(ns); Ct = oracle.covariances_from_normals(nt)
    corr = np.stack([np.arange(300), r.permutation(300)], 1).astype(np.int32)[:250]
    for loss, code in (("l1", oracle.LOSS_L1), ("l2", oracle.LOSS_L2), ("gm", oracle.LOSS_GM)):
        A, b, r2 = oracle.gicp_linearize(P, Cs, Q, Ct, corr, code, 1.0)
        A0, b0, r20 = _numpy_linearize(P, Cs, Q, Ct, corr, loss)
        assert np.allclose(A, A0, rtol=1e-9, atol=1e-9 * abs(A0).max())
        assert np.allclose(b, b0, rtol=1e-9, atol=1e-9 * abs(b0).max())
        assert np.isclose(r2, r20, rtol=1e-10)


def test_solve_update_is_rz_ry_rx(oracle):
    r = _rng(9)
    A = r.standard_normal((6, 6)); A = A @ A.T + np.eye(6)
    b = r.standard_normal(6) * 0.1
    T, rc = oracle.solve_update(A, b)
    x = np.linalg.solve(A, -b)
    def Rx(a): c, s = np.cos(a), np.sin(a); return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    def Ry(a): c, s = np.cos(a), np.sin(a); return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    def Rz(a): c, s = np.cos(a), np.sin(a); return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    assert rc == 0
    assert np.allclose(T[:3, :3], Rz(x[2]) @ Ry(x[1]) @ Rx(x[0]), atol=1e-12)
    assert np.allclose(T[:3, 3], x[3:], atol=1e-12)
    T2, rc2 = oracle.solve_update(np.zeros((6, 6)), b)        # singular -> identity, as Open3D
    assert rc2 != 0 and np.array_equal(T2, np.eye(4))


def test_gicp_recovers_planted_motion_on_box_scene(oracle):
    """Noise-free planar box: identical source/target point sets, planted SE(3)."""
    r = _rng(10)
    faces = []
    for axis in range(3):
        for side in (-1.0, 1.0):
            p = r.uniform(-1, 1, (600, 3)) * np.array([3.0, 2.0, 1.5]); p[:, axis] = side * [3.0, 2.0, 1.5][axis]
            faces.append(p)
    tgt = np.vstack(faces)
    ang = 0.03
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    T_true = np.eye(4); T_true[:3, :3] = R; T_true[:3, 3] = [0.05, -0.03, 0.02]
    Ti = np.linalg.inv(T_true)
    src = tgt @ Ti[:3, :3].T + Ti[:3, 3]
    nt = oracle.estimate_normals(tgt, oracle.SEARCH_KNN, 20); nsrc = oracle.estimate_normals(src, oracle.SEARCH_KNN, 20)
    res = oracle.registration_gicp(src, tgt, 0.5, np.eye(4), src_normals=nsrc, tgt_normals=nt, loss=oracle.LOSS_L2, max_it=60)
    ang_e, dt = pose_error(res.transformation, T_true)
    assert ang_e < 1e-6 and dt < 1e-6, (ang_e, dt)
    assert res.fitness == 1.0 and res.inlier_rmse < 1e-6


def test_gicp_error_and_degenerate_conventions(oracle):
    p = _surface_cloud(200, 11); n = oracle.estimate_normals(p, oracle.SEARCH_KNN, 10)
    with pytest.raises(RuntimeError):
        oracle.registration_gicp(p, p, 0.0, np.eye(4), src_normals=n, tgt_normals=n)
    far = p + 1000.0
    res = oracle.registration_gicp(p, far, 0.5, np.eye(4), src_normals=n, tgt_normals=n, max_it=5)
    assert res.fitness == 0 and res.inlier_rmse == 0 and np.array_equal(res.transformation, np.eye(4))
    assert res.converged and res.iterations == 1


def test_information_matrix_and_evaluate(oracle):
    p = _surface_cloud(500, 12)
    ev = oracle.evaluate_registration(p, p, 0.1, np.eye(4))
    assert ev.fitness == 1.0 and ev.inlier_rmse == 0.0 and ev.n_corr == 500
    info = oracle.information_matrix(p, p, 0.1, np.eye(4))
    G = np.zeros((6, 6))
    for x, y, z in p:
        for row in ([0, z, -y, 1, 0, 0], [-z, 0, x, 0, 1, 0], [y, -x, 0, 0, 0, 1]):
            G += np.outer(row, row)
    assert np.allclose(info, G, rtol=1e-12)
