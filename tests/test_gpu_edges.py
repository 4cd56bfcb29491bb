"""Edge cases of the preprocessing + GICP path against the oracle: duplicated points (exact distance ties), clouds smaller
than the neighbour counts, everything collapsing into one voxel, coordinates far from the origin (float32 grid), degenerate
geometry (all points on a line / in a plane), and calls that must fail loudly instead of returning numbers."""
import numpy as np
import pytest

from conftest import pkg, pose_error

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return pkg()


def _sorted_rows(a):
    a = np.asarray(a, dtype=np.float64)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def test_duplicated_points_ties(P, oracle, small_pair):
    """Every point twice: zero distances and exact ties at the k-th neighbour.  Voxel means, SOR mean distances (hence the
    mask) and stage counts must still be the oracle's."""
    base = small_pair["source"][:6000]
    pts = np.concatenate([base, base])[np.random.default_rng(3).permutation(2 * len(base))]
    v = P.PointCloud(pts).voxel_down_sample(0.2)
    rv = oracle.voxel_down_sample(pts, 0.2)
    assert np.array_equal(_sorted_rows(v.points), _sorted_rows(rv.astype(np.float32)))
    # SOR straight on the duplicated cloud (no voxel stage): ties everywhere
    pc = P.PointCloud(pts)
    clean, index = pc.remove_statistical_outlier(30, 1.0)
    keep, avg, mu, sd = oracle.remove_statistical_outlier(pts, 30, 1.0)
    assert np.array_equal(np.asarray(index), np.nonzero(keep)[0])


@pytest.mark.parametrize("n", [1, 2, 3, 5, 19, 20, 21, 29, 30, 31])
def test_clouds_smaller_than_the_neighbour_counts(P, oracle, small_pair, n):
    pts = small_pair["source"][100:100 + n]
    pc = P.PointCloud(pts)
    clean, index = pc.remove_statistical_outlier(30, 1.0)
    keep, *_ = oracle.remove_statistical_outlier(pts, 30, 1.0)
    assert np.array_equal(np.asarray(index), np.nonzero(keep)[0]), n
    pc.estimate_normals(P.KDTreeSearchParamKNN(knn=20))
    ref = oracle.estimate_normals(pts, oracle.SEARCH_KNN, 20)
    got = pc.normals
    assert got.shape == ref.shape
    # fewer than 3 neighbours: Open3D leaves (0,0,1); otherwise the same analytic solver (sign free: no prior)
    dots = np.abs((got * ref).sum(1))
    assert (dots > 1 - 1e-4).all(), (n, dots.min())


def test_everything_in_one_voxel_and_single_point_clouds(P, oracle, small_pair):
    pts = small_pair["source"][:500]
    one = P.PointCloud(pts).voxel_down_sample(1000.0)
    rv = oracle.voxel_down_sample(pts, 1000.0)
    assert len(one) == 1 == len(rv) and np.array_equal(one.points, rv.astype(np.float32).astype(np.float64))
    # a multiscale call whose clouds collapse to one point per cloud: no correspondences can be linearised into a solvable
    # system; the call must come back (no hang, no fault) with the initial pose or a clean error
    est = P.registration.TransformationEstimationForGeneralizedICP(P.registration.L1Loss())
    crit = P.registration.ICPConvergenceCriteria(1e-6, 1e-6, 10)
    T0 = small_pair["T_fgr"]
    try:
        r = P.registration.multiscale_gicp(P.PointCloud(pts), P.PointCloud(small_pair["target"][:500]), [1000.0], [3000.0], T0, est, crit)
        assert np.isfinite(r.transformation).all()
        ref = oracle.multiscale_gicp(pts, small_pair["target"][:500], [1000.0], [3000.0], T0)
        ang, dt = pose_error(r.transformation, ref.transformation)
        assert ang < 1e-6 and dt < 1e-6
    except RuntimeError as e:
        assert "pose" in str(e) or "numeric" in str(e).lower() or "singular" in str(e).lower(), str(e)


def test_far_from_the_origin(P, oracle, small_pair):
    """UTM-like coordinates: 300 km / 4000 km offsets leave float32 a 3 cm / 25 cm grid; the voxel keys are computed in
    float64 from the float32 values exactly like the oracle, so counts and means agree bit for bit."""
    off = np.array([3.0e5, 4.0e6, 100.0])
    pts = (small_pair["source"][:8000].astype(np.float64) + off).astype(np.float32)
    for voxel in (0.5, 2.0):
        v = P.PointCloud(pts).voxel_down_sample(voxel)
        rv = oracle.voxel_down_sample(pts, voxel)
        assert len(v) == len(rv)
        assert np.array_equal(_sorted_rows(v.points), _sorted_rows(rv.astype(np.float32)))
    pc = P.PointCloud(pts).voxel_down_sample(2.0)
    clean, index = pc.remove_statistical_outlier(30, 1.0)
    keep, *_ = oracle.remove_statistical_outlier(pc.points, 30, 1.0)
    assert np.array_equal(np.asarray(index), np.nonzero(keep)[0])


def test_degenerate_geometry_line_and_plane(P, oracle):
    rng = np.random.default_rng(5)
    t = rng.uniform(-20, 20, 4000)
    line = np.stack([t, 0.5 * t + 1.0, np.full_like(t, 2.0)], 1).astype(np.float32)
    plane = np.stack([rng.uniform(-20, 20, 4000), rng.uniform(-20, 20, 4000), np.full(4000, -1.5)], 1).astype(np.float32)
    for name, pts in (("line", line), ("plane", plane)):
        pc = P.PointCloud(pts).voxel_down_sample(0.3)
        rv = oracle.voxel_down_sample(pts, 0.3)
        assert len(pc) == len(rv), name
        clean, index = pc.remove_statistical_outlier(30, 1.0)
        keep, *_ = oracle.remove_statistical_outlier(pc.points, 30, 1.0)
        assert np.array_equal(np.asarray(index), np.nonzero(keep)[0]), name
        clean.estimate_normals(P.KDTreeSearchParamKNN(knn=20))
        assert np.isfinite(clean.normals).all(), name
        if name == "plane":
            assert (np.abs(clean.normals[:, 2]) > 1 - 1e-6).all()


def test_calls_that_must_fail_loudly(P, small_pair):
    src = P.PointCloud(small_pair["source"][:2000]); tgt = P.PointCloud(small_pair["target"][:2000])
    est = P.registration.TransformationEstimationForGeneralizedICP(P.registration.L1Loss())
    crit = P.registration.ICPConvergenceCriteria(1e-6, 1e-6, 10)
    with pytest.raises((RuntimeError, ValueError)):
        P.registration.multiscale_gicp(src, tgt, [0.4, 0.2], [1.2], np.eye(4), est, crit)          # ragged scale tables
    with pytest.raises((RuntimeError, ValueError)):
        P.registration.multiscale_gicp(src, tgt, [], [], np.eye(4), est, crit)                      # no scales
    with pytest.raises((RuntimeError, ValueError)):
        P.registration.multiscale_gicp(src, tgt, [0.4], [-1.0], np.eye(4), est, crit)               # negative radius
    bad = np.eye(4); bad[0, 0] = np.nan
    with pytest.raises((RuntimeError, ValueError)):
        P.registration.multiscale_gicp(src, tgt, [0.4], [1.2], bad, est, crit)                      # NaN initial pose
    with pytest.raises((RuntimeError, ValueError, TypeError)):
        P.PointCloud(np.zeros((10, 2)))                                                             # not N x 3
    # an empty cloud in a batch reports an error for that pair only
    res = P.registration.register_pairs([(src, tgt, np.eye(4))], [0.4], [1.2], est, crit)
    assert np.isfinite(res[0].transformation).all()


def test_calls_are_ordered_after_torch_work_on_the_default_stream(P, oracle, small_pair):
    """torch's default stream is the legacy NULL stream while the library works on a stream of its own: every call must wait
    for the torch kernels that produce its inputs (deepcopy = clone, transform = matmul, float conversions), and torch work
    enqueued after an asynchronous call (estimate_normals returns without a host sync) must see its outputs.  A missing fence
    shows up as a voxel grid / normals of half-written buffers."""
    import copy
    import torch
    rng = np.random.default_rng(11)
    base = np.concatenate([small_pair["source"] + rng.normal(0, 3.0, 3).astype(np.float32) for _ in range(40)])   # ~600k points
    T = np.eye(4); T[:3, :3] = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]); T[:3, 3] = [5.0, -3.0, 1.0]
    pc = P.PointCloud(base)
    torch.cuda.synchronize()
    for rep in range(4):
        q = copy.deepcopy(pc)
        q.transform(T)                                 # asynchronous torch kernels on the default stream ...
        v = q.voxel_down_sample(0.7)                   # ... and a library call right behind them
        ref = oracle.voxel_down_sample(q.points.astype(np.float32), 0.7)
        assert len(v) == len(ref)
        assert np.array_equal(_sorted_rows(v.points), _sorted_rows(ref.astype(np.float32))), rep
    # asynchronous call followed by torch work on its output
    small = P.PointCloud(base[:50000])
    small.estimate_normals(P.KDTreeSearchParamKNN(knn=20))
    doubled = (small._nrm * 2.0).cpu().numpy()         # default-stream kernel reading the normals just enqueued
    torch.cuda.synchronize()
    assert np.array_equal(doubled, small.normals.astype(np.float32) * 2.0)
    assert np.allclose(np.linalg.norm(small.normals, axis=1), 1.0, atol=1e-5)


def test_register_pairs_waits_for_the_producers_of_its_inputs(P, small_pair):
    """pcr_register_pairs: the workers wait for the stream the inputs were produced on, also when that is the default (NULL)
    stream.  Inputs made by torch kernels right before the call give the same poses as inputs that were synchronised first."""
    import copy
    import torch
    est = P.registration.TransformationEstimationForGeneralizedICP(P.registration.L2Loss())
    crit = P.registration.ICPConvergenceCriteria(1e-6, 1e-6, 20)
    vox, dst = [0.4, 0.2], [1.2, 0.4]
    T = np.eye(4); T[:3, 3] = [0.3, -0.2, 0.05]
    src0, tgt0 = P.PointCloud(small_pair["source"]), P.PointCloud(small_pair["target"])

    def inputs():
        out = []
        for k in range(6):
            s = copy.deepcopy(src0); s.transform(T)
            out.append((s, copy.deepcopy(tgt0), small_pair["T_fgr"] @ np.linalg.inv(T)))
        return out
    a = inputs()
    torch.cuda.synchronize()
    ref = P.registration.register_pairs(a, vox, dst, est, crit, inflight=3)
    got = P.registration.register_pairs(inputs(), vox, dst, est, crit, inflight=3)       # no synchronisation in between
    for r, g in zip(ref, got):
        assert np.array_equal(r.transformation, g.transformation)
        assert [s["n_clean"] for s in r.scales] == [s["n_clean"] for s in g.scales]
