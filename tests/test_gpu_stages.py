"""GPU parity of every per-cloud stage against the CPU oracle on identical inputs (calls go through the C ABI).

Tolerances: index / mask / count work is bit-exact; float32 outputs equal the float32 rounding of the
oracle's float64 value up to the stated ulp counts."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return pkg()


def _lexsort_rows(a):
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


@pytest.mark.parametrize("voxel", [0.1, 0.3, 0.5])
def test_voxel_down_sample_matches_oracle(P, oracle, small_pair, voxel):
    src = small_pair["source"]
    dev = P.PointCloud(src).voxel_down_sample(voxel).points            # float32 values as float64
    ref = oracle.voxel_down_sample(src, voxel)
    assert dev.shape == ref.shape
    a, b = _lexsort_rows(dev), _lexsort_rows(ref.astype(np.float32).astype(np.float64))
    # same stable member order -> identical float64 sums -> identical float32 roundings
    assert np.array_equal(a, b)


def test_voxel_with_normals_and_errors(P, oracle, small_pair):
    src = small_pair["source"][:5000]
    nrm = np.random.default_rng(0).standard_normal(src.shape).astype(np.float32)
    pc = P.PointCloud(src); pc.normals = nrm
    out = pc.voxel_down_sample(0.4)
    rp, rn = oracle.voxel_down_sample(src, 0.4, normals=nrm)
    key = np.lexsort((out.points[:, 2], out.points[:, 1], out.points[:, 0]))
    rp = rp.astype(np.float32).astype(np.float64)        # order by the same float32-rounded coordinates
    rkey = np.lexsort((rp[:, 2], rp[:, 1], rp[:, 0]))
    assert np.array_equal(out.normals[key], rn[rkey].astype(np.float32).astype(np.float64))
    with pytest.raises(RuntimeError):
        P.PointCloud(src).voxel_down_sample(0.0)
    with pytest.raises(RuntimeError):
        P.PointCloud(src).voxel_down_sample(1e-9)             # "voxel_size is too small"
    assert len(P.PointCloud(np.zeros((0, 3))).voxel_down_sample(0.1)) == 0


def test_bounds(P, small_pair):
    src = small_pair["source"]
    pc = P.PointCloud(src)
    assert np.array_equal(pc.get_min_bound(), src.min(0).astype(np.float64))
    assert np.array_equal(pc.get_max_bound(), src.max(0).astype(np.float64))


@pytest.fixture
def knn_form(P):
    """Selects the exact k-NN kernel for the calls of one test (pcr_set_option "knn_wave": 0 octet, 1 one query per lane) and puts
    the default rule (-1: by size and call form) and budget back afterwards."""
    yield lambda form: P._lib.set_option("knn_wave", form)
    P._lib.set_option("knn_wave", -1)
    P._lib.set_option("knnw_budget", 80)


@pytest.mark.parametrize("wave", [0, 1])
@pytest.mark.parametrize("k", [1, 20, 30, 33, 64])
def test_knn_index_is_exact(P, oracle, small_pair, knn_form, k, wave):
    """Both exact k-NN kernels -- the octet kernel and the one-query-per-lane kernel the lockstep groups run (52 % of the measured
    path) -- directly against the oracle's kd-tree."""
    import ctypes as C
    import torch
    knn_form(wave)
    pts = P.PointCloud(small_pair["source"]).voxel_down_sample(0.2).points.astype(np.float32)
    n = len(pts)
    ctx = P._lib.Context.current()
    d = torch.from_numpy(pts).cuda()
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((n, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.pcr_debug_knn(ctx.handle, C.c_void_p(d.data_ptr()), C.c_int64(n), C.c_int(k), C.c_double(0.0),
                                    C.c_void_p(idx.data_ptr()), C.c_void_p(d2.data_ptr()), C.c_void_p(cnt.data_ptr())), "debug_knn")
    ridx, rd2, rcnt = oracle.knn(pts, pts, k)
    idx = idx.cpu().numpy(); d2 = d2.cpu().numpy()
    order = np.lexsort((idx, d2), axis=1)                     # device rows are an unordered k-best set
    idx = np.take_along_axis(idx, order, 1); d2 = np.take_along_axis(d2, order, 1)
    assert (cnt.cpu().numpy() == k).all()
    assert np.allclose(d2, rd2, rtol=2e-6, atol=1e-12)
    same = idx == ridx
    # any index disagreement must be an exact distance tie at float32 resolution
    bad = np.argwhere(~same)
    for i, j in bad:
        assert abs(rd2[i, j] - ((pts[i] - pts[idx[i, j]]).astype(np.float64) ** 2).sum()) <= 4e-6 * max(rd2[i, j], 1e-12)
    assert same.mean() > 0.9999


def _debug_knn(P, pts, k, radius=0.0):
    import ctypes as C
    import torch
    n = len(pts)
    ctx = P._lib.Context.current()
    d = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)).cuda()
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((n, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.pcr_debug_knn(ctx.handle, C.c_void_p(d.data_ptr()), C.c_int64(n), C.c_int(k), C.c_double(radius),
                                    C.c_void_p(idx.data_ptr()), C.c_void_p(d2.data_ptr()), C.c_void_p(cnt.data_ptr())), "debug_knn")
    return idx.cpu().numpy(), d2.cpu().numpy(), cnt.cpu().numpy()


@pytest.mark.parametrize("k,radius,n", [(30, 0.0, None), (20, 0.25, 9000), (64, 0.0, 5000), (7, 0.0, 100), (30, 0.0, 11), (20, 0.0, 65), (30, 0.0, 1)])
def test_knn_wave_equals_octet(P, small_pair, knn_form, k, radius, n):
    """The one-query-per-lane search (pcr_knn_wave.h, option knn_wave = 1) against the octet kernel (the default): two independent exact
    searches must return the same distances row for row, and the same indices wherever distances do not tie -- also on clouds
    smaller than k, smaller than a wavefront, and under a radius cap (Open3D SearchHybrid)."""
    pts = P.PointCloud(small_pair["source"]).voxel_down_sample(0.2).points.astype(np.float32) if radius == 0.0 else small_pair["source"].astype(np.float32)
    if n is not None:
        pts = pts[:n].copy()
    knn_form(1)
    wi, wd, wc = _debug_knn(P, pts, k, radius)
    knn_form(0)
    oi, od, oc = _debug_knn(P, pts, k, radius)
    assert np.array_equal(wc, oc)
    assert (wc == np.minimum(k, len(pts))).all() or radius > 0
    ow = np.lexsort((wi, wd), axis=1); oo = np.lexsort((oi, od), axis=1)
    wi, wd = np.take_along_axis(wi, ow, 1), np.take_along_axis(wd, ow, 1)
    oi, od = np.take_along_axis(oi, oo, 1), np.take_along_axis(od, oo, 1)
    assert np.array_equal(wd, od)                                  # same float32 distances, slot for slot (inf = empty)
    differ = wi != oi
    if differ.any():                                               # only inside runs of equal distances
        r, c = np.nonzero(differ)
        for i, j in zip(r, c):
            assert (wd[i] == wd[i, j]).sum() > 1
        assert differ.mean() < 1e-3
    for row_i, row_d in zip(wi[:50], wd[:50]):                     # no index twice
        v = row_i[np.isfinite(row_d)]
        assert len(set(v.tolist())) == len(v)


def test_knn_wave_hands_hard_wavefronts_to_the_octet_kernel(P, oracle, small_pair, knn_form):
    """A wavefront of the one-query-per-lane search gives up after `knnw_budget` candidate batches and leaves its 64 queries to the octet
    kernel (default 40 batches: ~1 % of the wavefronts).  With a budget of 3 most wavefronts give up, with 8 about half: the outlier
    mask must stay the oracle's bit for bit and the normals the same either way."""
    pc = P.PointCloud(small_pair["source"]).voxel_down_sample(0.2)
    pts = pc.points
    keep, avg, mu, sd = oracle.remove_statistical_outlier(pts, 30, 1.0)
    ref_n = oracle.estimate_normals(pts, oracle.SEARCH_KNN, 20)
    knn_form(1)
    for budget in (3, 8, 40):
        P._lib.set_option("knnw_budget", budget)
        clean, index = pc.remove_statistical_outlier(30, 1.0)
        assert np.array_equal(np.asarray(index), np.nonzero(keep)[0])
        pc2 = P.PointCloud(pts); pc2.estimate_normals(P.KDTreeSearchParamKNN(knn=20))
        assert ((pc2.normals * ref_n).sum(1) > 1 - 1e-5).mean() > 0.999


def test_knn_hybrid_radius(P, oracle, small_pair):
    import ctypes as C
    import torch
    pts = small_pair["source"][:8000].copy()
    n, k, r = len(pts), 20, 0.2
    ctx = P._lib.Context.current()
    d = torch.from_numpy(pts).cuda()
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((n, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.pcr_debug_knn(ctx.handle, C.c_void_p(d.data_ptr()), C.c_int64(n), C.c_int(k), C.c_double(r),
                                    C.c_void_p(idx.data_ptr()), C.c_void_p(d2.data_ptr()), C.c_void_p(cnt.data_ptr())), "debug_knn")
    ridx, rd2, rcnt = oracle.knn(pts, pts, k, radius=r)
    cnt = cnt.cpu().numpy()
    d2 = torch.sort(d2, dim=1).values
    # counts may differ only where a neighbour sits within float32 rounding of the radius
    assert (cnt != rcnt).mean() < 1e-3
    ok = cnt == rcnt
    assert np.allclose(np.where(np.isfinite(rd2[ok]), d2.cpu().numpy()[ok], 0), np.where(np.isfinite(rd2[ok]), rd2[ok], 0), rtol=2e-6, atol=1e-12)


def test_sor_mask_is_exact(P, oracle, small_pair):
    pc = P.PointCloud(small_pair["source"]).voxel_down_sample(0.2)
    pts = pc.points
    clean, index = pc.remove_statistical_outlier(30, 1.0)
    keep, avg, mu, sd = oracle.remove_statistical_outlier(pts, 30, 1.0)
    assert np.array_equal(np.asarray(index), np.nonzero(keep)[0])
    assert np.array_equal(clean.points, pts[keep])
    with pytest.raises(RuntimeError):
        pc.remove_statistical_outlier(0, 1.0)
    with pytest.raises(RuntimeError):
        pc.remove_statistical_outlier(30, 0.0)
    tiny = P.PointCloud(pts[:7])                     # fewer points than k
    c2, i2 = tiny.remove_statistical_outlier(30, 1.0)
    k2, *_ = oracle.remove_statistical_outlier(pts[:7], 30, 1.0)
    assert np.array_equal(np.asarray(i2), np.nonzero(k2)[0])


@pytest.mark.parametrize("knn", [20, 64])
def test_normals_knn_match_oracle(P, oracle, small_pair, knn):
    """knn = 20: the reference's own setting (ALL_FUNCTIONS.py:301-302); knn = 64: BASELINE config 5."""
    pc = P.PointCloud(small_pair["source"]).voxel_down_sample(0.2)
    pts = pc.points
    pc.estimate_normals(P.KDTreeSearchParamKNN(knn=knn))
    ref = oracle.estimate_normals(pts, oracle.SEARCH_KNN, knn)
    dev = pc.normals
    dots = (dev * ref).sum(1)
    # same analytic solver on (almost) the same float64 covariance: sign included
    assert (dots > 1 - 1e-5).mean() > 0.999, (dots < 1 - 1e-5).sum()
    assert (np.abs(dots) > 1 - 1e-3).mean() > 0.9999
    # prior orientation
    pc2 = P.PointCloud(pts); pc2.normals = -ref
    pc2.estimate_normals(P.KDTreeSearchParamKNN(knn=knn))
    assert ((pc2.normals * ref).sum(1) < -1 + 1e-5).mean() > 0.999


def test_normals_hybrid_sparse_neighbourhoods(P, oracle, small_pair):
    """FGR's normal search (radius 0.2, max_nn 20) on raw NCLT clouds leaves 9-15 % of points with < 3
    neighbours -> normal (0,0,1)."""
    pts = small_pair["source"][:10000].copy()
    pc = P.PointCloud(pts)
    pc.estimate_normals(P.KDTreeSearchParamHybrid(radius=0.2, max_nn=20))
    ref = oracle.estimate_normals(pts, oracle.SEARCH_HYBRID, 20, 0.2)
    dev = pc.normals
    flat = (ref == np.array([0, 0, 1.0])).all(1)
    assert flat.mean() > 0.02
    assert ((dev == np.array([0, 0, 1.0])).all(1) == flat).mean() > 0.999
    dots = (dev * ref).sum(1)
    assert (dots > 1 - 1e-5).mean() > 0.995


def test_covariances_match_oracle(P, oracle, small_pair):
    pc = P.PointCloud(small_pair["source"]).voxel_down_sample(0.3)
    pts = pc.points
    pc.estimate_covariances(P.KDTreeSearchParamKNN(knn=30))
    ref = oracle.estimate_covariances(pts, oracle.SEARCH_KNN, 30)
    dev = pc.covariances
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    assert (np.abs(dev - ref) <= 2e-6 * scale + 1e-9).mean() > 0.999


def test_amostragem_multiescala_otimizada_and_random_down_sample(P, oracle, small_pair):
    """ALL_FUNCTIONS.py:233-254 (SURVEY f-4): the voxel cloud plus n-1 random subsets of it with the reference's size model, in the
    reference's order (subsets reversed, voxel cloud last); `random_down_sample` keeps int(n * ratio) points of the cloud, each once,
    and raises like Open3D outside (0, 1]."""
    pts = small_pair["source"]
    pc = P.PointCloud(pts)
    out = P.amostragem_multiescala_otimizada(pc, 4, 0.1, seed=3)
    vox = oracle.voxel_down_sample(pts, 0.1)
    assert len(out) == 4 and len(out[-1]) == len(vox)
    esc = np.array([0.1 + 0.1 * i for i in range(4)])
    pct = 1.18397758 * np.exp(-5.09388767 * esc) * len(pts) / len(vox)
    pct = (pct / np.linalg.norm(pct))[1:10]
    assert [len(c) for c in out[:3]] == [int(len(vox) * pct[i]) for i in (2, 1, 0)]
    allv = {tuple(r) for r in out[-1].points.tolist()}
    for c in out[:3]:
        rows = [tuple(r) for r in c.points.tolist()]
        assert len(set(rows)) == len(rows) and set(rows) <= allv          # a subset of the voxel cloud, no point twice
    again = P.amostragem_multiescala_otimizada(P.PointCloud(pts), 4, 0.1, seed=3)
    assert all(np.array_equal(a.points, b.points) for a, b in zip(out, again))
    half = pc.random_down_sample(0.5)
    assert len(half) == int(len(pts) * 0.5)
    assert len(pc.random_down_sample(1.0)) == len(pts)
    for bad in (0.0, -0.1, 1.5):
        with pytest.raises(RuntimeError):
            pc.random_down_sample(bad)
