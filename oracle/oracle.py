"""ctypes binding of the CPU ORACLE (``oracle/libpcr_oracle.so``).

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg; the product package never imports this module.
See ``pcr_oracle.h`` for the reference call sites each function restates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpcr_oracle.so")

SEARCH_KNN, SEARCH_RADIUS, SEARCH_HYBRID = 0, 1, 2
LOSS_L2, LOSS_L1, LOSS_GM = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in os.listdir(_HERE) if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class _Result(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("fitness", C.c_double), ("inlier_rmse", C.c_double),
                ("n_corr", C.c_int64), ("iterations", C.c_int32), ("converged", C.c_int32)]


class _ScaleStats(C.Structure):
    _fields_ = [("n_voxel", C.c_int64 * 2), ("n_clean", C.c_int64 * 2), ("icp", _Result)]


class _FgrOption(C.Structure):
    _fields_ = [("division_factor", C.c_double), ("use_absolute_scale", C.c_int32), ("decrease_mu", C.c_int32),
                ("maximum_correspondence_distance", C.c_double), ("iteration_number", C.c_int32),
                ("tuple_scale", C.c_double), ("maximum_tuple_count", C.c_int32), ("tuple_test", C.c_int32),
                ("seed", C.c_uint64)]


@dataclass
class Result:
    transformation: np.ndarray
    fitness: float
    inlier_rmse: float
    n_corr: int
    iterations: int
    converged: bool
    correspondence_set: np.ndarray | None = None
    extra: dict = field(default_factory=dict)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _f64(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def _res(r: _Result, corr=None) -> Result:
    return Result(np.array(r.T, dtype=np.float64).reshape(4, 4), r.fitness, r.inlier_rmse, int(r.n_corr),
                  int(r.iterations), bool(r.converged), corr)


def set_num_threads(n: int) -> int:
    return lib().orc_set_num_threads(int(n))


def set_sum_chunk(chunk: int) -> int:
    """Size of the fixed summation chunks of the GICP normal equations (default 256); returns the previous value.  Changing it
    changes NOTHING but the float64 summation tree -- tests use it to measure the L1-IRLS noise floor of the algorithm itself."""
    return lib().orc_set_sum_chunk(int(chunk))


def l1_spread(run, chunks=(64, 1024, 4096)):
    """Poses of `run()` under the default and the given summation chunkings -> (default result, max rotation [rad] and translation
    [m] distance of any variant from the default).  The reference's L1 weights 1/|r| make the end pose chaotic in the last bits
    of the sums, so this spread -- not a hand-picked constant -- is what bounds a device-vs-oracle comparison on L1."""
    base = run()
    ang = dt = 0.0
    base.extra["variant_poses"] = []           # end poses of the other chunkings: more samples of the same scatter (tests/conftest.py)
    for c in chunks:
        old = set_sum_chunk(c)
        try:
            r = run()
        finally:
            set_sum_chunk(old)
        A, B = base.transformation, r.transformation
        base.extra["variant_poses"].append(B.copy())
        ang = max(ang, float(2.0 * np.arcsin(min(1.0, np.linalg.norm(A[:3, :3] - B[:3, :3]) / (2.0 * np.sqrt(2.0))))))
        dt = max(dt, float(np.linalg.norm(A[:3, 3] - B[:3, 3])))
    return base, ang, dt


def get_num_threads() -> int:
    return lib().orc_get_num_threads()


def knn(pts, queries, k, radius=0.0):
    pts = _f64(pts); queries = _f64(queries)
    dim = pts.shape[1]
    nq = queries.shape[0]
    idx = np.empty((nq, k), np.int64); d2 = np.empty((nq, k), np.float64); cnt = np.empty(nq, np.int32)
    _check(lib().orc_knn(_p(pts), C.c_int64(pts.shape[0]), dim, _p(queries), C.c_int64(nq), int(k), C.c_double(radius),
                         _p(idx), _p(d2), _p(cnt)), "knn")
    return idx, d2, cnt


def voxel_down_sample(xyz, voxel, normals=None):
    xyz = _f64(xyz, 3)
    n = xyz.shape[0]
    out = np.empty((max(n, 1), 3), np.float64)
    nin = _f64(normals, 3) if normals is not None else None
    nout = np.empty((max(n, 1), 3), np.float64) if normals is not None else None
    m = C.c_int64(0)
    _check(lib().orc_voxel_down_sample(_p(xyz), C.c_int64(n), C.c_double(voxel), _p(out), C.byref(m), _p(nin), _p(nout)),
           "voxel_down_sample")
    if normals is not None:
        return out[: m.value].copy(), nout[: m.value].copy()
    return out[: m.value].copy()


def remove_statistical_outlier(xyz, nb_neighbors, std_ratio):
    xyz = _f64(xyz, 3)
    n = xyz.shape[0]
    keep = np.zeros(max(n, 1), np.uint8); avg = np.empty(max(n, 1), np.float64)
    mu = C.c_double(0); sd = C.c_double(0)
    _check(lib().orc_remove_statistical_outlier(_p(xyz), C.c_int64(n), int(nb_neighbors), C.c_double(std_ratio), _p(keep),
                                                _p(avg), C.byref(mu), C.byref(sd)), "remove_statistical_outlier")
    return keep[:n].astype(bool), avg[:n], mu.value, sd.value


def estimate_covariances(xyz, mode=SEARCH_KNN, knn=30, radius=0.0):
    xyz = _f64(xyz, 3)
    n = xyz.shape[0]
    cov = np.empty((max(n, 1), 9), np.float64)
    _check(lib().orc_estimate_covariances(_p(xyz), C.c_int64(n), int(mode), int(knn), C.c_double(radius), _p(cov)),
           "estimate_covariances")
    return cov[:n].reshape(-1, 3, 3)


def estimate_normals(xyz, mode=SEARCH_KNN, knn=20, radius=0.0, prior=None, cov=None):
    xyz = _f64(xyz, 3)
    n = xyz.shape[0]
    nr = np.empty((max(n, 1), 3), np.float64)
    pr = _f64(prior, 3) if prior is not None else None
    cv = _f64(cov, 9) if cov is not None else None
    _check(lib().orc_estimate_normals(_p(xyz), C.c_int64(n), int(mode), int(knn), C.c_double(radius), _p(pr), _p(cv), _p(nr)),
           "estimate_normals")
    return nr[:n]


def fast_eigen3x3(cov):
    cov = _f64(cov).reshape(9)
    out = np.empty(3, np.float64)
    lib().orc_fast_eigen3x3(_p(cov), _p(out))
    return out


def covariances_from_normals(normals, eps=1e-3):
    nr = _f64(normals, 3)
    cov = np.empty((max(nr.shape[0], 1), 9), np.float64)
    _check(lib().orc_covariances_from_normals(_p(nr), C.c_int64(nr.shape[0]), C.c_double(eps), _p(cov)), "cov_from_normals")
    return cov[: nr.shape[0]].reshape(-1, 3, 3)


def gicp_linearize(src_xyz, src_cov, tgt_xyz, tgt_cov, corr, loss=LOSS_L1, loss_k=1.0):
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3); cs = _f64(src_cov, 9); ct = _f64(tgt_cov, 9)
    corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
    JTJ = np.empty(36); JTr = np.empty(6); r2 = C.c_double(0)
    _check(lib().orc_gicp_linearize(_p(s), _p(cs), _p(t), _p(ct), _p(corr), C.c_int64(corr.shape[0]), int(loss),
                                    C.c_double(loss_k), _p(JTJ), _p(JTr), C.byref(r2)), "gicp_linearize")
    return JTJ.reshape(6, 6), JTr, r2.value


def find_correspondences(src_xyz, tgt_xyz, max_dist):
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3)
    corr = np.empty((max(s.shape[0], 1), 2), np.int32)
    nc = C.c_int64(0); fit = C.c_double(0); rm = C.c_double(0)
    _check(lib().orc_find_correspondences(_p(s), C.c_int64(s.shape[0]), _p(t), C.c_int64(t.shape[0]), C.c_double(max_dist),
                                          _p(corr), C.byref(nc), C.byref(fit), C.byref(rm)), "find_correspondences")
    return corr[: nc.value].copy(), fit.value, rm.value


def solve_update(JTJ, JTr):
    A = _f64(JTJ).reshape(36); b = _f64(JTr).reshape(6)
    T = np.empty(16)
    rc = lib().orc_solve_update(_p(A), _p(b), _p(T))
    return T.reshape(4, 4), rc


def registration_gicp(src_xyz, tgt_xyz, max_dist, T0, src_normals=None, tgt_normals=None, src_cov=None, tgt_cov=None,
                      loss=LOSS_L1, loss_k=1.0, eps=1e-3, rel_fitness=1e-6, rel_rmse=1e-6, max_it=30, want_trace=False):
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3)
    sn = _f64(src_normals, 3) if src_normals is not None else None
    tn = _f64(tgt_normals, 3) if tgt_normals is not None else None
    sc = _f64(src_cov, 9) if src_cov is not None else None
    tc = _f64(tgt_cov, 9) if tgt_cov is not None else None
    T0 = _f64(T0).reshape(16)
    r = _Result()
    corr = np.empty((max(s.shape[0], 1), 2), np.int32)
    trace = np.full((max_it + 1, 2), np.nan) if want_trace else None
    _check(lib().orc_registration_gicp(_p(s), _p(sn), _p(sc), C.c_int64(s.shape[0]), _p(t), _p(tn), _p(tc),
                                       C.c_int64(t.shape[0]), C.c_double(max_dist), _p(T0), int(loss), C.c_double(loss_k),
                                       C.c_double(eps), C.c_double(rel_fitness), C.c_double(rel_rmse), int(max_it),
                                       C.byref(r), _p(corr), _p(trace)), "registration_gicp")
    out = _res(r, corr[: r.n_corr].copy())
    if want_trace:
        out.extra["trace"] = trace
    return out


def multiscale_gicp(src_xyz, tgt_xyz, voxels, dists, T0, src_normals=None, tgt_normals=None, sor_k=30, sor_std=1.0,
                    normal_k=20, loss=LOSS_L1, loss_k=1.0, eps=1e-3, rel_fitness=1e-6, rel_rmse=1e-6, max_it=100):
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3)
    sn = _f64(src_normals, 3) if src_normals is not None else None
    tn = _f64(tgt_normals, 3) if tgt_normals is not None else None
    vox = _f64(voxels).reshape(-1); dst = _f64(dists).reshape(-1)
    assert vox.size == dst.size
    T0 = _f64(T0).reshape(16)
    stats = (_ScaleStats * vox.size)()
    corr = np.empty((max(s.shape[0], 1), 2), np.int32)
    _check(lib().orc_multiscale_gicp(_p(s), _p(sn), C.c_int64(s.shape[0]), _p(t), _p(tn), C.c_int64(t.shape[0]), _p(vox),
                                     _p(dst), int(vox.size), int(sor_k), C.c_double(sor_std), int(normal_k), _p(T0),
                                     int(loss), C.c_double(loss_k), C.c_double(eps), C.c_double(rel_fitness),
                                     C.c_double(rel_rmse), int(max_it), stats, _p(corr)), "multiscale_gicp")
    last = stats[vox.size - 1].icp
    out = _res(last, corr[: last.n_corr].copy())
    out.extra["scales"] = [dict(n_voxel=tuple(st.n_voxel), n_clean=tuple(st.n_clean), iterations=int(st.icp.iterations),
                                fitness=st.icp.fitness, inlier_rmse=st.icp.inlier_rmse, converged=bool(st.icp.converged), n_corr=int(st.icp.n_corr),
                                T=np.array(st.icp.T).reshape(4, 4)) for st in stats]
    return out


def evaluate_registration(src_xyz, tgt_xyz, max_dist, T):
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3); T = _f64(T).reshape(16)
    r = _Result(); corr = np.empty((max(s.shape[0], 1), 2), np.int32)
    _check(lib().orc_evaluate_registration(_p(s), C.c_int64(s.shape[0]), _p(t), C.c_int64(t.shape[0]), C.c_double(max_dist),
                                           _p(T), C.byref(r), _p(corr)), "evaluate_registration")
    return _res(r, corr[: r.n_corr].copy())


def information_matrix(src_xyz, tgt_xyz, max_dist, T):
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3); T = _f64(T).reshape(16)
    info = np.empty(36)
    _check(lib().orc_information_matrix(_p(s), C.c_int64(s.shape[0]), _p(t), C.c_int64(t.shape[0]), C.c_double(max_dist),
                                        _p(T), _p(info)), "information_matrix")
    return info.reshape(6, 6)


def compute_fpfh(xyz, normals, mode=SEARCH_HYBRID, knn=200, radius=1.0):
    p = _f64(xyz, 3); nr = _f64(normals, 3)
    feat = np.zeros((max(p.shape[0], 1), 33), np.float64)
    _check(lib().orc_compute_fpfh(_p(p), _p(nr), C.c_int64(p.shape[0]), int(mode), int(knn), C.c_double(radius), _p(feat)),
           "compute_fpfh")
    return feat[: p.shape[0]]


def registration_fgr(src_xyz, src_feat, tgt_xyz, tgt_feat, division_factor=1.4, use_absolute_scale=False, decrease_mu=True,
                     maximum_correspondence_distance=0.025, iteration_number=64, tuple_scale=0.95,
                     maximum_tuple_count=1000, tuple_test=True, seed=0):
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3); fs = _f64(src_feat, 33); ft = _f64(tgt_feat, 33)
    opt = _FgrOption(division_factor, int(use_absolute_scale), int(decrease_mu), maximum_correspondence_distance,
                     int(iteration_number), tuple_scale, int(maximum_tuple_count), int(tuple_test), int(seed))
    r = _Result(); ncross = C.c_int64(0); ntup = C.c_int64(0)
    _check(lib().orc_registration_fgr(_p(s), _p(fs), C.c_int64(s.shape[0]), _p(t), _p(ft), C.c_int64(t.shape[0]),
                                      C.byref(opt), C.byref(r), C.byref(ncross), C.byref(ntup)), "registration_fgr")
    out = _res(r)
    out.extra.update(n_cross=ncross.value, n_tuple_corr=ntup.value)
    return out


def radius_from_cloud_pair(src_xyz, tgt_xyz) -> float:
    """ALL_FUNCTIONS.py:1092-1101: mean over the two clouds of the cube root of the AABB volume."""
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3)
    d1 = s.max(0) - s.min(0); d2 = t.max(0) - t.min(0)
    return float(((d1[0] * d1[1] * d1[2]) ** (1 / 3) + (d2[0] * d2[1] * d2[2]) ** (1 / 3)) / 2)


def registro_fgr(src_xyz, tgt_xyz, voxel_size, use_absolute_scale=True, seed=0, src_prior=None, tgt_prior=None):
    """``registro_FGR`` (ALL_FUNCTIONS.py:178-203; script 1:41-66 with use_absolute_scale=False) composed from the oracle's
    restatements of the five Open3D calls.  ``extra`` carries the normals the reference leaves on the clouds and the features."""
    s = _f64(src_xyz, 3); t = _f64(tgt_xyz, 3)
    n_pontos = int((len(s) + len(t)) / 2)
    sn = estimate_normals(s, SEARCH_HYBRID, 20, 2 * voxel_size, prior=src_prior)
    tn = estimate_normals(t, SEARCH_HYBRID, 20, 2 * voxel_size, prior=tgt_prior)
    fs = compute_fpfh(s, sn, SEARCH_HYBRID, 200, 10 * voxel_size)
    ft = compute_fpfh(t, tn, SEARCH_HYBRID, 200, 10 * voxel_size)
    out = registration_fgr(s, fs, t, ft, 1.4, use_absolute_scale, True, 2 * voxel_size, 300, 0.95, int(n_pontos * 0.2), True, seed)
    out.extra.update(src_normals=sn, tgt_normals=tn, src_feat=fs, tgt_feat=ft)
    return out
