"""Stand-ins for the ``o3d.geometry`` objects the reference's hot path touches:
``PointCloud`` with device-resident float32 storage and ``KDTreeSearchParam{KNN,Radius,Hybrid}``.

Reference call sites: ``voxel_down_sample`` ALL_FUNCTIONS.py:293-294, ``remove_statistical_outlier``
:297-298, ``estimate_normals`` :182-183/:214-215/:301-302, ``estimate_covariances`` :216-217,
``get_min_bound/get_max_bound`` :1093-1097, ``transform`` :46, ``copy.deepcopy`` :289-290.
Every method dispatches to ``libpcr_hip.so``; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class KDTreeSearchParamKNN:
    def __init__(self, knn: int = 30):
        self.knn = int(knn)

    def _spec(self):
        return _lib.SEARCH_KNN, self.knn, 0.0


class KDTreeSearchParamRadius:
    def __init__(self, radius: float):
        self.radius = float(radius)

    def _spec(self):
        return _lib.SEARCH_RADIUS, 0, self.radius


class KDTreeSearchParamHybrid:
    def __init__(self, radius: float, max_nn: int):
        self.radius = float(radius)
        self.max_nn = int(max_nn)

    def _spec(self):
        return _lib.SEARCH_HYBRID, self.max_nn, self.radius


def _torch():
    import torch
    return torch


def _dev_f32(a, cols):
    """array-like / torch tensor -> contiguous float32 cuda tensor of shape (N, cols)."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the MI355X registration path cannot run (no CPU fallback)")
    if isinstance(a, torch.Tensor):
        t = a.to(device="cuda", dtype=torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).cuda()
    return t.reshape(-1, cols).contiguous()


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class PointCloud:
    """Device-resident point cloud (float32 xyz on the GPU; Open3D keeps float64 on the host)."""

    def __init__(self, points=None):
        self._xyz = None          # torch (N,3) float32 cuda
        self._nrm = None          # torch (N,3) float32 cuda
        self._cov = None          # torch (N,6) float32 cuda  (xx,xy,xz,yy,yz,zz)
        if points is not None:
            self.points = points

    # ---- o3d-like attribute surface -------------------------------------------------------------
    @property
    def points(self):
        if self._xyz is None:
            return np.zeros((0, 3), np.float64)
        return self._xyz.detach().cpu().numpy().astype(np.float64)

    @points.setter
    def points(self, value):
        self._xyz = _dev_f32(value, 3)
        self._nrm = None
        self._cov = None

    @property
    def normals(self):
        if self._nrm is None:
            return np.zeros((0, 3), np.float64)
        return self._nrm.detach().cpu().numpy().astype(np.float64)

    @normals.setter
    def normals(self, value):
        self._nrm = _dev_f32(value, 3)

    @property
    def covariances(self):
        if self._cov is None:
            return np.zeros((0, 3, 3), np.float64)
        c = self._cov.detach().cpu().numpy().astype(np.float64)
        out = np.empty((c.shape[0], 3, 3))
        out[:, 0, 0] = c[:, 0]; out[:, 0, 1] = out[:, 1, 0] = c[:, 1]; out[:, 0, 2] = out[:, 2, 0] = c[:, 2]
        out[:, 1, 1] = c[:, 3]; out[:, 1, 2] = out[:, 2, 1] = c[:, 4]; out[:, 2, 2] = c[:, 5]
        return out

    def __len__(self):
        return 0 if self._xyz is None else int(self._xyz.shape[0])

    def has_points(self):
        return len(self) > 0

    def has_normals(self):
        return self._nrm is not None and self._nrm.shape[0] == len(self) and len(self) > 0

    def has_covariances(self):
        return self._cov is not None and self._cov.shape[0] == len(self) and len(self) > 0

    def __deepcopy__(self, memo):
        out = PointCloud()
        out._xyz = None if self._xyz is None else self._xyz.clone()
        out._nrm = None if self._nrm is None else self._nrm.clone()
        out._cov = None if self._cov is None else self._cov.clone()
        return out

    def __repr__(self):
        return f"PointCloud with {len(self)} points."

    # ---- device views used by the registration layer ---------------------------------------------
    def device_xyz(self):
        if self._xyz is None:
            self._xyz = _torch().zeros((0, 3), dtype=_torch().float32, device="cuda")
        return self._xyz

    def device_normals(self):
        return self._nrm

    # ---- methods -----------------------------------------------------------------------------
    def get_min_bound(self):
        return self._bounds()[:3]

    def get_max_bound(self):
        return self._bounds()[3:]

    def _bounds(self):
        ctx = _lib.Context.current()
        b = (C.c_double * 6)()
        xyz = self.device_xyz()
        ctx.check(ctx.lib.pcr_bounds(ctx.handle, _ptr(xyz), C.c_int64(len(self)), b), "get_min_bound/get_max_bound")
        return np.array(b, dtype=np.float64)

    def voxel_down_sample(self, voxel_size: float) -> "PointCloud":
        ctx = _lib.Context.current()
        torch = _torch()
        n = len(self)
        xyz = self.device_xyz()
        out_xyz = torch.empty((max(n, 1), 3), dtype=torch.float32, device="cuda")
        has_n = self.has_normals()
        out_nrm = torch.empty((max(n, 1), 3), dtype=torch.float32, device="cuda") if has_n else None
        m = C.c_int64(0)
        ctx.check(ctx.lib.pcr_voxel_down_sample(ctx.handle, _ptr(xyz), _ptr(self._nrm if has_n else None), C.c_int64(n),
                                                C.c_double(voxel_size), _ptr(out_xyz), _ptr(out_nrm), C.byref(m)),
                  "voxel_down_sample")
        out = PointCloud()
        out._xyz = out_xyz[: m.value].contiguous()
        if has_n:
            out._nrm = out_nrm[: m.value].contiguous()
        return out

    def remove_statistical_outlier(self, nb_neighbors: int, std_ratio: float):
        ctx = _lib.Context.current()
        torch = _torch()
        n = len(self)
        xyz = self.device_xyz()
        keep = torch.zeros(max(n, 1), dtype=torch.uint8, device="cuda")
        idx = torch.empty(max(n, 1), dtype=torch.int64, device="cuda")
        m = C.c_int64(0)
        ctx.check(ctx.lib.pcr_remove_statistical_outlier(ctx.handle, _ptr(xyz), C.c_int64(n), C.c_int(nb_neighbors),
                                                         C.c_double(std_ratio), _ptr(keep), None, _ptr(idx), C.byref(m)),
                  "remove_statistical_outlier")
        index = idx[: m.value]
        return self.select_by_index(index), index.cpu().numpy().tolist()

    def select_by_index(self, indices, invert: bool = False) -> "PointCloud":
        torch = _torch()
        idx = indices if isinstance(indices, torch.Tensor) else torch.as_tensor(np.asarray(indices, dtype=np.int64), device="cuda")
        idx = idx.to(device="cuda", dtype=torch.int64)
        if invert:
            mask = torch.ones(len(self), dtype=torch.bool, device="cuda")
            mask[idx] = False
            idx = torch.nonzero(mask).reshape(-1)
        out = PointCloud()
        out._xyz = self.device_xyz()[idx].contiguous()
        if self.has_normals():
            out._nrm = self._nrm[idx].contiguous()
        if self.has_covariances():
            out._cov = self._cov[idx].contiguous()
        return out

    def random_down_sample(self, sampling_ratio: float, seed=None) -> "PointCloud":
        """``PointCloud.random_down_sample`` (ALL_FUNCTIONS.py:248): a uniformly random subset of ``int(n * sampling_ratio)`` points
        in shuffled order (Open3D shuffles the index list with a ``random_device``-seeded engine and keeps its head; ``seed`` makes
        the draw repeatable here).  Raises like Open3D for a ratio outside (0, 1]."""
        if not (0.0 < sampling_ratio <= 1.0):
            raise RuntimeError("Illegal sampling_ratio, sampling_ratio must be between 0 and 1.")
        torch = _torch()
        n = len(self)
        m = int(n * sampling_ratio)
        g = torch.Generator(device="cuda")
        if seed is None:
            g.seed()
        else:
            g.manual_seed(int(seed))
        return self.select_by_index(torch.randperm(n, device="cuda", generator=g)[:m])

    def estimate_normals(self, search_param=None, fast_normal_computation: bool = True):
        ctx = _lib.Context.current()
        torch = _torch()
        kind, knn, radius = (search_param or KDTreeSearchParamKNN(30))._spec()
        n = len(self)
        out = torch.empty((max(n, 1), 3), dtype=torch.float32, device="cuda")
        prior = self._nrm if self.has_normals() else None
        ctx.check(ctx.lib.pcr_estimate_normals(ctx.handle, _ptr(self.device_xyz()), C.c_int64(n), C.c_int(kind), C.c_int(knn),
                                               C.c_double(radius), _ptr(prior), _ptr(out)), "estimate_normals")
        self._nrm = out[:n].contiguous()

    def estimate_covariances(self, search_param=None):
        ctx = _lib.Context.current()
        torch = _torch()
        kind, knn, radius = (search_param or KDTreeSearchParamKNN(30))._spec()
        n = len(self)
        out = torch.empty((max(n, 1), 6), dtype=torch.float32, device="cuda")
        ctx.check(ctx.lib.pcr_estimate_covariances(ctx.handle, _ptr(self.device_xyz()), C.c_int64(n), C.c_int(kind), C.c_int(knn),
                                                   C.c_double(radius), _ptr(out)), "estimate_covariances")
        self._cov = out[:n].contiguous()

    def transform(self, T):
        """In place, like Open3D (points, normals; covariances rotated)."""
        torch = _torch()
        T = np.asarray(T, dtype=np.float64).reshape(4, 4)
        R = torch.as_tensor(T[:3, :3], dtype=torch.float64, device="cuda")
        t = torch.as_tensor(T[:3, 3], dtype=torch.float64, device="cuda")
        if len(self):
            self._xyz = (self._xyz.double() @ R.T + t).float().contiguous()
            if self.has_normals():
                self._nrm = (self._nrm.double() @ R.T).float().contiguous()
            if self.has_covariances():
                C3 = torch.as_tensor(self.covariances, dtype=torch.float64, device="cuda")
                C3 = R @ C3 @ R.T
                self._cov = torch.stack([C3[:, 0, 0], C3[:, 0, 1], C3[:, 0, 2], C3[:, 1, 1], C3[:, 1, 2], C3[:, 2, 2]], 1).float().contiguous()
        return self
