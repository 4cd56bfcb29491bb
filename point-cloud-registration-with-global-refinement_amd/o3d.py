"""``import <package>.o3d as o3d`` gives the slice of the Open3D Python namespace that the reference's
hot path uses (ALL_FUNCTIONS.py:4 ``import open3d as o3d``), backed by libpcr_hip.so."""
from types import SimpleNamespace

import numpy as _np

from . import geometry as _g
from . import io as _io
from . import registration as _r


def _read_point_cloud(path):
    return _g.PointCloud(_io.read_pcd_xyz(path))


geometry = SimpleNamespace(PointCloud=_g.PointCloud, KDTreeSearchParamKNN=_g.KDTreeSearchParamKNN,
                           KDTreeSearchParamRadius=_g.KDTreeSearchParamRadius,
                           KDTreeSearchParamHybrid=_g.KDTreeSearchParamHybrid)
utility = SimpleNamespace(Vector3dVector=lambda a: _np.asarray(a, dtype=_np.float64).reshape(-1, 3))
io = SimpleNamespace(read_point_cloud=_read_point_cloud)
pipelines = SimpleNamespace(registration=_r)
