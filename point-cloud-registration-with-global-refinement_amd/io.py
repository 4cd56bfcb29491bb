"""On-disk formats either side of the hot path.

* binary PCD v0.7 clouds as shipped under ``nuvens/nuvens_pre_processadas`` (read with
  ``o3d.io.read_point_cloud`` at ALL_FUNCTIONS.py:19,405 and script 1:127 / 2:169 / 3:289);
* 4x4 pose text files written by ``np.savetxt`` (script 1:176-177 ``%.10f``; script 2:250-253
  default ``%.18e``) and read back by ``np.loadtxt`` (script 2:173-175, script 3:298-304).
"""
from __future__ import annotations

import os

import numpy as np

_PCD_TYPES = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4",
              ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}


def pcd_point_count(path: str) -> int:
    """``POINTS`` of a PCD header (``WIDTH x HEIGHT`` without it) -- the header alone is read: what the cost-balanced sharding of a
    circuit needs of every cloud (``sharding.circuit_costs``)."""
    width = height = None
    with open(path, "rb") as f:
        for _ in range(64):
            line = f.readline()
            if not line:
                break
            key, _, rest = line.decode("ascii", "replace").strip().partition(" ")
            key = key.upper()
            if key == "POINTS":
                return int(rest.split()[0])
            if key == "WIDTH":
                width = int(rest.split()[0])
            if key == "HEIGHT":
                height = int(rest.split()[0])
            if key == "DATA":
                break
    if width is None or height is None:
        raise ValueError(f"{path}: no POINTS / WIDTH x HEIGHT in the PCD header")
    return width * height


def read_pcd_xyz(path: str) -> np.ndarray:
    """Return the ``x y z`` columns of a PCD file as an (N, 3) float32 array.

    Supports ``DATA binary`` and ``DATA ascii`` with any extra fields (e.g. ``rgb``).
    Non-finite points are dropped, as ``o3d.io.read_point_cloud`` does by default.
    """
    with open(path, "rb") as f:
        header = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PCD header")
            s = line.decode("ascii", "replace").strip()
            if not s or s.startswith("#"):
                continue
            key, _, rest = s.partition(" ")
            header[key.upper()] = rest.split()
            if key.upper() == "DATA":
                break
        fields = header["FIELDS"]
        sizes = [int(v) for v in header["SIZE"]]
        types = header["TYPE"]
        counts = [int(v) for v in header.get("COUNT", ["1"] * len(fields))]
        npts = int(header["POINTS"][0]) if "POINTS" in header else int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
        kind = header["DATA"][0].lower()
        dt = []
        for name, sz, ty, cnt in zip(fields, sizes, types, counts):
            base = _PCD_TYPES[(ty, sz)]
            dt.append((name, base) if cnt == 1 else (name, base, (cnt,)))
        dt = np.dtype(dt)
        if kind == "binary":
            raw = f.read(npts * dt.itemsize)
            if len(raw) < npts * dt.itemsize:
                raise ValueError(f"{path}: expected {npts} points, file is short")
            rec = np.frombuffer(raw, dtype=dt, count=npts)
        elif kind == "ascii":
            arr = np.loadtxt(f, dtype=np.float64, ndmin=2)
            rec = np.zeros(arr.shape[0], dtype=dt)
            col = 0
            for name, cnt in zip(fields, counts):
                rec[name] = arr[:, col] if cnt == 1 else arr[:, col:col + cnt]
                col += cnt
        else:
            raise ValueError(f"{path}: unsupported PCD DATA kind {kind!r}")
    xyz = np.stack([rec["x"], rec["y"], rec["z"]], axis=1).astype(np.float32)
    ok = np.isfinite(xyz).all(axis=1)
    return np.ascontiguousarray(xyz[ok])


def write_pcd_xyz(path: str, xyz: np.ndarray) -> None:
    """Write an (N,3) array as the same binary PCD v0.7 layout the reference ships."""
    xyz = np.ascontiguousarray(xyz, dtype="<f4").reshape(-1, 3)
    n = xyz.shape[0]
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
           f"COUNT 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(xyz.tobytes())


def read_pose(path: str) -> np.ndarray:
    T = np.loadtxt(path, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError(f"{path}: expected a 4x4 matrix, got {T.shape}")
    return T


def write_pose(path: str, T: np.ndarray, fmt: str = "%.18e") -> None:
    """``np.savetxt`` layout; ``fmt='%.10f'`` reproduces script 1:176-177."""
    np.savetxt(path, np.asarray(T, dtype=np.float64).reshape(4, 4), fmt=fmt)


def relative_pose_name(i: int, n_clouds: int, closure_as_read: bool = True) -> str:
    """File name of the pose registering cloud i+1 onto cloud i.

    The reference WRITES the loop closure (i = n-1) as ``pose_{n}_{n-1}.txt`` (script 1:177)
    but READS it as ``pose_0_{n-1}.txt`` (script 2:174, script 3:303); SURVEY.md App. C-9.
    ``closure_as_read=True`` (default) uses the name the downstream stages read.
    """
    if i == n_clouds - 1 and closure_as_read:
        return f"pose_0_{n_clouds - 1}.txt"
    return f"pose_{i + 1}_{i}.txt"


def load_relative_poses(folder: str, n_clouds: int) -> list:
    """Script 2:173-175 / script 3:302-304."""
    poses = [read_pose(os.path.join(folder, f"pose_{i + 1}_{i}.txt")) for i in range(n_clouds - 1)]
    poses.append(read_pose(os.path.join(folder, f"pose_0_{n_clouds - 1}.txt")))
    return poses
