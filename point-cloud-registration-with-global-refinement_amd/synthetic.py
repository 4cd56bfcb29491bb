"""Synthetic NCLT-shaped LiDAR scenes for the headline benchmark.

There is no network and the reference ships only ~20k-point clouds, so the 200k-point workload BASELINE.json
quotes is synthesised: an outdoor block (ground with undulation, roofs, facades within +-15 deg of the axes,
poles / trunks / cars), points drawn uniformly per surface area and thinned with probability ~ 1/range
(LiDAR-like fall-off), Gaussian noise of 2 cm along the surface normal.  Source and target are INDEPENDENT
samples of the same scene; the source is moved by the inverse of a planted NCLT-like ego-motion and the
initial pose is that motion perturbed by 0.5 deg / 0.15 m (the median FGR->GICP correction, App. B.2).
Pure numpy; deterministic for a given seed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SEED = 20241008


@dataclass
class SyntheticPair:
    source: np.ndarray          # (N,3) float32
    target: np.ndarray          # (N,3) float32
    T_true: np.ndarray          # 4x4, source -> target
    T_init: np.ndarray          # 4x4, perturbed start
    voxel_sizes: list
    max_distances_script: list  # script-2 table rule  (2_MGICP...py:112-120)


def _rot(axis, ang):
    axis = np.asarray(axis, float); axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


class Scene:
    """Surface primitives of one block; ``sample(n, rng)`` draws a LiDAR-like cloud seen from the origin."""

    def __init__(self, seed: int = SEED, half_extent: float = 117.5, z_ground: float = -2.9):
        r = np.random.default_rng(seed)
        self.h = half_extent
        self.ground_pow = 2.2      # radius ~ u^p: p = 1 gives areal density ~ 1/r, larger p concentrates near the sensor
        self.falloff = 2.0         # acceptance ~ (r_min / range)^falloff on facades, roofs and clutter
        self.zg = z_ground
        self.phase = r.uniform(0, 2 * np.pi, 4)
        # facades: centre, yaw, width, height
        nf = int(56 * (half_extent / 117.5) ** 2)
        dist = r.uniform(5.0, min(120.0, half_extent), nf)
        bearing = r.uniform(0, 2 * np.pi, nf)
        self.f_c = np.stack([dist * np.cos(bearing), dist * np.sin(bearing)], 1)
        self.f_yaw = r.choice([0.0, np.pi / 2], nf) + np.deg2rad(r.uniform(-15, 15, nf))
        self.f_w = r.uniform(10.0, 40.0, nf)
        self.f_h = r.uniform(3.0, 11.5, nf)
        # roofs on 60 % of the facades: a slab behind the facade
        self.roof = r.random(nf) < 0.6
        self.r_depth = r.uniform(6.0, 15.0, nf)
        # clutter
        nc = int(400 * (half_extent / 117.5) ** 2)
        rad = np.sqrt(r.uniform(4.0 ** 2, (0.9 * half_extent) ** 2, nc))
        ang = r.uniform(0, 2 * np.pi, nc)
        self.c_xy = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
        self.c_kind = r.choice(3, nc, p=[0.45, 0.3, 0.25])            # pole, trunk, car
        self.c_yaw = r.uniform(0, np.pi, nc)
        self.c_h = np.where(self.c_kind == 0, r.uniform(4, 8, nc), np.where(self.c_kind == 1, r.uniform(2.5, 4, nc), 1.5))

    # ---- primitive samplers: return points and unit normals -------------------------------------
    def _ground(self, n, rng):
        # radial density ~ 1/range is obtained by sampling the radius uniformly
        rad = 2.0 + (self.h * np.sqrt(2) - 2.0) * rng.uniform(0, 1, n) ** self.ground_pow
        ang = rng.uniform(0, 2 * np.pi, n)
        x, y = rad * np.cos(ang), rad * np.sin(ang)
        ok = (np.abs(x) <= self.h) & (np.abs(y) <= self.h)
        x, y = x[ok], y[ok]
        z = self.zg + 0.05 * np.sin(0.07 * x + self.phase[0]) + 0.05 * np.sin(0.05 * y + self.phase[1])
        nrm = np.stack([-0.0035 * np.cos(0.07 * x + self.phase[0]), -0.0025 * np.cos(0.05 * y + self.phase[1]), np.ones_like(x)], 1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        return np.stack([x, y, z], 1), nrm

    def _facades(self, n, rng):
        area = self.f_w * self.f_h
        k = rng.choice(len(area), n, p=area / area.sum())
        u = rng.uniform(-0.5, 0.5, n) * self.f_w[k]
        v = rng.uniform(0, 1, n) * self.f_h[k]
        d = np.stack([np.cos(self.f_yaw[k]), np.sin(self.f_yaw[k])], 1)
        xy = self.f_c[k] + d * u[:, None]
        pts = np.stack([xy[:, 0], xy[:, 1], self.zg + v], 1)
        nrm = np.stack([-d[:, 1], d[:, 0], np.zeros(n)], 1)
        return pts, nrm

    def _roofs(self, n, rng):
        idx = np.nonzero(self.roof)[0]
        area = self.f_w[idx] * self.r_depth[idx]
        k = idx[rng.choice(len(idx), n, p=area / area.sum())]
        u = rng.uniform(-0.5, 0.5, n) * self.f_w[k]
        w = rng.uniform(0, 1, n) * self.r_depth[k]
        d = np.stack([np.cos(self.f_yaw[k]), np.sin(self.f_yaw[k])], 1)
        nn = np.stack([-d[:, 1], d[:, 0]], 1)
        # slab on the far side of the facade as seen from the sensor
        side = np.sign((self.f_c[k] * nn).sum(1))[:, None]
        xy = self.f_c[k] + d * u[:, None] + nn * side * w[:, None]
        pts = np.stack([xy[:, 0], xy[:, 1], self.zg + self.f_h[k]], 1)
        nrm = np.tile([0.0, 0.0, 1.0], (n, 1))
        return pts, nrm

    def _clutter(self, n, rng):
        size = np.where(self.c_kind == 0, 2 * np.pi * 0.15 * self.c_h, np.where(self.c_kind == 1, 2 * np.pi * 0.3 * self.c_h, 2 * (4.5 + 1.8) * 1.5 + 4.5 * 1.8))
        k = rng.choice(len(size), n, p=size / size.sum())
        pts = np.empty((n, 3)); nrm = np.empty((n, 3))
        cyl = self.c_kind[k] < 2
        r = np.where(self.c_kind[k] == 0, 0.15, 0.3)
        a = rng.uniform(0, 2 * np.pi, n)
        pts[cyl] = np.stack([self.c_xy[k, 0] + r * np.cos(a), self.c_xy[k, 1] + r * np.sin(a), self.zg + rng.uniform(0, 1, n) * self.c_h[k]], 1)[cyl]
        nrm[cyl] = np.stack([np.cos(a), np.sin(a), np.zeros(n)], 1)[cyl]
        car = ~cyl
        m = int(car.sum())
        if m:
            face = rng.choice(5, m, p=np.array([4.5 * 1.5, 4.5 * 1.5, 1.8 * 1.5, 1.8 * 1.5, 4.5 * 1.8]) / (2 * (4.5 + 1.8) * 1.5 + 4.5 * 1.8))
            lu = rng.uniform(-0.5, 0.5, m); lv = rng.uniform(0, 1, m)
            lx = np.where(face < 2, lu * 4.5, np.where(face < 4, np.where(face == 2, 2.25, -2.25), lu * 4.5))
            ly = np.where(face < 2, np.where(face == 0, 0.9, -0.9), np.where(face < 4, lu * 1.8, (lv - 0.5) * 1.8))
            lz = np.where(face < 4, lv * 1.5, 1.5)
            ln = np.zeros((m, 3))
            ln[face == 0] = [0, 1, 0]; ln[face == 1] = [0, -1, 0]; ln[face == 2] = [1, 0, 0]; ln[face == 3] = [-1, 0, 0]; ln[face == 4] = [0, 0, 1]
            c, s = np.cos(self.c_yaw[k][car]), np.sin(self.c_yaw[k][car])
            pts[car] = np.stack([self.c_xy[k][car, 0] + c * lx - s * ly, self.c_xy[k][car, 1] + s * lx + c * ly, self.zg + lz], 1)
            nrm[car] = np.stack([c * ln[:, 0] - s * ln[:, 1], s * ln[:, 0] + c * ln[:, 1], ln[:, 2]], 1)
        return pts, nrm

    def sample(self, n: int, rng, pre_voxel: float = 0.08, oversample: float = 3.0) -> np.ndarray:
        """n points shaped like the reference's PRE-PROCESSED clouds (ALL_FUNCTIONS.py:19-21 voxel-filters the raw
        scans): draw ``oversample*n`` raw returns, average them on a ``pre_voxel`` grid, keep n at random.  With the
        defaults a 200k cloud has NN spacing median ~0.09 m, planar range median ~16 m and voxel retention
        ~0.87/0.59/0.33 at 0.1/0.2/0.4 m (NCLT: 0.085 m, 16.5 m, 0.89/0.55/0.30)."""
        raw = self.sample_raw(int(oversample * n), rng).astype(np.float64)
        k = np.floor((raw - raw.min(0)) / pre_voxel).astype(np.int64)
        key = (k[:, 0] * 4000037 + k[:, 1]) * 4000037 + k[:, 2]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        starts = np.nonzero(np.r_[True, ks[1:] != ks[:-1]])[0]
        sums = np.add.reduceat(raw[order], starts, axis=0)
        mean = sums / np.diff(np.r_[starts, len(ks)])[:, None]
        if len(mean) < n:
            return self.sample(n, rng, pre_voxel, oversample * 1.6)
        return mean[rng.permutation(len(mean))[:n]].astype(np.float32)

    def sample_raw(self, n: int, rng, noise: float = 0.02, r_min: float = 4.0) -> np.ndarray:
        """n raw returns: ~35 % horizontal, ~45 % vertical planes, ~20 % clutter; density falls off with range."""
        want = {"h": int(0.35 * n), "v": int(0.45 * n)}
        want["c"] = n - want["h"] - want["v"]
        out = []
        for cls, samplers in (("h", (self._ground, self._roofs)), ("v", (self._facades,)), ("c", (self._clutter,))):
            got, need = [], want[cls]
            while need > 0:
                for j, smp in enumerate(samplers):
                    m = int(need * (12.0 if smp is not self._ground else 1.3) * (0.75 if (cls == "h" and j == 0) else (0.25 if cls == "h" else 1.0))) + 64
                    p, nr = smp(m, rng)
                    rng_ = np.linalg.norm(p, axis=1)
                    keep = np.ones(len(p), bool) if smp is self._ground else (rng.random(len(p)) < np.minimum(1.0, r_min / np.maximum(rng_, 1e-6)) ** self.falloff)
                    keep &= (np.abs(p[:, 0]) <= self.h) & (np.abs(p[:, 1]) <= self.h)
                    p = p[keep] + nr[keep] * rng.normal(0.0, noise, (int(keep.sum()), 1))
                    got.append(p)
                tot = sum(len(g) for g in got)
                need = want[cls] - tot
            g = np.concatenate(got)
            g = g[rng.permutation(len(g))[: want[cls]]]
            out.append(g)
        pts = np.concatenate(out)
        return pts[rng.permutation(len(pts))].astype(np.float32)


def planted_motion(rng):
    """NCLT-like per-frame ego-motion (App. B.2): yaw ~ N(0, 2 deg), roll/pitch ~ N(0, 0.3 deg),
    |t| log-normal with median 0.6 m (sigma 0.5), mostly in the ground plane."""
    yaw, roll, pitch = np.deg2rad(rng.normal(0, 2.0)), np.deg2rad(rng.normal(0, 0.3)), np.deg2rad(rng.normal(0, 0.3))
    R = _rot([0, 0, 1], yaw) @ _rot([0, 1, 0], pitch) @ _rot([1, 0, 0], roll)
    d = rng.normal(0, 1, 3) * np.array([1.0, 1.0, 0.05]); d /= np.linalg.norm(d)
    t = d * 0.6 * np.exp(rng.normal(0, 0.5))
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T


def perturbation(rng, ang_deg=0.5, dist=0.15):
    a = rng.normal(0, 1, 3); d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
    T = np.eye(4); T[:3, :3] = _rot(a, np.deg2rad(ang_deg)); T[:3, 3] = d * dist
    return T


def _scales(n_scales: int):
    if n_scales == 3:
        vox = [0.4, 0.2, 0.1]            # ALL_FUNCTIONS.py:260-264,274-275
        dst = [3 * 0.4, 2 * 0.2, 0.1]    # 2_MGICP...py:115
    else:
        vox = [0.1 + 0.1 * i for i in range(n_scales)][::-1]
        dst = {4: [3, 2.5, 2, 1], 5: [3, 2.5, 2, 1.5, 1]}[n_scales]
        dst = [a * b for a, b in zip(dst, vox)]
    return vox, dst


def make_pair(n_points: int = 200_000, seed: int = SEED, index: int = 0, n_scales: int = 3, cache_dir: str = None) -> SyntheticPair:
    """Config 2 (n_points=200k, 3 scales) / config 5 (2M, 5 scales). ``index`` selects the pair of a stream.
    The sampler is deterministic but slow (rejection sampling, ~20 s per 200k pair on one core), so the clouds are
    cached as .npz under ``cache_dir`` (default: $PCR_SYNTH_CACHE, else `<repo>/.synth_cache` if that directory exists, else
    the system temp dir); the cache holds exactly what
    the generator returns."""
    import os
    import tempfile
    vox, dst = _scales(n_scales)
    repo_cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".synth_cache")
    cache_dir = cache_dir or os.environ.get("PCR_SYNTH_CACHE") or (repo_cache if os.path.isdir(repo_cache) else os.path.join(tempfile.gettempdir(), "pcr_synth_cache"))
    path = os.path.join(cache_dir, f"pair_v1_{n_points}_{seed}_{index}.npz")
    if os.path.exists(path):
        try:
            d = np.load(path)
            return SyntheticPair(d["source"], d["target"], d["T_true"], d["T_init"], vox, dst)
        except Exception:
            pass
    half = 117.5 * np.sqrt(n_points / 200_000.0)
    scene = Scene(seed, half_extent=half)
    rng = np.random.default_rng([seed, 1000 + index])
    target = scene.sample(n_points, rng)
    moved = scene.sample(n_points, rng)
    T_true = planted_motion(rng)
    Ti = np.linalg.inv(T_true)
    source = (moved.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    T_init = perturbation(rng) @ T_true
    try:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = path + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, source=source, target=target, T_true=T_true, T_init=T_init)
        os.replace(tmp, path)
    except OSError:
        pass
    return SyntheticPair(source, target, T_true, T_init, vox, dst)


def tile_pair(p: SyntheticPair, copies: int, pitch: float = 260.0, n_scales: int = None) -> SyntheticPair:
    """A `copies`-times larger pair in about a second: the block is repeated on a square lattice of `pitch` metres (larger
    than the block, so tiles do not touch) and the SAME planted motion relates the two big clouds.  Point spacing, and so
    every per-scale retention ratio, stays that of the base pair -- the cheap stand-in for config 5's larger scene."""
    side = int(np.ceil(np.sqrt(copies)))
    offs = np.array([[(i % side) - (side - 1) / 2.0, (i // side) - (side - 1) / 2.0, 0.0] for i in range(copies)]) * pitch
    moved = p.source.astype(np.float64) @ p.T_true[:3, :3].T + p.T_true[:3, 3]                 # source in the target frame
    tgt = np.concatenate([p.target.astype(np.float64) + o for o in offs])
    mov = np.concatenate([moved + o for o in offs])
    Ti = np.linalg.inv(p.T_true)
    src = mov @ Ti[:3, :3].T + Ti[:3, 3]
    vox, dst = (p.voxel_sizes, p.max_distances_script) if n_scales is None else _scales(n_scales)
    return SyntheticPair(src.astype(np.float32), tgt.astype(np.float32), p.T_true, p.T_init, vox, dst)


def derive_pair(p: SyntheticPair, k: int, seed: int = SEED) -> SyntheticPair:
    """Distinct pair number k from a generated pair in milliseconds (the generator needs ~20 s per pair): BOTH clouds are moved by
    the same rigid motion G_k (yaw about the sensor axis and a shift in the ground plane), so the scene is seen at another
    heading and lands differently on every voxel grid, and the points are re-ordered.  The planted motion and the start become
    G T G^-1.  k = 0 is the pair itself."""
    if k == 0:
        return p
    rng = np.random.default_rng([seed, 77, k])
    G = np.eye(4)
    G[:3, :3] = _rot([0, 0, 1], rng.uniform(0, 2 * np.pi))
    G[:2, 3] = rng.uniform(-15.0, 15.0, 2)
    Gi = np.linalg.inv(G)

    def move(x):
        x = x.astype(np.float64) @ G[:3, :3].T + G[:3, 3]
        return x[rng.permutation(len(x))].astype(np.float32)
    return SyntheticPair(move(p.source), move(p.target), G @ p.T_true @ Gi, G @ p.T_init @ Gi, p.voxel_sizes, p.max_distances_script)


def make_loop(p: SyntheticPair, n_clouds: int = 4, copies: int = 5, keep: float = 0.97, jitter: float = 0.003, seed: int = SEED):
    """A closed circuit of `n_clouds` dense scans for BASELINE config 4 (the reference ships only two of the dense Courtyard scans):
    the world is the target cloud of `p` tiled `copies` times (copies x 200k points); scan i is a random `keep` share of it with
    `jitter` metres of isotropic noise, seen from the absolute pose A_i (A_0 = I, A_(i+1) = A_i M_i with NCLT-like motions M_i).
    Returns (clouds, A, T_true, T_init): pair i registers scan i+1 onto scan i (T_true[i] = A_i^-1 A_(i+1)), the last pair closes
    the loop (scan 0 onto scan n-1), T_init is the true relative pose perturbed by 0.5 deg / 0.15 m."""
    rng = np.random.default_rng([seed, 4, n_clouds, copies])
    world = tile_pair(p, copies).target.astype(np.float64)
    A = [np.eye(4)]
    for _ in range(n_clouds - 1):
        A.append(A[-1] @ planted_motion(rng))
    clouds = []
    for i in range(n_clouds):
        idx = np.nonzero(rng.random(len(world)) < keep)[0]
        w = world[idx] + rng.normal(0.0, jitter, (len(idx), 3))
        Ai = np.linalg.inv(A[i])
        clouds.append((w @ Ai[:3, :3].T + Ai[:3, 3]).astype(np.float32))
    T_true = [np.linalg.inv(A[i]) @ A[(i + 1) % n_clouds] for i in range(n_clouds)]
    T_init = [perturbation(rng) @ T for T in T_true]
    return clouds, A, T_true, T_init
