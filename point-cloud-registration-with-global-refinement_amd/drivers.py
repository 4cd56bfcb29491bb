"""Script-equivalent drivers over a closed circuit of clouds: the three stages the reference runs as
module-level code with hard-wired constants, as functions with the same data flow and on-disk formats.

* stage 1 = `1_FGR_pairwise_registration_in_NCLT_dataset.py:125-177`: FGR on every pair of the circuit, poses saved
  as `pose_{i+1}_{i}.txt` with `%.10f`;
* stage 2 = `2_MGICP_refinement_in_NCLT_dataset.py:166-253`: multiscale GICP refinement of every pair from the stage-1
  poses (5 scales, 100 iterations per scale);
* stage 3 = `3_Global_Optimizations_in_NCLT_dataset.py:296-364`: LUM / SLERP / SLERP+LUM on the relative poses
  (host side, `refinement.py`).

Pair i registers cloud i+1 onto cloud i; the last pair closes the loop (cloud 0 onto cloud n-1).  Pairs are independent:
under `torch.distributed` every rank takes a contiguous block of them (`sharding.partition`), keeps `inflight` pairs in
flight on its GPU (library worker threads: `pcr_register_pairs_plan`) and ONE all-gather leaves the ordered pose table on every rank.
Nothing here is on the measured hot path; it only feeds it.
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import functions, io, refinement, sharding
from .geometry import PointCloud


def load_circuit(cloud_dir: str, n_clouds: int, pattern: str = "s{i}.pcd", indices=None) -> dict:
    """`{i: PointCloud}` for the requested cloud indices (all by default): `o3d.io.read_point_cloud(f".../s{i}.pcd")`."""
    want = range(n_clouds) if indices is None else sorted(set(indices))
    return {i: PointCloud(io.read_pcd_xyz(os.path.join(cloud_dir, pattern.format(i=i)))) for i in want}


def _shard(n_pairs: int, cloud_dir: str = None, pattern: str = "s{i}.pcd"):
    """This rank's contiguous block of the circuit's pairs, balanced by the points of the pairs' clouds (PCD headers only; every rank
    computes the same cuts): the time of a sharded stage is its slowest rank's."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        costs = None
        if cloud_dir is not None and dist.get_world_size() > 1:
            costs = sharding.circuit_costs([io.pcd_point_count(os.path.join(cloud_dir, pattern.format(i=i))) for i in range(n_pairs)])
        return sharding.partition(n_pairs, dist.get_world_size(), dist.get_rank(), costs), dist.get_rank()
    return range(n_pairs), 0


def _clouds_of(pairs, n_clouds):
    need = set()
    for i in pairs:
        s, t = sharding.circuit_pair(i, n_clouds)
        need.update((s, t))
    return need


def _gather(results, n_pairs):
    import torch
    import torch.distributed as dist
    recs = np.stack([sharding.pack_record(i, r) for i, r, _ in results]) if results else np.zeros((0, sharding.RECORD_DOUBLES))
    on_gpu = torch.cuda.is_available() and not (dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo")
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else None          # RCCL gathers device tensors, gloo (tests, rehearsals) host ones
    table = sharding.gather_records(recs, n_pairs, device=dev)
    return [sharding.unpack_record(r) for r in table]


def stage1_fgr(cloud_dir: str, out_dir: str, n_clouds: int, voxel_size: float = 0.1, pattern: str = "s{i}.pcd",
               inflight: int = 8, seed=None, verbose: bool = True) -> list:
    """FGR over the circuit (script 1).  Returns the n relative poses; rank 0 writes them under `out_dir` with the
    names the later stages READ (`pose_0_{n-1}.txt` for the closure) and `%.10f` (S1:177)."""
    mine, rank = _shard(n_clouds, cloud_dir, pattern)
    clouds = load_circuit(cloud_dir, n_clouds, pattern, _clouds_of(mine, n_clouds))

    # one library call for the rank's whole block of pairs (stage FGR): the library keeps `inflight` of them in flight
    from . import registration as reg
    mine = list(mine)
    batch = [(clouds[sharding.circuit_pair(i, n_clouds)[0]], clouds[sharding.circuit_pair(i, n_clouds)[1]], None) for i in mine]
    if verbose:
        for i in mine:
            s, t = sharding.circuit_pair(i, n_clouds)
            print(f"Registering cloud {s} in cloud {t}")
    res = reg.register_pairs_plan(batch, "fgr", inflight=inflight, fgr_voxel_size=voxel_size, fgr_use_absolute_scale=False,
                                  fgr_seed=None if seed is None else seed + (mine[0] if mine else 0))
    table = _gather([(i, r, 0.0) for i, r in zip(mine, res)], n_clouds)
    poses = [r["transformation"] for r in table]
    if rank == 0 and out_dir:
        os.makedirs(out_dir, exist_ok=True)
        for i, T in enumerate(poses):
            io.write_pose(os.path.join(out_dir, io.relative_pose_name(i, n_clouds)), T, fmt="%.10f")
    return poses


def stage2_mgicp(cloud_dir: str, init_dir: str, out_dir: str, n_clouds: int, n_scales: int = 5, iterations: int = 100,
                 pattern: str = "s{i}.pcd", inflight: int = 4, absolute_dir: str = None, verbose: bool = True):
    """Multiscale GICP over the circuit from the stage-1 poses (script 2).  Returns (relative poses, absolute poses,
    per-pair records); rank 0 writes `pose_{i+1}_{i}.txt` (+ `pose_0_{n-1}.txt`) and, if asked, `pose{i}.txt`."""
    initial_T = io.load_relative_poses(init_dir, n_clouds)
    mine, rank = _shard(n_clouds, cloud_dir, pattern)
    clouds = load_circuit(cloud_dir, n_clouds, pattern, _clouds_of(mine, n_clouds))

    # one library call for the rank's whole block of pairs: the library keeps `inflight` of them in flight
    from . import registration as reg
    vox = functions.script2.create_scales(n_scales)
    dst = functions.script2.max_correspondence_distances(vox)
    mine = list(mine)
    batch = [(clouds[sharding.circuit_pair(i, n_clouds)[0]], clouds[sharding.circuit_pair(i, n_clouds)[1]], initial_T[i]) for i in mine]
    t0 = time.perf_counter()
    res = reg.register_pairs(batch, vox, dst, reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()),
                             reg.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=iterations), inflight=inflight,
                             group=None)            # lockstep groups sized by the clouds (registration.register_pairs_plan)
    dt = time.perf_counter() - t0
    if verbose:
        for i, r in zip(mine, res):
            s, t = sharding.circuit_pair(i, n_clouds)
            print(f"Pair {s}->{t} RMSE: {round(r.inlier_rmse, 3)} m")
        print(f"{len(mine)} pairs in {dt:.3f} s")
    table = _gather([(i, r, 0.0) for i, r in zip(mine, res)], n_clouds)
    rel = [r["transformation"] for r in table]
    ab = refinement.poses_relativas_para_absolutas(rel)
    if rank == 0 and out_dir:
        os.makedirs(out_dir, exist_ok=True)
        for i, T in enumerate(rel):
            io.write_pose(os.path.join(out_dir, io.relative_pose_name(i, n_clouds)), T)
        if absolute_dir:
            os.makedirs(absolute_dir, exist_ok=True)
            for i, T in enumerate(ab):
                io.write_pose(os.path.join(absolute_dir, f"pose{i}.txt"), T)
    return rel, ab, table


def stage12_fgr_mgicp(cloud_dir: str, out_dir: str, n_clouds: int, voxel_size: float = 0.1, n_scales: int = 5, iterations: int = 100,
                      pattern: str = "s{i}.pcd", inflight: int = 4, fgr_dir: str = None, seed=None, verbose: bool = True):
    """BASELINE config 3 as ONE plan per rank: scripts 1 and 2 back to back (1_FGR...py:134-147 then 2_MGICP...py:187-214) without the
    pose files in between -- every pair goes FGR (script-1 parameters: abs-scale False) -> 5-scale GICP (script-2 table, L1, 1e-6 / 1e-6 /
    `iterations`) from the FGR pose inside one `register_pairs_plan(stage="fgr+gicp")` call: the FGR stage runs over the whole block with
    many pairs in flight, the GICP stage in lockstep groups sized by the clouds.  The clouds are reloaded between the stages in the
    reference (script 2 reads the PCDs again), so the FGR normals are NOT an orientation prior here.  Returns (relative GICP poses,
    relative FGR poses, records); rank 0 writes `pose_{i+1}_{i}.txt` under `out_dir` (and the FGR poses under `fgr_dir`, `%.10f`)."""
    mine, rank = _shard(n_clouds, cloud_dir, pattern)
    clouds = load_circuit(cloud_dir, n_clouds, pattern, _clouds_of(mine, n_clouds))
    from . import registration as reg
    vox = functions.script2.create_scales(n_scales)
    dst = functions.script2.max_correspondence_distances(vox)
    mine = list(mine)
    batch = [(clouds[sharding.circuit_pair(i, n_clouds)[0]], clouds[sharding.circuit_pair(i, n_clouds)[1]], None) for i in mine]
    t0 = time.perf_counter()
    res = reg.register_pairs_plan(batch, "fgr+gicp", vox, dst, reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()),
                                  reg.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=iterations), inflight=inflight,
                                  fgr_voxel_size=voxel_size, fgr_use_absolute_scale=False,
                                  fgr_seed=None if seed is None else seed + (mine[0] if mine else 0), prior_from_fgr=False, group=None) if batch else []
    dt = time.perf_counter() - t0
    if verbose:
        print(f"{len(mine)} pairs (FGR + {n_scales}-scale GICP) in {dt:.3f} s")
    table = _gather([(i, r, 0.0) for i, r in zip(mine, res)], n_clouds)
    ftable = _gather([(i, r.fgr, 0.0) for i, r in zip(mine, res)], n_clouds)
    rel = [r["transformation"] for r in table]
    rel_fgr = [r["transformation"] for r in ftable]
    if rank == 0:
        for d, poses, fmt in ((out_dir, rel, "%.18e"), (fgr_dir, rel_fgr, "%.10f")):
            if d:
                os.makedirs(d, exist_ok=True)
                for i, T in enumerate(poses):
                    io.write_pose(os.path.join(d, io.relative_pose_name(i, n_clouds)), T, fmt=fmt)
    return rel, rel_fgr, table


def stage3_refine(rel_dir: str, n_clouds: int, out_dir: str = None, groundtruth_dir: str = None) -> dict:
    """Host-side global refinement of the circuit (script 3, steps 8-9): absolute poses by LUM, SLERP and SLERP+LUM
    (the script-3 variants), optionally written as `out_dir/<method>/pose{i}.txt` and compared with a ground truth."""
    rel = io.load_relative_poses(rel_dir, n_clouds)
    out = {
        "plain": refinement.poses_relativas_para_absolutas(rel),
        "LUM": refinement.script3.reconstruir_Ts_para_origem_LUM(rel),
        "SLERP": refinement.script3.reconstruir_Ts_para_origem_SLERP(rel),
        "SLERP_LUM": refinement.script3.reconstruir_Ts_para_origem_SLERP_LUM(rel),
        "closure": refinement.Calcular_Erro_LoopClosure(rel),
    }
    if out_dir:
        for name in ("LUM", "SLERP", "SLERP_LUM"):
            d = os.path.join(out_dir, name)
            os.makedirs(d, exist_ok=True)
            for i, T in enumerate(out[name]):
                io.write_pose(os.path.join(d, f"pose{i}.txt"), T)
    if groundtruth_dir:
        gt = [io.read_pose(os.path.join(groundtruth_dir, f"pose{i}.txt")) for i in range(n_clouds)]
        out["errors"] = {name: refinement.script3.subtract_squared_poses(gt, out[name]) for name in ("plain", "LUM", "SLERP", "SLERP_LUM")}
    return out


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="circuit drivers (stages 1-3 of the reference's scripts)")
    sub = ap.add_subparsers(dest="stage", required=True)
    a = sub.add_parser("stage1"); a.add_argument("--clouds", required=True); a.add_argument("--out", required=True)
    a.add_argument("--n", type=int, required=True); a.add_argument("--voxel", type=float, default=0.1); a.add_argument("--inflight", type=int, default=8)
    a.add_argument("--seed", type=int, default=None)
    b = sub.add_parser("stage2"); b.add_argument("--clouds", required=True); b.add_argument("--init", required=True); b.add_argument("--out", required=True)
    b.add_argument("--absolute", default=None); b.add_argument("--n", type=int, required=True); b.add_argument("--scales", type=int, default=5)
    b.add_argument("--iterations", type=int, default=100); b.add_argument("--inflight", type=int, default=4)
    d = sub.add_parser("stage12"); d.add_argument("--clouds", required=True); d.add_argument("--out", required=True); d.add_argument("--fgr-out", default=None)
    d.add_argument("--n", type=int, required=True); d.add_argument("--voxel", type=float, default=0.1); d.add_argument("--scales", type=int, default=5)
    d.add_argument("--iterations", type=int, default=100); d.add_argument("--inflight", type=int, default=4); d.add_argument("--seed", type=int, default=None)
    c = sub.add_parser("stage3"); c.add_argument("--relative", required=True); c.add_argument("--n", type=int, required=True)
    c.add_argument("--out", default=None); c.add_argument("--groundtruth", default=None)
    args = ap.parse_args(argv)
    sharded = ("stage1", "stage2", "stage12")
    if args.stage in sharded and "RANK" in os.environ:
        import torch
        import torch.distributed as dist
        # PCR_REHEARSE=1: every rank on device 0 and the gather over gloo -- the N > 1 path on a one-GPU box (tests/test_drivers.py)
        rehearse = os.environ.get("PCR_REHEARSE") == "1"
        torch.cuda.set_device(0 if rehearse else int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearse else "nccl")
    if args.stage == "stage1":
        stage1_fgr(args.clouds, args.out, args.n, args.voxel, inflight=args.inflight, seed=args.seed)
    elif args.stage == "stage2":
        stage2_mgicp(args.clouds, args.init, args.out, args.n, args.scales, args.iterations, inflight=args.inflight, absolute_dir=args.absolute)
    elif args.stage == "stage12":
        stage12_fgr_mgicp(args.clouds, args.out, args.n, args.voxel, args.scales, args.iterations, inflight=args.inflight, fgr_dir=args.fgr_out, seed=args.seed)
    else:
        r = stage3_refine(args.relative, args.n, args.out, args.groundtruth)
        print("closure error [R | t]:\n", r["closure"])
    if args.stage in sharded and "RANK" in os.environ:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
