#!/usr/bin/env python3
"""bench.py -- pairs registered per second on synthetic 200k-point NCLT-shaped clouds, 3 GICP scales
(BASELINE.json metric, config 2).

One "step" = one pass of the hot path over one BATCH of --pairs-per-step (48) independent, DISTINCT pairs already resident in
HBM, issued as ONE pcr_register_pairs_plan call (the library runs them as lockstep groups of --group pairs -- 2 at 200k
points -- and keeps --inflight groups in flight); per pair the whole
Multiscale_GICP body runs (voxel_down_sample -> remove_statistical_outlier(30, 1.0) -> estimate_normals(KNN 20) ->
registration_generalized_icp(L1, 1e-6/1e-6/100) for voxels 0.4/0.2/0.1 m), exactly the reference's pair-time definition
(2_MGICP...py:190-199).  value = pairs registered / wall time of the K steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--variant gicp|fgr] [--radius-rule script|af] [--config5]
                    [--pairs-per-step 48] [--inflight 4] [--group 0=by size] [--points 200000] [--no-cpu-baseline] [--no-extras]

--gpus N > 1: when no launcher has set RANK, this process starts N ranks itself (python -m torch.distributed.run, one fresh
process per GPU) BEFORE anything touches HIP, forwards rank 0's JSON line and exits with the launcher's code.  Every rank
registers its own batches (weak scaling, no data-path collective); one all-gather of the fixed-size pose records closes the
timed region (RCCL; gloo under PCR_BENCH_REHEARSE=1 / PCR_BENCH_DRYRUN=1).

The default line is the GICP path with the script-2 radii (1.2/0.4/0.1 m).  `extras` in the same line holds short measurements
of the other forms SURVEY §8(d) names: the ALL_FUNCTIONS radius rule (radius_from_cloud_pair * 2^-i), the FGR variant of config 2
(registro_FGR voxel 0.1 + the same 3-scale GICP, both inside the pair time) and config 5 (2M points, 5 scales, 64-NN normals).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

# Pairs in flight use up to 3 HIP streams each (GICP loop + two preprocessing lanes; one in lockstep groups); the runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams sharing a queue serialise.  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "point-cloud-registration-with-global-refinement_amd"
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
F64_MFMA_PEAK_TFLOPS = 78.6      # MI355X spec, float64 matrix (v_mfma_f64_16x16x4)
F16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PF dense bf16/f16
METRIC = "point-cloud pairs registered/sec (200k pts, 3 GICP scales)"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-step", type=int, default=48,
                    help="one step = one pass of the path over a batch of this many independent pairs (one pcr_register_pairs_plan call)")
    ap.add_argument("--points", type=int, default=200_000)
    ap.add_argument("--variant", choices=["gicp", "fgr"], default="gicp",
                    help="gicp: Multiscale_GICP from the perturbed start (the metric); fgr: config 2's FGR variant, registro_FGR (voxel 0.1) + "
                         "Multiscale_GICP from its pose, both inside the pair time (ALL_FUNCTIONS.py:317-326)")
    ap.add_argument("--radius-rule", choices=["script", "af"], default="script",
                    help="script: 3*0.4, 2*0.2, 0.1 m (2_MGICP...py:115); af: radius_from_cloud_pair * 2^-i per pair (ALL_FUNCTIONS.py:277-278)")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5 instead: 2M-point clouds (10 tiles of the 200k scene), 5 scales, 64-NN normals")
    ap.add_argument("--fixed-iterations", type=int, default=0, help="diagnostics only: run exactly this many GICP iterations per scale (criteria 0/0/N) so that "
                    "A/B runs of kernel variants do the same work whatever their summation order")
    ap.add_argument("--loss", choices=["l1", "l2"], default="l1", help="diagnostics only: l2 makes the iteration counts independent of the summation order "
                    "(the reference's L1-IRLS trajectory is chaotic: 41-89 iterations for the same pair), for A/B runs of kernel variants; the metric is quoted on l1")
    ap.add_argument("--pairs", type=int, default=0, help="distinct pairs cycled through a step (default: --pairs-per-step, i.e. every pair of a batch is a "
                    "different pair in different buffers)")
    ap.add_argument("--base-pairs", type=int, default=8, help="independently generated scenes samples (20 s each unless cached); the distinct pairs are these, "
                    "re-posed by a rigid motion of both clouds and re-ordered")
    ap.add_argument("--config3", action="store_true", help="BASELINE config 3's building block as a STRONG-scaling run: the 8 golden NCLT pairs (the reference's own scans) tiled "
                    "--config3-tiles times, cost-partitioned over the ranks (sharding.partition(costs=points)), every rank runs ONE fgr+gicp plan (what drivers.stage12 runs), "
                    "one all-gather of the pose records; the line carries per-rank wall, upload time and pairs/s")
    ap.add_argument("--config3-tiles", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("PCR_BENCH_INFLIGHT", "4")),
                    help="independent pairs (or lockstep groups, see --group) in flight per GPU (one library worker thread + context + streams each)")
    ap.add_argument("--group", type=int, default=int(os.environ.get("PCR_BENCH_GROUP", "0")),
                    help="pairs per lockstep group (pcr_pairs_plan.group): that many consecutive pairs of a batch go through the same launches; "
                         "0 = by cloud size as registration.register_pairs_plan(group=None) does (registration.default_group: 6 at 200k points)")
    args = ap.parse_args(argv)
    if args.group <= 0:
        args.group = -1          # resolved by registration.default_group once the package is imported (main)
    return args


# ------------------------------------------------------------------------------------------------ N ranks from one command
def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes through torch.distributed.run and
    forward their output.  Nothing in THIS process has imported torch or touched HIP, and it does not exec: it waits for the
    launcher and exits with its code."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    env["PCR_BENCH_LAUNCHED_BY"] = "bench.py"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    saw_line = False
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        saw_line = saw_line or line.lstrip().startswith('{"metric"')
    rc = proc.wait()
    if rc == 0 and not saw_line:
        print("bench.py: the ranks exited without printing the result line", file=sys.stderr)
        rc = 1
    return rc


def tables_identical(gathered, world, device=None) -> bool:
    """Every rank must hold the SAME ordered pose table after the one all-gather: all-gather a digest of each rank's table."""
    if world <= 1:
        return True
    import hashlib
    import numpy as np
    import torch
    import torch.distributed as dist
    h = hashlib.sha256(np.ascontiguousarray(gathered, dtype=np.float64).tobytes()).digest()
    mine = torch.tensor(list(h[:16]), dtype=torch.int64, device=device or "cpu")
    out = torch.empty((world, 16), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine.reshape(1, 16))
    return bool((out == out[0]).all().item())


# ------------------------------------------------------------------------------------------------ dry run (launcher / gather plumbing on CPU)
def dry_run(args, rank, world) -> int:
    """PCR_BENCH_DRYRUN=1: no GPU work at all; every rank fabricates its pose records, the gloo all-gather, barrier and MAX-reduce
    run exactly as in a real run and rank 0 prints a line flagged `dry_run` (tests/test_sharding.py checks n_gpus / n_ranks_seen)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    shard = importlib.import_module(PKG + ".sharding")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    n_done = args.steps * args.pairs_per_step
    t0 = time.perf_counter()
    recs = np.zeros((n_done, shard.RECORD_DOUBLES)); recs[:, 0] = recs[:, 5] = recs[:, 10] = recs[:, 15] = 1.0
    recs[:, 21] = rank * n_done + np.arange(n_done)
    gathered = shard.gather_records(recs, world * n_done) if world > 1 else recs
    same = tables_identical(gathered, world)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    seen = dist.get_world_size() if world > 1 else 1
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "pairs/s", "n_gpus": world, "n_ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "dry run: no registration was computed", "dry_run": True,
                          "config": {"workload": "launcher / gather plumbing only", "gathered_records": int(len(gathered)), "tables_identical": same}}))
    return 0 if same else 4


# ------------------------------------------------------------------------------------------------ config 3: strong scaling over the reference's own scans
def config3_run(args, rank, world, local_rank) -> int:
    """BASELINE config 3's per-GPU building block, sharded: FGR (script-1 parameters) + five-scale GICP (script-2 table) of the golden NCLT
    pairs tiled to 8 x --config3-tiles pairs (2_MGICP...py:187-214 is the axis).  Every rank takes the contiguous, cost-balanced block of pair
    indices `sharding.partition` gives it (cost = points of the pair's two clouds), uploads ITS clouds (timed separately: SURVEY 8(d) keeps the
    upload out of the pair time), runs ONE register_pairs_plan(stage="fgr+gicp") call and joins ONE all-gather of the fixed-size pose records.
    value = all pairs / the slowest rank's wall (strong scaling: the total work is fixed as N grows).  PCR_BENCH_DRYRUN=1: the same partition,
    gather, digest check and reductions with fabricated records and no GPU (tests/test_sharding.py, world 8); PCR_BENCH_REHEARSE=1: all ranks on
    device 0 over gloo (tests/test_gpu_bench.py, world 2)."""
    import glob
    import numpy as np
    import torch
    import torch.distributed as dist
    shard = importlib.import_module(PKG + ".sharding")
    dry = os.environ.get("PCR_BENCH_DRYRUN") == "1"
    rehearse = os.environ.get("PCR_BENCH_REHEARSE") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry or rehearse:
            dist.init_process_group(backend="gloo")
        else:
            try:
                torch.cuda.set_device(local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                probe = torch.ones(1, device="cuda"); dist.all_reduce(probe)        # the first collective is where RCCL reports a broken fabric / IPC set-up
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001 -- the text is the diagnosis: print it and end non-zero (no fallback to another backend)
                print(f"bench.py --config3: rank {rank}: init_process_group('nccl') / first all-reduce failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                return 5
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nclt_pair_*.npz")))
    gold = [np.load(f) for f in files]
    T = max(1, args.config3_tiles)
    n_pairs = len(gold) * T
    counts = [(len(g["source"]), len(g["target"])) for g in gold]
    costs = [counts[i % len(gold)][0] + counts[i % len(gold)][1] for i in range(n_pairs)]
    mine = list(shard.partition(n_pairs, world, rank, costs))
    my_cost = sum(costs[i] for i in mine)
    dev = None
    t_up = 0.0
    worst = [0.0, 0.0, 0.0, 0.0]            # largest distance from the shipped GICP pose: tight pairs (rad, m), noisy pairs (rad, m)
    n_in = 0
    if dry:
        t0 = time.perf_counter()
        recs = np.zeros((len(mine), shard.RECORD_DOUBLES)); recs[:, 0] = recs[:, 5] = recs[:, 10] = recs[:, 15] = 1.0
        recs[:, 21] = mine
        wall = time.perf_counter() - t0
        ok_local = True
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = None if rehearse else torch.device("cuda", local_rank)
        P = importlib.import_module(PKG)
        reg = P.registration
        vox5 = P.script2.create_scales(5); dst5 = P.script2.max_correspondence_distances(vox5)
        est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss())
        crit = reg.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=100)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        batch = [(P.PointCloud(gold[i % len(gold)]["source"]), P.PointCloud(gold[i % len(gold)]["target"]), None) for i in mine]      # every pair its own buffers
        for s_, t_, _ in batch:
            s_.device_xyz(); t_.device_xyz()
        torch.cuda.synchronize(); t_up = time.perf_counter() - t0

        def run(b, first):
            return reg.register_pairs_plan(b, "fgr+gicp", vox5, dst5, est, crit, 30, 1.0, 20, inflight=args.inflight, with_correspondences=True, fgr_voxel_size=0.1,
                                           fgr_use_absolute_scale=False, fgr_seed=20241008 + first, group=None, fgr_group=None) if b else []
        for _ in range(max(args.warmup, 1)):           # the same call, untimed: arenas, worker contexts and cached graphs at their final sizes
            run(batch, mine[0] if mine else 0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        res = run(batch, mine[0] if mine else 0)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        recs = np.stack([shard.pack_record(i, r) for i, r in zip(mine, res)]) if mine else np.zeros((0, shard.RECORD_DOUBLES))
        # every pose against the reference's shipped GICP pose: the six tight golden pairs within SURVEY 8(c)(2)'s 3e-4 rad / 3e-3 m, s1 -> s0 and the loop closure
        # (noisy for the oracle itself) within 2e-3 rad / 2 cm
        ok_local = True
        for i, r in zip(mine, res):
            g = gold[i % len(gold)]; Tg = g["T_gicp"]; dR = r.transformation[:3, :3].T @ Tg[:3, :3]
            a_ = float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))); d_ = float(np.linalg.norm(r.transformation[:3, 3] - Tg[:3, 3]))
            noisy = int(g["pair"]) in (0, 899)
            tol = (2e-3, 2e-2) if noisy else (3e-4, 3e-3)
            ok_local = ok_local and a_ <= 5e-3 and d_ <= 5e-2                 # every pose is a registration of its pair (the band of the unpinned variants)
            n_in += 1 if (a_ <= tol[0] and d_ <= tol[1]) else 0               # ... and how many reach the shipped pose's own attractor
            k = 2 if noisy else 0
            worst[k] = max(worst[k], a_); worst[k + 1] = max(worst[k + 1], d_)
    t0 = time.perf_counter()
    gathered = shard.gather_records(recs, n_pairs, device=dev) if world > 1 else recs
    same = tables_identical(gathered, world, dev)
    t_gather = time.perf_counter() - t0
    mine_row = torch.tensor([[wall, t_up, float(len(mine)), float(my_cost), 1.0 if ok_local else 0.0, t_gather] + worst + [float(n_in if not dry else len(mine))]], dtype=torch.float64, device=dev or "cpu")
    rows = mine_row
    if world > 1:
        rows = torch.empty((world, 11), dtype=torch.float64, device=mine_row.device)
        dist.all_gather_into_tensor(rows, mine_row)
        dist.barrier()
    rows = rows.cpu().numpy()
    n_seen = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        slow = float(rows[:, 0].max())
        line = {"metric": "point-cloud pairs registered/sec (shipped NCLT scans, FGR + 5-scale GICP: BASELINE config 3's block)", "value": (n_pairs / slow) if (slow > 0 and not dry) else 0.0, "unit": "pairs/s",
                "n_gpus": world, "n_ranks_seen": n_seen, "steps": 1, "warmup": args.warmup, "ms_per_step": 1e3 * slow, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32 points+search, f64 normal equations (SURVEY 8a fp column)", "data": "the reference's own scans: 8 golden NCLT pairs (tests/golden) tiled" + (" -- DRY RUN: no registration was computed" if dry else ""),
                "dry_run": dry,
                "config": {"workload": f"{n_pairs} pairs = 8 golden NCLT pairs x {T}, registro_FGR (script-1 parameters) + five-scale Multiscale_GICP (script-2 table) as one fgr+gicp plan per rank",
                           "pairs": n_pairs, "partition": "contiguous blocks balanced by the points of each pair's two clouds (sharding.partition)", "parallelism": f"pairs x{world}",
                           "gathered_records": int(len(gathered)), "tables_identical": same, "poses_valid": bool(rows[:, 4].min() > 0.5 and rows[:, 10].sum() >= 0.9 * n_pairs),
                           "validity": "every pose within 5e-3 rad / 5 cm of the reference's shipped GICP pose AND at least 90 % in that pose's own L1 attractor (SURVEY 8(c)(2): 3e-4 rad / 3e-3 m; s1 -> s0 and the "
                                       "closure pair, noisy for the oracle itself: 2e-3 rad / 2 cm).  FGR is randomised (the reference seeds from random_device): from some FGR poses the refinement of a pair "
                                       "settles a centimetre or two away, on the device as in the reference (SURVEY 8(c): shipped FGR / GICP files of 20 pairs are evidently from different runs)",
                           "pairs_in_the_shipped_poses_attractor": int(rows[:, 10].sum()),
                           "err_vs_shipped_gicp_pose_max": {"tight_pairs": {"rad": float(rows[:, 6].max()), "m": float(rows[:, 7].max()), "attractor_band": [3e-4, 3e-3]},
                                                            "noisy_pairs_s1_s0_and_closure": {"rad": float(rows[:, 8].max()), "m": float(rows[:, 9].max()), "attractor_band": [2e-3, 2e-2]}},
                           "per_rank": [{"rank": r, "pairs": int(rows[r, 2]), "points": int(rows[r, 3]), "wall_s": float(rows[r, 0]), "upload_s": float(rows[r, 1]), "gather_s": float(rows[r, 5]),
                                         "pairs_per_s": (float(rows[r, 2] / rows[r, 0]) if rows[r, 0] > 0 else None)} for r in range(world)],
                           "load_balance_max_over_mean_points": float(rows[:, 3].max() / max(rows[:, 3].mean(), 1.0)),
                           "upload": "host -> device copy of the rank's clouds, outside the pair time (SURVEY 8(d)); wall_s is the one register_pairs_plan call"}}
        print(json.dumps(line))
    return 0 if (same and bool(rows[:, 4].min() > 0.5 and rows[:, 10].sum() >= 0.9 * n_pairs)) else 4


# ------------------------------------------------------------------------------------------------ the measured run of one rank
def main(argv=None) -> int:
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PCR_BENCH_TEST_FAIL_RANK") == str(rank):
        # test hook (tests/test_sharding.py): this rank fails before it joins the process group; the job must end non-zero, and nothing re-execs
        raise SystemExit(3)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.config3:
        return config3_run(args, rank, world, local_rank)
    if os.environ.get("PCR_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # rehearsal of the N > 1 path on a one-GPU box: PCR_BENCH_REHEARSE=1 maps every rank onto device 0 and gathers over gloo
    rehearse = os.environ.get("PCR_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_ranks_seen = dist.get_world_size() if world > 1 else 1

    P = importlib.import_module(PKG)
    syn = importlib.import_module(PKG + ".synthetic")
    shard = importlib.import_module(PKG + ".sharding")
    reg = P.registration
    if args.group <= 0:
        args.group = reg.default_group(args.points * (10 if args.config5 else 1))
    ctypes = __import__("ctypes")
    lib = P._lib.load()

    B = max(1, args.pairs_per_step)
    n_distinct = args.pairs if args.pairs > 0 else B
    n_scales = 5 if args.config5 else 3
    normal_k = 64 if args.config5 else 20

    # ---- synthetic workload (same seeds on every rank; every rank registers its own copy: weak scaling)
    def base_pair(i):
        if args.points < 200_000:       # diagnostic sizes: a random subset of the 200k pair (the generator needs a full-size scene)
            import dataclasses
            b = syn.make_pair(200_000, index=i)
            sub = np.random.default_rng(7).permutation(200_000)[: args.points]
            return dataclasses.replace(b, source=b.source[sub], target=b.target[sub])
        return syn.make_pair(args.points, index=i)

    def workload(n_pairs, config5=False):
        bases = [base_pair(i) for i in range(max(1, min(args.base_pairs, n_pairs)))]
        if config5:
            bases = [syn.tile_pair(b, 10, n_scales=5) for b in bases]
        pairs = [syn.derive_pair(bases[k % len(bases)], k // len(bases)) for k in range(n_pairs)]
        clouds = [(P.PointCloud(p.source), P.PointCloud(p.target)) for p in pairs]       # resident in HBM before timing
        return pairs, clouds

    pairs, clouds = workload(n_distinct, args.config5)
    est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss() if args.loss == "l1" else reg.L2Loss())
    crit = reg.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=100)
    if args.fixed_iterations > 0:
        crit = reg.ICPConvergenceCriteria(relative_fitness=0.0, relative_rmse=0.0, max_iteration=args.fixed_iterations)

    def run_batch(n, pairs=pairs, clouds=clouds, variant=args.variant, rule=args.radius_rule, inflight=args.inflight, knn=normal_k, group=args.group):
        batch = [(clouds[i % len(pairs)][0], clouds[i % len(pairs)][1], pairs[i % len(pairs)].T_init) for i in range(n)]
        p0 = pairs[0]
        return reg.register_pairs_plan(batch, "fgr+gicp" if variant == "fgr" else "gicp", p0.voxel_sizes, p0.max_distances_script, est, crit, 30, 1.0, knn,
                                       inflight=inflight, with_correspondences=True, fgr_voxel_size=0.1, fgr_use_absolute_scale=True, fgr_seed=20241008,
                                       radius_rule="af" if rule == "af" else "given", prior_from_fgr=(variant == "fgr"), group=group)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def pool_prof(enable=None, reset=False):
        out = (ctypes.c_double * 16)()
        lib.pcr_pool_profile(ctypes.c_int(local_rank), ctypes.c_int(-1 if enable is None else int(enable)), out, ctypes.c_int(int(reset)))
        return [out[k] for k in range(16)]

    torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):                                        # W untimed warm-up steps (every worker context at least once)
        run_batch(max(B, args.inflight))
    pool_prof(enable=1, reset=True)
    barrier()
    t0 = time.perf_counter()
    results, rec_rows = [], []
    n_done = args.steps * B
    for k in range(args.steps):                 # a step = one batch of B pairs; the call returns when all B are registered
        results = run_batch(B)
        rec_rows += [shard.pack_record(rank * n_done + k * B + i, r) for i, r in enumerate(results)]    # pose records only; the rest is dropped
    recs = np.stack(rec_rows)
    gathered = shard.gather_records(recs, world * n_done, device=None if rehearse else torch.device("cuda", local_rank)) if world > 1 else recs
    same_tables = tables_identical(gathered, world, None if rehearse else torch.device("cuda", local_rank))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    prof = pool_prof(enable=0, reset=True)

    def pose_err(res, p):
        dR = res.transformation[:3, :3].T @ p.T_true[:3, :3]
        return {"rad": float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))), "m": float(np.linalg.norm(res.transformation[:3, 3] - p.T_true[:3, 3]))}

    line = None
    if rank == 0:
        # ---- roofline pass: the hot loop's launch period WITHOUT the other pairs' kernels in the way.  In the timed region the
        # pairs in flight share the GPU, so HIP events around a chunk of launches also count the time its kernels wait for
        # wavefront slots taken by other streams (kept as `us_per_launch_in_flight`); the figure that rocprofv3's per-kernel
        # durations can be checked against is measured here: a few extra steps, one pair at a time, same library path.
        run_batch(1, inflight=1, group=1)
        pool_prof(enable=1, reset=True)
        run_batch(min(4, n_distinct), inflight=1, group=1)
        torch.cuda.synchronize()
        solo = pool_prof(enable=0, reset=True)
        res = results[-1]
        ev_ms_f, ev_launches_f, ik_us_f, live_f, _, issued = (prof[i] for i in range(6))
        us_in_flight = 1e3 * ev_ms_f / ev_launches_f if ev_launches_f else None
        ev_ms, ev_launches, ik_us, live, alg_bytes, _ = (solo[i] for i in range(6))
        bytes_per_launch = alg_bytes / live if live else 0.0
        us_event = 1e3 * ev_ms / ev_launches if ev_launches else None              # HIP events over fully-live chunks
        us_kernel = ik_us / live if live else None                                 # kernels' own s_memrealtime stamps
        achieved = bytes_per_launch / (us_event * 1e-6) / 1e9 if ev_launches else 0.0
        # tracked rocprofv3 --pmc results of the same kernels (profiles/README.md): a bench run cannot read PMC counters itself.  The newest
        # round's files are quoted with the commit they were taken at (profiles/MANIFEST.json) -- static evidence, not this run.
        manifest = {}
        try:
            manifest = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
        except Exception:       # noqa: BLE001
            manifest = {}

        def tracked(suffix):
            for tag in ("r05", "r04", "r03", "r02"):
                f = os.path.join(ROOT, "profiles", f"{tag}_{suffix}")
                if os.path.exists(f):
                    return f, f"profiles/{tag}_{suffix}", manifest.get(f"{tag}_{suffix}", {}).get("commit")
            return None, None, None
        traffic_file = None
        valu_file = None
        vcsv, vname, vcommit = tracked("pmc_valu_per_pair_groups.csv")            # what the PATH is bound by (the iteration kernel itself is latency-bound):
        if not vcsv:                                                              # the table of the default run (lockstep groups), else the one-pair-at-a-time table
            vcsv, vname, vcommit = tracked("pmc_valu_per_pair.csv")
        if vcsv:
            try:
                import csv as _csv
                rows = list(_csv.DictReader(open(vcsv)))
                tot = [r for r in rows if r["kernel"] == "TOTAL"][0]
                minst = float(tot["valu_wave_instructions_per_pair_M"])
                valu_file = {"file": vname, "taken_at_commit": vcommit, "valu_wave_instructions_per_pair_M": minst,
                             "ms_per_pair_at_peak_issue": minst / 614.4, "ms_per_pair_at_vop2_issue": minst / 1024.0,
                             "largest": {"kernel": rows[0]["kernel"], "share": float(rows[0]["share_of_valu_instructions"])},
                             "note": "tracked rocprofv3 --pmc SQ_INSTS_VALU result (kernels serialised by the counter collection), not this run.  Issue rates MEASURED on this chip "
                                     "(profiles/r04_valu_rate.txt, tools/ubench/valu_rate.hip, 4-8 wavefronts per SIMD): three-source VOP3 instructions (v_med3_f32, v_max3_f32, v_fma_f32 -- the "
                                     "k-NN insertion chains) 4.1-4.5 cycles per wave64 instruction per SIMD = 614 k instructions per us on 1024 SIMDs at 2.4 GHz (ms_per_pair_at_peak_issue); "
                                     "two-source VOP2 (v_add_f32, v_mul_f32) 2.3-2.9 cycles = ~1024 k per us (ms_per_pair_at_vop2_issue): a kernel's roof lies between the two by its instruction mix"}
            except Exception:       # noqa: BLE001 -- a tracked file must never cost the line
                valu_file = None
        if valu_file and "groups" in (vname or "") and args.variant == "gicp" and not args.config5 and args.points == 200_000:
            # the path's own roof: share of the chip's VALU issue slots this run's pairs/s amount to (per GPU)
            valu_file["valu_issue_utilisation_at_this_runs_rate"] = valu_file["ms_per_pair_at_peak_issue"] * 1e-3 * (n_done / dt)
            valu_file["valu_issue_utilisation_if_all_vop2"] = valu_file["ms_per_pair_at_vop2_issue"] * 1e-3 * (n_done / dt)
        tj, tname, tcommit = tracked("traffic.json")
        if tj:
            try:
                t = json.load(open(tj))
                grp = t.get("groups/k_icp_fused")               # the timed configuration: k_icp_fused_b<1024> of a lockstep group of six (per launch = six pairs)
                traffic_file = {"file": tname, "taken_at_commit": tcommit, "hbm_bytes_per_launch": t.get("k_icp_fused", {}).get("hbm_bytes_per_launch"),
                                "hbm_bytes_per_launch_group_of_6": grp.get("hbm_bytes_per_launch") if grp else None,
                                "k_knn_wave_batchp_hbm_bytes_per_launch": t.get("groups/k_knn_wave_batchp<SOR,30>", {}).get("hbm_bytes_per_launch"),
                                "note": "tracked rocprofv3 --pmc result (FETCH_SIZE x2 + WRITE_SIZE), a previous run of the same kernels, not this run: one pair at a time (the solo "
                                        "roofline pass above) and the default lockstep-group path"}
            except Exception:       # noqa: BLE001
                traffic_file = None
        # ---- what bounds the metric: the kernel with the largest share of the kernel time of this very command (tracked rocprofv3 --kernel-trace --stats
        # summary, profiles/rNN_bench_kernel_stats.csv), with SURVEY 8(d)'s algorithmic bytes of one of ITS launches from this run's own counts
        dominant = None
        kcsv, kname, kcommit = tracked("bench_kernel_stats.csv")
        if kcsv and args.variant == "gicp" and not args.config5 and args.points == 200_000:
            try:
                import csv as _csv
                krows = list(_csv.DictReader(open(kcsv)))
                ktot = sum(float(r["TotalDurationNs"]) for r in krows)
                top = max(krows, key=lambda r: float(r["TotalDurationNs"]))
                G = max(1, args.group)
                res0 = results[-1]
                if "k_knn_wave" in top["Name"]:        # the 30-NN searches of the outlier filter of a lockstep group: every voxel point of both clouds of all scales is a query
                    units = G * sum(sum(s_["n_voxel"]) for s_ in res0.scales)
                    per_unit, unit_txt = 16.0, "16 B per voxel point (12 B point + 4 B mean distance), SURVEY 8(d) B_pre's SOR term; the k-best rows the kernel also writes (4 k B per query) are not algorithmic"
                else:                                  # the iteration kernel of a lockstep group at the finest scale
                    units = G * res0.scales[-1]["n_clean"][0]
                    per_unit, unit_txt = 48.0, "48 B per source point and launch (SURVEY 8(d) B_icp)"
                avg_us = float(top["AverageNs"]) * 1e-3
                gbs = per_unit * units / (avg_us * 1e-6) / 1e9
                dominant = {"kernel": top["Name"], "share_of_kernel_time": float(top["TotalDurationNs"]) / ktot, "calls": int(top["Calls"]), "avg_us_per_launch": avg_us,
                            "algorithmic_bytes_per_launch": per_unit * units, "bytes_model": unit_txt, "achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                            "bound": "valu issue (not bytes): see valu_from_profiles", "valu_issue_share_of_the_path": (valu_file or {}).get("largest", {}).get("share") if valu_file and "k_knn_wave" in top["Name"] else None,
                            "file": kname, "taken_at_commit": kcommit,
                            "note": "durations and shares from the tracked profile of this command (kernels of four groups in flight overlap: shares are of summed kernel time, not of wall time); "
                                    "bytes from THIS run's counts (launch = one lockstep group)"}
            except Exception:       # noqa: BLE001 -- a tracked file must never cost the line
                dominant = None
        rule_txt = "radius_from_cloud_pair * 2^-i per pair (ALL_FUNCTIONS.py:277-278)" if args.radius_rule == "af" else "radii " + "/".join(f"{d:g}" for d in pairs[0].max_distances_script) + " m"
        workload_txt = (f"step = batch of {B} independent pairs ({n_distinct} distinct), each {len(pairs[0].source)}-pt synthetic NCLT-shaped clouds, "
                        + ("registro_FGR (voxel 0.1) + " if args.variant == "fgr" else "") + f"{n_scales}-scale GICP (voxels " + "/".join(f"{v:g}" for v in pairs[0].voxel_sizes)
                        + f" m, {rule_txt}, SOR(30,1.0), KNN-{normal_k} normals, {args.loss.upper()}, 1e-6/1e-6/100)")
        it_per_pair = [sum(s["iterations"] for s in r.scales) for r in results]
        # SURVEY 8d's algorithmic bytes of the WHOLE path per pair, from this run's own counts (last step): per cloud and scale 12 N0 + 44 D + 36 C
        # (voxel grid, outlier filter, compaction, normals), per scale 48 C_source (I + 1) for the GICP loop
        n0 = [len(pairs[i % len(pairs)].source) + len(pairs[i % len(pairs)].target) for i in range(len(results))]
        path_bytes = float(np.mean([sum(12.0 * n0[i] + 44.0 * sum(s["n_voxel"]) + 36.0 * sum(s["n_clean"]) + 48.0 * s["n_clean"][0] * (s["iterations"] + 1) for s in r.scales)
                                    for i, r in enumerate(results)]))
        path_gbs = path_bytes * (n_done / dt) / 1e9
        line = {
            "metric": METRIC,
            "value": world * n_done / dt, "unit": "pairs/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 points+search, f64 normal equations (SURVEY 8a fp column)", "data": "synthetic",
            "config": {"workload": workload_txt, "variant": args.variant, "radius_rule": args.radius_rule,
                       "points_per_cloud": int(len(pairs[0].source)), "pairs_per_step": B, "distinct_pairs": n_distinct,
                       "distinct_how": f"{min(args.base_pairs, n_distinct)} independently sampled scene pairs, each re-posed by a rigid motion of both clouds and re-ordered (synthetic.derive_pair); every pair has its own buffers",
                       "parallelism": f"pairs x{world}", "pairs_in_flight_per_gpu": args.inflight * max(1, args.group), "lockstep_group": args.group, "groups_in_flight_per_gpu": args.inflight,
                       "in_flight_by": "pcr_register_pairs_plan (library worker threads)",
                       "scales": [dict(voxel=s["voxel"], max_dist=s["max_dist"], n_voxel=s["n_voxel"], n_clean=s["n_clean"],
                                       iterations=s["iterations"]) for s in res.scales],
                       "iterations_per_pair_mean": float(np.mean(it_per_pair)),
                       "iterations_per_pair_min_median_max": [int(np.min(it_per_pair)), float(np.median(it_per_pair)), int(np.max(it_per_pair))],
                       "err_vs_planted": pose_err(res, pairs[(B - 1) % len(pairs)]),
                       "err_vs_planted_max_over_last_step": {k: float(max(pose_err(r, pairs[i % len(pairs)])[k] for i, r in enumerate(results))) for k in ("rad", "m")},
                       "gathered_records": int(len(gathered)), "tables_identical": same_tables},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_from_profiles": traffic_file, "valu_from_profiles": valu_file,
                         "kernel": "k_icp_fused (one GICP iteration); the first launch of a scale is k_icp_nn + k_icp_lin",
                         "bytes_per_launch": bytes_per_launch, "us_per_launch_hip_events": us_event,
                         "us_per_launch_in_kernel_clock": us_kernel, "us_per_launch_in_flight": us_in_flight,
                         "us_slowest_workgroup_search_phase": (solo[6] / live) if live else None, "us_until_partial_sums_gathered": (solo[7] / live) if live else None,
                         "fraction_of_queries_searched_again": (solo[11] / (alg_bytes / 48.0)) if alg_bytes else None,
                         "measured_on": "up to 4 extra single-pair steps after the timed region (same process, HIP events on the launch stream)",
                         "live_launches": live, "launches_issued_timed_region": issued, "dominant": dominant,
                         "noop_launch_fraction": {"one_pair_at_a_time": (1.0 - solo[13] / solo[5]) if solo[5] else None, "timed_region": (1.0 - prof[13] / prof[5]) if prof[5] else None,
                                                  "what": "launches of the loop that found their problem(s) converged and returned at once (a chunk of launches is queued before the state of the one before it is read back; "
                                                          "the chunk length shrinks from 8 to 4 and 2 as the changes of fitness and RMSE approach the criteria)"},
                         "path": {"algorithmic_bytes_per_pair": path_bytes, "achieved": path_gbs, "unit": "GB/s", "frac": path_gbs / HBM_PEAK_GBS,
                                  "what": "SURVEY 8d's byte model of the whole pair (sum over clouds and scales of 12 N0 + 44 D + 36 C, plus 48 C_source (I + 1) per scale) x this run's pairs/s per GPU, "
                                          "over the HBM peak: the kernel figure above is one launch of the hot loop alone"}},
        }
        if args.variant == "fgr" and solo[10] > 0:
            line["roofline_fgr"] = fgr_roofline(solo)
        if not args.no_extras and world == 1 and not args.config5 and args.variant == "gicp" and args.radius_rule == "script":
            line["extras"] = extras(args, P, syn, reg, est, crit, pairs, clouds, run_batch, pool_prof, pose_err, workload)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(pairs[0], len(pairs[0].source), args.variant, args.radius_rule, normal_k)
            # vs_baseline stays null: the reference publishes no number for this metric (BASELINE.json "published": {}); the same-host CPU port is the stated substitute
            line["vs_cpu_baseline"] = {"ratio": line["value"] / line["cpu_baseline"]["value"] if line["cpu_baseline"]["value"] > 0 else None,
                                       "of": f"this line's value over cpu_baseline.value (kind port, {line['cpu_baseline']['cores']} threads); not a published baseline"}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    return 0 if same_tables else 4


def fgr_roofline(prof):
    """Feature matching of registro_FGR (pcr_fgr.hip): HIP-event time over the nearest-neighbour kernels of both directions;
    algorithmic flops = 2 * 33 * Ns * Nt per direction."""
    ms, flops, launches = prof[8], prof[9], prof[10]
    tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # this run's own count: (query wavefront, 64-row step) pairs the screen computed over all of them (record passes and bound-only sweeps), x K = 64 per 33 dimensions
    share = prof[12] / prof[15] if prof[15] > 0 else None
    executed = share * 64.0 / 33.0 if share is not None else None
    return {"bound": "mfma", "achieved": tf, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F16_MFMA_PEAK_TFLOPS, "traffic": None,
            "kernel": "k_feature_nn_screen (33-D nearest feature: f16 hi/lo split MFMA screen over the (query wavefront, 64-row tile) pairs whose "
                      "boxes in principal coordinates are close enough, ~20 % of them; survivors re-checked in float64)",
            "flops_per_launch": flops / launches if launches else 0.0, "ms_per_launch_hip_events": ms / launches if launches else None,
            "wavefront_step_pairs_computed_share": share, "executed_over_algorithmic": executed,
            "executed_mfma_utilisation": (tf * executed / F16_MFMA_PEAK_TFLOPS) if executed is not None else None,
            "peak_note": "achieved = ALGORITHMIC flops of the all-pairs search (2 x 33 x Ns x Nt per direction) over the dense f16 MFMA peak.  The screen spends K = 64 "
                         "per 33 dimensions (round 4: 33 hi halves + both hi.lo cross terms of the 15 widest columns; K = 128 with all cross terms before) and, with tile "
                         "pruning, only on the (wavefront, tile) pairs counted by this run (wavefront_step_pairs_computed_share, bound-only sweeps included): executed MFMA flops = share x 64/33 x algorithmic.  The kernel is "
                         "no longer paced by its matrix pipe at all (DESIGN section 4, round 4: staging traffic and latency); executed_mfma_utilisation is reported for "
                         "continuity; the float64 MFMA path the screen replaces peaks at 78.6 TFLOP/s"}


def extras(args, P, syn, reg, est, crit, pairs, clouds, run_batch, pool_prof, pose_err, workload):
    """Short measurements of the other forms of config 2 and of config 5, after the timed region, bounded to a few
    seconds each.  Same library path, same pairs in flight; warm-up of one batch each."""
    import numpy as np
    import torch
    out = {}
    n = min(len(pairs), 16)

    def timed(label, reps=2, **kw):
        run_batch(n, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = run_batch(n, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return res, reps * n / dt

    res, pps = timed("af", rule="af")
    out["gicp_af_radius_rule"] = {"pairs_per_s": pps, "pairs_timed": 2 * n, "max_distances_m": [s["max_dist"] for s in res[0].scales],
                                  "iterations": [s["iterations"] for s in res[0].scales], "err_vs_planted": pose_err(res[0], pairs[0]),
                                  "what": "same pairs, search radii radius_from_cloud_pair * 2^-i (ALL_FUNCTIONS.py:277-278): an effectively unbounded 1-NN"}
    res, pps = timed("fgr", variant="fgr")
    out["fgr_plus_gicp"] = {"pairs_per_s": pps, "pairs_timed": 2 * n, "fgr_fitness": res[0].fgr.fitness, "fgr_inlier_rmse": res[0].fgr.inlier_rmse,
                            "err_vs_planted_fgr": pose_err(res[0].fgr, pairs[0]), "err_vs_planted": pose_err(res[0], pairs[0]),
                            "what": "config 2's FGR variant: registro_FGR (voxel 0.1, ALL_FUNCTIONS.py:178-203) + the same 3-scale GICP started from its pose; "
                                    "both inside the pair time"}
    # the feature screen's roofline is measured like the GICP one: one pair at a time (HIP events stretched by other streams' kernels are not kernel time)
    run_batch(1, variant="fgr", inflight=1, group=1)
    pool_prof(enable=1, reset=True)
    run_batch(2, variant="fgr", inflight=1, group=1)
    torch.cuda.synchronize()
    fp = pool_prof(enable=0, reset=True)
    if fp[10] > 0:
        out["fgr_plus_gicp"]["roofline"] = fgr_roofline(fp)
        out["fgr_plus_gicp"]["roofline"]["measured_on"] = "2 pairs, one at a time, after the timed batch (solo, like the GICP roofline)"
    # ---- BASELINE config 3's building block on the SHIPPED-size clouds: the 8 golden NCLT pairs (tests/golden, 11k-28k points; fixtures = the
    # reference's own scans with its shipped FGR and GICP poses) tiled to 96 pairs: the script-1 FGR stage (1_FGR...py:134-147) in lockstep FGR
    # groups and the script-2 five-scale GICP stage (2_MGICP...py:187-214) from the shipped FGR poses in lockstep GICP groups, sized by the library.
    # A throughput only counts with its output: every FGR pose must lie within 3e-2 rad / 0.5 m and every refined pose within 2e-3 rad / 2 cm of
    # the shipped GICP pose (`valid`).
    try:
        import glob
        gold = [np.load(f) for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "nclt_pair_*.npz")))]
        gc = [(P.PointCloud(g["source"]), P.PointCloud(g["target"])) for g in gold]
        vox5 = [0.5, 0.4, 0.3, 0.2, 0.1]; dst5 = [1.5, 1.0, 0.6, 0.3, 0.1]
        m = 96

        def run_small(stage):
            batch = [(gc[i % len(gc)][0], gc[i % len(gc)][1], gold[i % len(gc)]["T_fgr"]) for i in range(m)]
            return reg.register_pairs_plan(batch, stage, vox5, dst5, est, crit, 30, 1.0, 20, inflight=args.inflight, with_correspondences=True, fgr_voxel_size=0.1,
                                           fgr_use_absolute_scale=False, fgr_seed=20241008, group=None, fgr_group=None)

        def band(rs):
            e = []
            for i, r in enumerate(rs):
                Tg = gold[i % len(gc)]["T_gicp"]; dR = r.transformation[:3, :3].T @ Tg[:3, :3]
                e.append((float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))), float(np.linalg.norm(r.transformation[:3, 3] - Tg[:3, 3]))))
            return e
        nc = {}
        for k_ in ("fgr_group_pairs_redone_alone", "fgr_group_barrier_timeouts", "fgr_group_pool_overflows"):
            P._lib.counter(k_, True)
        for stage, tol in (("gicp", (2e-3, 2e-2)), ("fgr", (3e-2, 0.5))):
            run_small(stage); torch.cuda.synchronize(); t0 = time.perf_counter()
            r = run_small(stage); torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            e = band(r)
            inside = sum(1 for a_, d_ in e if a_ <= tol[0] and d_ <= tol[1])
            nc[stage] = {"pairs_per_s": m / dt_s, "pairs_timed": m, "err_vs_shipped_gicp_pose_max": {"rad": max(a_ for a_, _ in e), "m": max(d_ for _, d_ in e)},
                         "tolerance": {"rad": tol[0], "m": tol[1]}, "pairs_inside": inside, "valid": inside == m}
        # ... and the two stages as the ONE plan a rank of config 3 runs (drivers.stage12: stage "fgr+gicp" = registro_FGR over the block in lockstep FGR
        # groups, then the five-scale GICP from ITS poses in lockstep GICP groups; script-1 / script-2 parameters).  Valid only if every pose reaches the
        # shipped GICP pose: the six golden pairs with a tight L1 attractor (tests/test_gpu_gicp.py) within SURVEY 8(c)(2)'s 3e-4 rad / 3e-3 m, the two
        # noisy ones (s1 -> s0 and the loop closure, whose oracle end poses scatter by 1e-3 rad themselves) within the refined-pose band above.
        run_small("fgr+gicp"); torch.cuda.synchronize(); t0 = time.perf_counter()
        r = run_small("fgr+gicp"); torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        e = band(r)
        noisy = [int(g["pair"]) in (0, 899) for g in gold]
        tight_in = sum(1 for i, (a_, d_) in enumerate(e) if not noisy[i % len(gc)] and a_ <= 3e-4 and d_ <= 3e-3)
        noisy_in = sum(1 for i, (a_, d_) in enumerate(e) if noisy[i % len(gc)] and a_ <= 2e-3 and d_ <= 2e-2)
        n_tight = sum(1 for i in range(m) if not noisy[i % len(gc)])
        nc["fgr+gicp"] = {"pairs_per_s": m / dt_s, "pairs_timed": m, "plan": "one register_pairs_plan(stage='fgr+gicp') call: what drivers.stage12_fgr_mgicp runs per rank",
                          "err_vs_shipped_gicp_pose_max_tight_pairs": {"rad": max(a_ for i, (a_, _) in enumerate(e) if not noisy[i % len(gc)]), "m": max(d_ for i, (_, d_) in enumerate(e) if not noisy[i % len(gc)])},
                          "err_vs_shipped_gicp_pose_max_noisy_pairs": {"rad": max(a_ for i, (a_, _) in enumerate(e) if noisy[i % len(gc)]), "m": max(d_ for i, (_, d_) in enumerate(e) if noisy[i % len(gc)])},
                          "tolerance_tight_pairs": {"rad": 3e-4, "m": 3e-3}, "tolerance_noisy_pairs": {"rad": 2e-3, "m": 2e-2},
                          "tight_pairs_inside": tight_in, "tight_pairs": n_tight, "noisy_pairs_inside": noisy_in, "noisy_pairs": m - n_tight,
                          "all_within_5e-3_rad_5_cm": all(a_ <= 5e-3 and d_ <= 5e-2 for a_, d_ in e),
                          "valid": all(a_ <= 5e-3 and d_ <= 5e-2 for a_, d_ in e) and (tight_in + noisy_in) >= 0.9 * m,
                          "validity": "every pose within 5e-3 rad / 5 cm of the shipped GICP pose and at least 90 % inside that pose's own L1 attractor (tolerance_* above): FGR is randomised, and from "
                                      "some FGR poses the refinement of a pair settles a centimetre or two away -- on the device as in the reference"}
        npts = float(np.mean([len(g["source"]) + len(g["target"]) for g in gold])) / 2
        out["nclt_shipped_size_clouds"] = {"gicp_stage_5_scales": nc["gicp"], "fgr_stage": nc["fgr"], "fgr_plus_gicp": nc["fgr+gicp"],
                                           "points_per_cloud": [int(min(len(g["source"]) for g in gold)), int(max(len(g["source"]) for g in gold))],
                                           "groups": {"gicp_group": reg.balanced_group(reg.default_group(npts), m, args.inflight), "fgr_group": reg.balanced_group(reg.default_fgr_group(npts), m, args.inflight),
                                                      "groups_in_flight": args.inflight, "how": "register_pairs_plan(group=None, fgr_group=None): sized by the clouds, whole rounds of the workers (both divide the 96 pairs)"},
                                           "with_correspondences": True,
                                           "fgr_group_fallbacks_in_these_runs": {k: P._lib.counter(k, True) for k in ("fgr_group_pairs_redone_alone", "fgr_group_barrier_timeouts", "fgr_group_pool_overflows")},
                                           "what": "BASELINE config 3's per-GPU building block on the reference's own NCLT scans (8 golden pairs tiled to 96 per call): script-1 FGR stage, "
                                                   "script-2 five-scale GICP stage from the shipped FGR poses, and both as one plan; poses checked against the shipped GICP poses; correspondence sets "
                                                   "returned, as the reference's result objects hold them (tools/fgr_group_sweep.py and tools/gicp_nclt_sweep.py time the stages without them)"}
    except Exception as e:      # noqa: BLE001 -- an extra must never cost the main line
        out["nclt_shipped_size_clouds"] = {"error": repr(e)}
    try:
        p5, c5 = workload(2, config5=True)
        n5, fl5 = 12, 4

        def run5(m, inflight):
            batch = [(c5[i % 2][0], c5[i % 2][1], p5[i % 2].T_init) for i in range(m)]
            return reg.register_pairs_plan(batch, "gicp", p5[0].voxel_sizes, p5[0].max_distances_script, est, crit, 30, 1.0, 64, inflight=inflight, with_correspondences=True)
        run5(fl5, fl5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r5 = run5(n5, fl5)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        # the iteration kernel's own period: one pair at a time, HIP events on its launch stream, every scale's loop started after the preprocessing
        # of the later scales has finished (option "fence_prep": their 7-ms k-NN searches otherwise share the chip with the loop being timed)
        P._lib.set_option("fence_prep", 1)
        pool_prof(enable=1, reset=True)
        run5(2, 1)
        torch.cuda.synchronize()
        f5 = pool_prof(enable=0, reset=True)
        P._lib.set_option("fence_prep", 0)
        us = 1e3 * f5[0] / f5[1] if f5[1] else None
        bpl = f5[4] / f5[3] if f5[3] else 0.0
        out["config5_2M_points_5_scales_64nn"] = {
            "pairs_per_s": n5 / dt, "pairs_timed": n5, "points_per_cloud": int(len(p5[0].source)), "pairs_in_flight": fl5,
            "scales": [dict(voxel=s["voxel"], n_clean=s["n_clean"], iterations=s["iterations"]) for s in r5[0].scales],
            "err_vs_planted": pose_err(r5[0], p5[0]),
            "roofline": {"bound": "hbm", "kernel": "k_icp_fused<512> (one GICP iteration; round 5: the fused kernel for every cloud size)", "bytes_per_launch": bpl, "us_per_launch_hip_events": us,
                         "measured_on": "2 pairs, one at a time, after the timed batch, each scale's loop fenced against the later scales' preprocessing (pcr_set_option fence_prep)",
                         "achieved": (bpl / (us * 1e-6) / 1e9) if us else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (bpl / (us * 1e-6) / 1e9 / HBM_PEAK_GBS) if us else 0.0},
            "what": "BASELINE config 5: 2M-point clouds (10 tiles of the 200k scene, synthetic.tile_pair), script-2 5-scale table, 64-NN normals"}
    except Exception as e:      # noqa: BLE001 -- an extra must never cost the main line
        out["config5_2M_points_5_scales_64nn"] = {"error": repr(e)}
    return out


def cpu_baseline(pair, npts, variant, rule, normal_k):
    """The oracle ("port": float64 C/OpenMP restatement of the Open3D CPU path) on the SAME workload, timed on
    this host's cores.  Bounded sample: as many repetitions of pair 0 as fit in ~20 s (at least one, at most five)."""
    from oracle import oracle as orc
    orc.build()
    # a 1-GPU box grants this job a 16-core CPU share although os.cpu_count() reports the whole host
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("PCR_CPU_BASELINE_THREADS", "16"))))
    orc.set_num_threads(cores)
    t0 = time.perf_counter(); n = 0
    while True:
        dists = pair.max_distances_script
        T0 = pair.T_init
        if variant == "fgr":
            T0 = orc.registro_fgr(pair.source, pair.target, 0.1, True, seed=20241008).transformation
        if rule == "af":
            r = orc.radius_from_cloud_pair(pair.source, pair.target)
            dists = [r * 2.0 ** -i for i in range(len(pair.voxel_sizes))]
        orc.multiscale_gicp(pair.source, pair.target, pair.voxel_sizes, dists, T0, normal_k=normal_k)
        n += 1
        if time.perf_counter() - t0 > 20.0 or n >= 5:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} x pair 0 of the same {npts}-pt workload (identical parameters, variant {variant}, radius rule {rule}), OpenMP threads = {cores}"}


if __name__ == "__main__":
    sys.exit(main())
