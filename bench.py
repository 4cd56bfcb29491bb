#!/usr/bin/env python3
"""bench.py -- pairs registered per second on synthetic 200k-point NCLT-shaped clouds, 3 GICP scales
(BASELINE.json metric, config 2).

One "step" = one pass of the hot path over one BATCH of --pairs-per-step (48) independent pairs already resident in HBM,
issued as ONE pcr_register_pairs call (the library keeps --inflight pairs in flight); per pair the whole Multiscale_GICP
body runs (voxel_down_sample -> remove_statistical_outlier(30, 1.0) -> estimate_normals(KNN 20) ->
registration_generalized_icp(L1, 1e-6/1e-6/100) for voxels 0.4/0.2/0.1 m, search radii 1.2/0.4/0.1 m), exactly the
reference's pair-time definition (2_MGICP...py:190-199).  value = pairs registered / wall time of the K steps.
N>1: one process per GPU, each registers its own batches (no data-path collective); one all-gather of the fixed-size
pose records closes the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs-per-step 48] [--inflight 3] [--points 200000] [--no-cpu-baseline]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

# Pairs in flight use 3 HIP streams each (GICP loop + two preprocessing lanes); the runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams sharing a queue serialise.  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "point-cloud-registration-with-global-refinement_amd"
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-step", type=int, default=48,
                    help="one step = one pass of the path over a batch of this many independent pairs (one pcr_register_pairs call)")
    ap.add_argument("--points", type=int, default=200_000)
    ap.add_argument("--fixed-iterations", type=int, default=0, help="diagnostics only: run exactly this many GICP iterations per scale (criteria 0/0/N) so that "
                    "A/B runs of kernel variants do the same work whatever their summation order")
    ap.add_argument("--loss", choices=["l1", "l2"], default="l1", help="diagnostics only: l2 makes the iteration counts independent of the summation order "
                    "(the reference's L1-IRLS trajectory is chaotic: 41-89 iterations for the same pair), for A/B runs of kernel variants; the metric is quoted on l1")
    ap.add_argument("--pairs", type=int, default=2, help="distinct synthetic pairs cycled through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("PCR_BENCH_INFLIGHT", "4")),
                    help="independent pairs in flight per GPU (one host thread + one HIP stream + one library context each)")
    ap.add_argument("--python-threads", action="store_true", help="keep the pairs in flight with host threads in Python (one pcr_multiscale_gicp call per pair) "
                    "instead of ONE pcr_register_pairs call for the timed steps (default: the library keeps them in flight)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    P = importlib.import_module(PKG)
    syn = importlib.import_module(PKG + ".synthetic")
    shard = importlib.import_module(PKG + ".sharding")
    ctypes = __import__("ctypes")

    # ---- synthetic workload (same seeds on every rank; every rank registers its own copy: weak scaling)
    if args.points < 200_000:       # diagnostic sizes: a random subset of the 200k pair (the generator needs a full-size scene)
        import dataclasses
        base = [syn.make_pair(200_000, index=i) for i in range(args.pairs)]
        sub = np.random.default_rng(7).permutation(200_000)[: args.points]
        pairs = [dataclasses.replace(b, source=b.source[sub], target=b.target[sub]) for b in base]
    else:
        pairs = [syn.make_pair(args.points, index=i) for i in range(args.pairs)]
    clouds = [(P.PointCloud(p.source), P.PointCloud(p.target)) for p in pairs]       # resident in HBM before timing
    est = P.registration.TransformationEstimationForGeneralizedICP(P.registration.L1Loss() if args.loss == "l1" else P.registration.L2Loss())
    crit = P.registration.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=100)
    if args.fixed_iterations > 0:
        crit = P.registration.ICPConvergenceCriteria(relative_fitness=0.0, relative_rmse=0.0, max_iteration=args.fixed_iterations)

    def step(i):
        p = pairs[i % len(pairs)]; s, t = clouds[i % len(pairs)]
        return P.registration.multiscale_gicp(s, t, p.voxel_sizes, p.max_distances_script, p.T_init, est, crit, 30, 1.0, 20)

    # ---- executor.  Default: the timed steps are ONE pcr_register_pairs call, the library keeps `inflight` pairs in flight
    # (worker threads, contexts and streams of its own).  --python-threads: `inflight` host threads in Python, each with
    # its own HIP stream and library context, one pcr_multiscale_gicp call per pair (9 % slower: interpreter overhead).
    args.batch_api = not args.python_threads
    B = max(1, args.pairs_per_step)
    import threading
    from concurrent.futures import ThreadPoolExecutor
    ctxs = []
    tls = threading.local()

    def worker_init():
        torch.cuda.set_device(local_rank)
        tls.stream = torch.cuda.Stream(priority=int(os.environ.get("PCR_BENCH_STREAM_PRIO", "0")))
        with torch.cuda.stream(tls.stream):
            c = P._lib.Context.current()
        ctxs.append(c)

    def run_step(i):
        with torch.cuda.stream(tls.stream):
            return step(i)

    pool = None
    torch.cuda.synchronize()
    if not args.batch_api:
        pool = ThreadPoolExecutor(max_workers=args.inflight, initializer=worker_init)
        for _ in range(max(args.warmup, 1)):                                    # warm-up steps (every worker at least once)
            list(pool.map(run_step, range(max(B, args.inflight))))
    prof = (ctypes.c_double * 8)()
    for c in ctxs:
        c.lib.pcr_profile_enable(c.handle, 1)
        c.lib.pcr_profile_read(c.handle, prof, 1)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_batch(n):
        batch = [(clouds[i % len(pairs)][0], clouds[i % len(pairs)][1], pairs[i % len(pairs)].T_init) for i in range(n)]
        p0 = pairs[0]
        return P.registration.register_pairs(batch, p0.voxel_sizes, p0.max_distances_script, est, crit, 30, 1.0, 20, inflight=args.inflight)

    lib = P._lib.load()
    if args.batch_api:
        for _ in range(max(args.warmup, 1)):
            run_batch(max(B, args.inflight))
        lib.pcr_pool_profile(ctypes.c_int(local_rank), ctypes.c_int(1), None, ctypes.c_int(1))
    barrier()
    t0 = time.perf_counter()
    results, rec_rows = [], []
    n_done = args.steps * B
    for k in range(args.steps):                 # a step = one batch of B pairs; the call returns when all B are registered
        results = run_batch(B) if args.batch_api else list(pool.map(run_step, range(B)))
        rec_rows += [shard.pack_record(rank * n_done + k * B + i, r) for i, r in enumerate(results)]    # pose records only; the rest is dropped
    recs = np.stack(rec_rows)
    gathered = shard.gather_records(recs, world * n_done, device=None if rehearse else torch.device("cuda", local_rank)) if world > 1 else recs
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    acc = [0.0] * 8
    if args.batch_api:
        lib.pcr_pool_profile(ctypes.c_int(local_rank), ctypes.c_int(0), prof, ctypes.c_int(1))
        acc = [prof[k] for k in range(8)]
    for c in ([] if args.batch_api else ctxs):
        c.lib.pcr_profile_read(c.handle, prof, 0)
        c.lib.pcr_profile_enable(c.handle, 0)
        for k in range(8):
            acc[k] += prof[k]
    prof = acc

    # ---- roofline pass: the hot loop's launch period WITHOUT the other pairs' kernels in the way.  In the timed region
    # three pairs share the GPU, so HIP events around a chunk of launches also count the time its kernels wait for
    # wavefront slots taken by other streams (kept as `us_per_launch_in_flight`); the figure that rocprofv3's per-kernel
    # durations can be checked against is measured here: a few extra steps, one pair at a time, same streams and contexts.
    solo = [0.0] * 8
    if rank == 0:
        def solo_step(i):
            with torch.cuda.stream(solo_stream):
                return step(i)
        solo_stream = torch.cuda.Stream()
        with torch.cuda.stream(solo_stream):
            cs = P._lib.Context.current()
        solo_step(0)
        cs.lib.pcr_profile_enable(cs.handle, 1); cs.lib.pcr_profile_read(cs.handle, prof_buf := (ctypes.c_double * 8)(), 1)
        for i in range(4):
            solo_step(i)
        torch.cuda.synchronize()
        cs.lib.pcr_profile_read(cs.handle, prof_buf, 0); cs.lib.pcr_profile_enable(cs.handle, 0)
        solo = [prof_buf[k] for k in range(8)]

    if rank == 0:
        res = results[-1]
        # ---- roofline of the hot loop: ONE GICP iteration = k_icp_nn + k_icp_iter (SURVEY K10+K11), 48 B per source point
        ev_ms_f, ev_launches_f, ik_us_f, live_f, _, issued = (prof[i] for i in range(6))
        us_in_flight = 1e3 * ev_ms_f / ev_launches_f if ev_launches_f else None     # None: the pairs in flight share their launches (IcpEngine)
        ev_ms, ev_launches, ik_us, live, alg_bytes, _ = (solo[i] for i in range(6))
        bytes_per_launch = alg_bytes / live if live else 0.0
        us_event = 1e3 * ev_ms / ev_launches if ev_launches else None              # HIP events over fully-live chunks
        us_kernel = ik_us / live if live else None                                 # kernels' own s_memrealtime stamps
        achieved = bytes_per_launch / (us_event * 1e-6) / 1e9 if ev_launches else 0.0
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")                    # rocprofv3 --pmc passes (profiles/README.md)
        if os.path.exists(tj):
            t = json.load(open(tj))
            traffic = t.get("k_icp_fused", {}).get("hbm_bytes_per_launch") or None
        # sanity of the result itself (planted motion) -- printed, not part of the contract
        p_last = pairs[(B - 1) % len(pairs)]
        dR = res.transformation[:3, :3].T @ p_last.T_true[:3, :3]
        ang = float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))); dtr = float(np.linalg.norm(res.transformation[:3, 3] - p_last.T_true[:3, 3]))
        line = {
            "metric": "point-cloud pairs registered/sec (200k pts, 3 GICP scales)",
            "value": world * n_done / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 points+search, f64 normal equations", "data": "synthetic",
            "config": {"workload": f"step = batch of {B} independent pairs, each {args.points}-pt synthetic NCLT-shaped clouds, 3-scale GICP "
                                   f"(voxels 0.4/0.2/0.1 m, radii 1.2/0.4/0.1 m, SOR(30,1.0), KNN-20 normals, {args.loss.upper()}, 1e-6/1e-6/100)",
                       "points_per_cloud": args.points, "pairs_per_step": B, "distinct_pairs": args.pairs, "parallelism": f"pairs x{world}", "pairs_in_flight_per_gpu": args.inflight,
                       "in_flight_by": "pcr_register_pairs (library worker threads)" if args.batch_api else "python host threads",
                       "scales": [dict(voxel=s["voxel"], max_dist=s["max_dist"], n_voxel=s["n_voxel"], n_clean=s["n_clean"],
                                       iterations=s["iterations"]) for s in res.scales],
                       "err_vs_planted": {"rad": ang, "m": dtr}, "gathered_records": int(len(gathered))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_icp_fused (one GICP iteration; the first launch of a scale is k_icp_nn + k_icp_iter)",
                         "bytes_per_launch": bytes_per_launch, "us_per_launch_hip_events": us_event,
                         "us_per_launch_in_kernel_clock": us_kernel, "us_per_launch_in_flight": us_in_flight,
                         "measured_on": "4 extra single-pair steps after the timed region (same process, HIP events on the launch stream)",
                         "live_launches": live, "launches_issued_timed_region": issued},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(pairs[0], args.points)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(pair, npts):
    """The oracle ("port": float64 C/OpenMP restatement of the Open3D CPU path) on the SAME workload, timed on
    this host's cores.  Bounded sample: as many repetitions of pair 0 as fit in ~20 s (at least one)."""
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
        orc.multiscale_gicp(pair.source, pair.target, pair.voxel_sizes, pair.max_distances_script, pair.T_init)
        n += 1
        if time.perf_counter() - t0 > 20.0 or n >= 5:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} x pair 0 of the same {npts}-pt workload (identical parameters), OpenMP threads = {cores}"}


if __name__ == "__main__":
    main()
