"""Diagnostic: stage "fgr+gicp" on the shipped NCLT scans as two passes (all FGR groups, then all GICP groups) against one library call whose
workers run FGR and GICP of a group back to back (PCR_PLAN_ONE_CALL=1)."""
import glob, importlib, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
reg = P.registration
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gold = [np.load(f) for f in sorted(glob.glob(os.path.join(root, "tests", "golden", "nclt_pair_*.npz")))]
TILES = 12
vox5 = [0.5, 0.4, 0.3, 0.2, 0.1]; dst5 = [1.5, 1.0, 0.6, 0.3, 0.1]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["PCR_PLAN_ONE_CALL"] = mode
    infl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for rep in range(2):
        work = [(P.PointCloud(g["source"]), P.PointCloud(g["target"]), None) for g in gold] * TILES
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rs = reg.register_pairs_plan(work, "fgr+gicp", vox5, dst5, est, crit, 30, 1.0, 20, inflight=infl, with_correspondences=True, fgr_voxel_size=0.1, fgr_use_absolute_scale=False, fgr_seed=2024,
                                     group=24, fgr_group=24, pair_forms=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[mode] = [r.transformation.copy() for r in rs]
    print(f"one_call={mode} inflight {infl}: {len(work) / dt:.1f} pairs/s", flush=True)
same = sum(np.array_equal(a, b) for a, b in zip(res["0"], res["1"]))
print("identical poses:", same, "of", len(res["0"]))
