"""Five-scale GICP stage (script-2 parameters) on the shipped-size golden NCLT clouds from the shipped FGR poses: the 8 golden pairs tiled to
8 x tiles pairs, through register_pairs_plan with `group` pairs in lockstep and `inflight` groups in flight; pairs/s and the error band against the
shipped GICP poses.  usage: gicp_nclt_sweep.py "g1xf1,g2xf2,..." [tiles=12] [shuffle_seed]   (a seed puts the pairs in random order: groups of mixed composition)"""
import glob, importlib, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
reg = P.registration
gold = [np.load(f) for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "nclt_pair_*.npz")))]
TILES = int(sys.argv[2]) if len(sys.argv) > 2 else 12
work = [(P.PointCloud(g["source"]), P.PointCloud(g["target"]), g["T_fgr"]) for g in gold] * TILES
truth = [g["T_gicp"] for g in gold] * TILES
if len(sys.argv) > 3:
    order = np.random.default_rng(int(sys.argv[3])).permutation(len(work))
    work = [work[i] for i in order]; truth = [truth[i] for i in order]
vox5 = [0.5, 0.4, 0.3, 0.2, 0.1]; dst5 = [1.5, 1.0, 0.6, 0.3, 0.1]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
def err(T, R):
    dR = T[:3, :3].T @ R[:3, :3]
    return float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))), float(np.linalg.norm(T[:3, 3] - R[:3, 3]))
combos = [tuple(int(v) for v in c.split("x")) for c in (sys.argv[1] if len(sys.argv) > 1 else "16x4").split(",")]
for g, f in combos:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rs = reg.register_pairs_plan(work, "gicp", vox5, dst5, est, crit, 30, 1.0, 20, inflight=f, with_correspondences=False, group=g, pair_forms=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    e = [err(r.transformation, T) for r, T in zip(rs, truth)]
    good = sum(1 for a, d in e if a < 2e-3 and d < 2e-2)
    print(f"gicp group {g} x {f} in flight: {len(work) / dt:.1f} pairs/s; {good}/{len(e)} within 2e-3 rad / 2 cm of the shipped GICP pose", flush=True)
