"""Diagnostic: how much of a register_pairs_plan step (48 pairs of 200k points, the bench's default) is spent outside the library call
(argument marshalling before, result objects after)."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
reg = P.registration; L = P._lib
base = [syn.make_pair(200000, index=i) for i in range(2)]
pairs = [syn.derive_pair(base[k % 2], k // 2) for k in range(48)]
clouds = [(P.PointCloud(p.source), P.PointCloud(p.target), p.T_init) for p in pairs]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
lib = L.load(); orig = lib.pcr_register_pairs_plan
inner = [0.0]
def timed(*a):
    t0 = time.perf_counter(); r = orig(*a); inner[0] += time.perf_counter() - t0; return r
lib.pcr_register_pairs_plan = timed
def run(): return reg.register_pairs_plan(clouds, "gicp", pairs[0].voxel_sizes, pairs[0].max_distances_script, est, crit, inflight=4, with_correspondences=True, group=None)
run(); torch.cuda.synchronize()
for rep in range(3):
    inner[0] = 0.0; t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {dt * 1e3:.1f} ms, inside the library call {inner[0] * 1e3:.1f} ms, outside {1e3 * (dt - inner[0]):.1f} ms")
