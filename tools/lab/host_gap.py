"""Diagnostic: how much of a bench step is spent OUTSIDE pcr_register_pairs_plan (Python packing of 48 pairs, result objects)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
reg = P.registration
lib = P._lib.load()
inner = [0.0]
orig = lib.pcr_register_pairs_plan
def timed(*a):
    t0 = time.perf_counter(); rc = orig(*a); inner[0] += time.perf_counter() - t0; return rc
lib.pcr_register_pairs_plan = timed
bases = [syn.make_pair(200000, index=i) for i in range(8)]
pairs = [syn.derive_pair(bases[k % 8], k // 8) for k in range(48)]
clouds = [(P.PointCloud(p.source), P.PointCloud(p.target)) for p in pairs]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
def step():
    batch = [(clouds[i][0], clouds[i][1], pairs[i].T_init) for i in range(48)]
    return reg.register_pairs_plan(batch, "gicp", pairs[0].voxel_sizes, pairs[0].max_distances_script, est, crit, 30, 1.0, 20, inflight=4, with_correspondences=True, group=6)
for _ in range(3): step()
torch.cuda.synchronize(); inner[0] = 0.0; t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"8 steps: {dt * 1e3 / 8:.2f} ms per step, {inner[0] * 1e3 / 8:.2f} ms inside the library call, {(dt - inner[0]) * 1e3 / 8:.2f} ms outside ({100 * (dt - inner[0]) / dt:.1f} %); {48 * 8 / dt:.1f} pairs/s")
