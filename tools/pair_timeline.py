"""Host-clock breakdown of a pair in flight (PCR_PAIR_TIMELINE=1): enqueue of the preprocessing, waiting for it, the GICP loops."""
import ctypes, importlib, os, sys, time
os.environ["PCR_PAIR_TIMELINE"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
reg = P.registration
inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 4
base = [syn.make_pair(200000, index=i) for i in range(2)]
pairs = [syn.derive_pair(base[k % 2], k // 2) for k in range(48)]
clouds = [(P.PointCloud(p.source), P.PointCloud(p.target)) for p in pairs]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
lib = P._lib.load()
def run(n):
    return reg.register_pairs_plan([(clouds[i][0], clouds[i][1], pairs[i].T_init) for i in range(n)], "gicp", pairs[0].voxel_sizes, pairs[0].max_distances_script, est, crit, inflight=inflight, with_correspondences=False)
run(48)
out = (ctypes.c_double * 16)()
lib.pcr_pool_profile(0, 1, out, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): run(48)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
lib.pcr_pool_profile(0, 0, out, 1)
n = out[15]
print(f"inflight {inflight}: {144 / dt:.1f} pairs/s, wall per pair {dt / 144 * 1e3:.2f} ms; per pair (ms): enqueue prep {out[11] / n * 1e3:.2f}, wait prep {out[12] / n * 1e3:.2f}, GICP loops {out[13] / n * 1e3:.2f}; "
      f"ICP launches per pair {out[5] / n:.0f} (live {out[3] / n:.0f}), ICP event period {out[0] / out[1] * 1e3:.1f} us")
