"""Throughput of lockstep groups: group_sweep.py points "g,f g,f ..." """
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
reg = P.registration
npts = int(sys.argv[1]); combos = [tuple(int(v) for v in c.split(",")) for c in sys.argv[2].split()]
base = [syn.make_pair(200000, index=i) for i in range(2)]
if npts < 200000:
    import dataclasses
    sub = np.random.default_rng(7).permutation(200000)[:npts]
    base = [dataclasses.replace(b, source=b.source[sub], target=b.target[sub]) for b in base]
pairs = [syn.derive_pair(base[k % 2], k // 2) for k in range(48)]
clouds = [(P.PointCloud(p.source), P.PointCloud(p.target)) for p in pairs]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
def run(n, g, f):
    return reg.register_pairs_plan([(clouds[i % 48][0], clouds[i % 48][1], pairs[i % 48].T_init) for i in range(n)], "gicp", pairs[0].voxel_sizes, pairs[0].max_distances_script,
                                   est, crit, inflight=f, with_correspondences=False, group=g)
for g, f in combos:
    run(48, g, f); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): run(48, g, f)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"points {npts} ppl {os.environ.get('PCR_ICP_PPL', '1')} group {g} x {f} in flight: {144 / dt:.1f} pairs/s", flush=True)
