"""Lockstep groups against the pair-by-pair path: identical poses, counts and iteration numbers; throughput of both.
usage: group_check.py [points] [group] [inflight]"""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
reg = P.registration
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
group = int(sys.argv[2]) if len(sys.argv) > 2 else 4
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 2
base = [syn.make_pair(200000, index=i) for i in range(2)]
if npts < 200000:
    import dataclasses
    sub = np.random.default_rng(7).permutation(200000)[:npts]
    base = [dataclasses.replace(b, source=b.source[sub], target=b.target[sub]) for b in base]
pairs = [syn.derive_pair(base[k % 2], k // 2) for k in range(16)]
clouds = [(P.PointCloud(p.source), P.PointCloud(p.target)) for p in pairs]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
def run(n, g, f):
    return reg.register_pairs_plan([(clouds[i % 16][0], clouds[i % 16][1], pairs[i % 16].T_init) for i in range(n)], "gicp", pairs[0].voxel_sizes, pairs[0].max_distances_script,
                                   est, crit, inflight=f, with_correspondences=True, group=g)
a = run(16, 1, 4); b = run(16, group, inflight)
same = all(np.array_equal(x.transformation, y.transformation) and [s["iterations"] for s in x.scales] == [s["iterations"] for s in y.scales]
           and [s["n_clean"] for s in x.scales] == [s["n_clean"] for s in y.scales] and np.array_equal(x.correspondence_set, y.correspondence_set) for x, y in zip(a, b))
print("group results identical to pair-by-pair:", same)
for g, f in ((1, 4), (group, inflight), (group, 1), (2 * group, 1)):
    run(48, g, f); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): run(48, g, f)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"points {npts} group {g} x {f} in flight: {144 / dt:.1f} pairs/s")
