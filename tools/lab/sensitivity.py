"""Where the in-flight throughput goes (sensitivity only: the cheaper variants compute something else): the default batch of 48
distinct 200k-point pairs in lockstep groups, with the search sizes and the iteration counts cut one at a time.
usage: sensitivity.py [points] [group] [inflight]"""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
reg = P.registration
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = int(sys.argv[3]) if len(sys.argv) > 3 else 4
base = [syn.make_pair(200000, index=i) for i in range(2)]
if npts < 200000:
    import dataclasses
    sub = np.random.default_rng(7).permutation(200000)[:npts]
    base = [dataclasses.replace(b, source=b.source[sub], target=b.target[sub]) for b in base]
pairs = [syn.derive_pair(base[k % 2], k // 2) for k in range(48)]
clouds = [(P.PointCloud(p.source), P.PointCloud(p.target)) for p in pairs]
est = reg.TransformationEstimationForGeneralizedICP(reg.L1Loss())
def run(n, sor_k, nk, crit):
    return reg.register_pairs_plan([(clouds[i % 48][0], clouds[i % 48][1], pairs[i % 48].T_init) for i in range(n)], "gicp", pairs[0].voxel_sizes, pairs[0].max_distances_script,
                                   est, crit, sor_k, 1.0, nk, inflight=f, with_correspondences=False, group=g)
full = reg.ICPConvergenceCriteria(1e-6, 1e-6, 100)
fixed = lambda n: reg.ICPConvergenceCriteria(0.0, 0.0, n)
for name, sor_k, nk, crit in (("reference parameters", 30, 20, full), ("25 iterations per scale", 30, 20, fixed(25)), ("1 iteration per scale (preprocessing only)", 30, 20, fixed(1)),
                              ("SOR k 8, normals k 8, 25 iterations", 8, 8, fixed(25)), ("SOR k 8, normals k 8, 1 iteration", 8, 8, fixed(1)),
                              ("SOR k 30, normals k 8, 25 iterations", 30, 8, fixed(25)), ("50 iterations per scale", 30, 20, fixed(50))):
    run(48, sor_k, nk, crit); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): run(48, sor_k, nk, crit)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"points {npts} group {g} x {f}: {name}: {144 / dt:.1f} pairs/s = {dt / 144 * 1e3:.2f} ms per pair", flush=True)
