"""Diagnostic: are the hybrid normals / FPFH features of a cloud the same bits while other contexts keep the chip busy?"""
import importlib, os, sys, glob, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
lib = importlib.import_module("point-cloud-registration-with-global-refinement_amd._lib")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gl = [np.load(f) for f in sorted(glob.glob(os.path.join(root, "tests", "golden", "nclt_pair_*.npz")))][:8]
reg = P.registration
stop = False
def load():
    while not stop:
        work = [(P.PointCloud(g["source"]), P.PointCloud(g["target"]), None) for g in gl]
        reg.register_pairs_plan(work, "fgr", inflight=2, fgr_voxel_size=0.1, fgr_use_absolute_scale=False, fgr_seed=2024, fgr_group=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
def feats(g, key, mode):
    pc = P.PointCloud(g[key]).voxel_down_sample(0.1)
    pc.estimate_normals(P.KDTreeSearchParamHybrid(radius=0.2, max_nn=20))
    lib.set_option("spfh_float64", mode)
    try:
        f = reg.compute_fpfh_feature(pc, P.KDTreeSearchParamHybrid(radius=1.0, max_nn=200))
    except Exception as e:
        print("mode", mode, "error:", e); f = None
    lib.set_option("spfh_float64", 0)
    return np.asarray(pc.normals).copy(), None if f is None else np.asarray(f.data).copy()
quiet = [feats(g, k, 1) for g in gl for k in ("source", "target")]
i = 0
for g in gl:
    for k in ("source", "target"):
        for mode in (0, 2):
            nrm, f = feats(g, k, mode)
            if f is not None and not np.array_equal(f, quiet[i][1]): print("QUIET: mode", mode, "differs from the float64 pass, cloud", i)
        i += 1
if len(sys.argv) > 2 and sys.argv[2] == "noload":
    t = None
else:
    t = threading.Thread(target=load); t.start()
bad = {0: 0, 1: 0, 2: 0}
for rep in range(8):
    i = 0
    for g in gl:
        for k in ("source", "target"):
            for mode in (0, 1, 2):
                nrm, f = feats(g, k, mode)
                if not np.array_equal(nrm, quiet[i][0]): print("normals differ", rep, i)
                if f is not None and not np.array_equal(f, quiet[i][1]):
                    rows = np.nonzero((f != quiet[i][1]).any(axis=0))[0]
                    bad[mode] += 1; print("features differ: mode", mode, "rep", rep, "cloud", i, len(rows), "points, first", rows[:6], "max abs", float(np.abs(f - quiet[i][1]).max()))
            i += 1
stop = True
if t: t.join()
print("done: features differ by mode", bad, "of", 8 * 16, "each")
