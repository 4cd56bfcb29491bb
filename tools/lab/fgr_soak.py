"""Diagnostic: the FGR stage on the shipped-size golden NCLT clouds (96 pairs per call, default lockstep groups) many times over: every
repetition must return the SAME poses bit for bit (the record pool of the feature screen fills in a different order every time; the exact
re-check and the fixed summation orders make the result independent of it), free device memory must not shrink.
usage: fgr_soak.py [repetitions=20] [inflight=4]"""
import glob, importlib, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
reg = P.registration
gold = [np.load(f) for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "nclt_pair_*.npz")))]
work = [(P.PointCloud(g["source"]), P.PointCloud(g["target"]), None) for g in gold] * 12
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 4
def run():
    return reg.register_pairs_plan(work, "fgr", inflight=inflight, with_correspondences=True, fgr_voxel_size=0.1, fgr_use_absolute_scale=False, fgr_seed=5)
ref = run(); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
bits = [(np.asarray(r.transformation).tobytes(), r.fitness, np.asarray(r.correspondence_set).tobytes()) for r in ref]
t0 = time.perf_counter(); bad = 0
for k in range(reps):
    rs = run()
    for i, r in enumerate(rs):
        if (np.asarray(r.transformation).tobytes(), r.fitness, np.asarray(r.correspondence_set).tobytes()) != bits[i]:
            bad += 1
            if bad < 5: print(f"repetition {k} pair {i}: result differs from the first run", flush=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
free1 = torch.cuda.mem_get_info()[0]
print(f"fgr soak: {reps} repetitions x {len(work)} pairs, {reps * len(work) / dt:.0f} pairs/s, {bad} results differing, free memory {free0 >> 20} -> {free1 >> 20} MiB")
sys.exit(1 if bad else 0)
