"""The exact 30-NN search with the chip saturated: 8 shifted copies of the 0.1 m voxel grid of the bench cloud in ONE cloud (1.4 M queries),
wavefront kernel against octet kernel.  Run under rocprofv3 --kernel-trace --stats and read the kernels' durations.  usage: knn_sat.py [k] [copies]"""
import importlib, os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 30
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pair = syn.make_pair(200000, index=0)
ctx = P._lib.Context.current()
base = P.PointCloud(pair.source).voxel_down_sample(0.1).points.astype(np.float32)
pts = np.concatenate([base + np.array([300.0 * (c % 4), 300.0 * (c // 4), 0.0], np.float32) for c in range(copies)])
n = len(pts)
d = torch.from_numpy(pts).cuda()
idx = torch.empty((n, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((n, k), dtype=torch.float32, device="cuda"); cnt = torch.empty(n, dtype=torch.int32, device="cuda")
for wave in ("1", "0", "1", "0"):
    os.environ["PCR_KNN_WAVE"] = wave
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.check(ctx.lib.pcr_debug_knn(ctx.handle, C.c_void_p(d.data_ptr()), C.c_int64(n), C.c_int(k), C.c_double(0.0), C.c_void_p(idx.data_ptr()), C.c_void_p(d2.data_ptr()), C.c_void_p(cnt.data_ptr())), "knn")
    torch.cuda.synchronize()
    print(f"n {n} k {k} PCR_KNN_WAVE={wave}: import + tree + search {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)

pc = P.PointCloud(pts)
for wave in ("1", "0", "1", "0"):
    os.environ["PCR_KNN_WAVE"] = wave
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pc.remove_statistical_outlier(k, 1.0)
    torch.cuda.synchronize()
    print(f"n {n} k {k} PCR_KNN_WAVE={wave}: remove_statistical_outlier {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
