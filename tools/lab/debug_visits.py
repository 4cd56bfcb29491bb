"""Diagnostic: node/leaf visit counts of the GICP 1-NN walk per source point (PCR_DEBUG_VISITS=1)."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PCR_DEBUG_VISITS", "1")
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
p = syn.make_pair(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)
for voxel, dist in zip(p.voxel_sizes, p.max_distances_script):
    cl = []
    for c in (p.source, p.target):
        pc = P.PointCloud(c).voxel_down_sample(voxel); pc, _ = pc.remove_statistical_outlier(30, 1.0); pc.estimate_normals(P.KDTreeSearchParamKNN(20)); cl.append(pc)
    src, tgt = cl
    ctx = P._lib.Context.current()
    prm = P._lib.PcrGicpParams(1, 1.0, 1e-3, 1e-6, 1e-6, 30)
    JTJ = np.zeros(36); JTr = np.zeros(6); st = np.zeros(3)
    match = torch.empty(len(src), dtype=torch.int32, device="cuda")
    T = np.ascontiguousarray(p.T_init); dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ctx.check(ctx.lib.pcr_debug_gicp_linearize(ctx.handle, C.c_void_p(src.device_xyz().data_ptr()), C.c_void_p(src.device_normals().data_ptr()), C.c_int64(len(src)),
        C.c_void_p(tgt.device_xyz().data_ptr()), C.c_void_p(tgt.device_normals().data_ptr()), C.c_int64(len(tgt)), C.c_double(dist), dp(T), C.byref(prm), dp(JTJ), dp(JTr), dp(st), C.c_void_p(match.data_ptr())), "lin")
    v = match.cpu().numpy()
    pops, asc, leafs = v & 1023, (v >> 10) & 1023, v >> 20
    print(f"voxel {voxel} dist {dist} ns {len(src)}: pops mean {pops.mean():.1f} p99 {np.percentile(pops,99):.0f} | ascents mean {asc.mean():.1f} p99 {np.percentile(asc,99):.0f} | leaf pops mean {leafs.mean():.1f}")

# ---- k-NN traversal counters on the voxelised target cloud
for voxel in p.voxel_sizes:
    pc = P.PointCloud(p.target).voxel_down_sample(voxel)
    pts = pc.device_xyz(); n = len(pc)
    for k in (30, 20):
        idx = torch.empty((n, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((n, k), dtype=torch.float32, device="cuda")
        cnt = torch.empty(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize(); import time; t0 = time.perf_counter()
        ctx.check(ctx.lib.pcr_debug_knn(ctx.handle, C.c_void_p(pts.data_ptr()), C.c_int64(n), C.c_int(k), C.c_double(0.0),
                                        C.c_void_p(idx.data_ptr()), C.c_void_p(d2.data_ptr()), C.c_void_p(cnt.data_ptr())), "knn")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        v = cnt.cpu().numpy(); pops, asc, leafs = v & 1023, (v >> 10) & 1023, v >> 20
        if os.environ["PCR_DEBUG_VISITS"] == "2":
            us = v.reshape(-1)[: (n // 8) * 8].reshape(-1, 8).max(1) * 0.01       # per wavefront (8 queries), 100 MHz ticks
            print(f"kNN voxel {voxel} k {k}: per-wave us mean {us.mean():.1f} p50 {np.percentile(us,50):.1f} p90 {np.percentile(us,90):.1f} p99 {np.percentile(us,99):.1f} max {us.max():.1f}; waves {len(us)}  (call {dt*1e3:.2f} ms)")
            continue
        print(f"kNN voxel {voxel} k {k} n {n}: pops mean {pops.mean():.1f} p99 {np.percentile(pops,99):.0f} | ascents mean {asc.mean():.1f} p99 {np.percentile(asc,99):.0f} | leaf pops mean {leafs.mean():.1f} p99 {np.percentile(leafs,99):.0f}  ({dt*1e3:.2f} ms incl. sort/build)")
