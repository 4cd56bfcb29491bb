"""Timing of the mutual nearest-feature search alone (pcr_debug_feature_nn) on the FPFH features of the synthetic 200k pair.
usage: featnn_bench.py [n_points] [mode]; PCR_FEATNN_VARIANT selects diagnostic variants of the screen kernel (timing only)."""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = syn.make_pair(200000)
sub = np.random.default_rng(7).permutation(200000)[:n]
S, T = P.PointCloud(p.source[sub]), P.PointCloud(p.target[sub])
for c in (S, T):
    c.estimate_normals(P.KDTreeSearchParamHybrid(0.2, 20))
fs = P.registration.compute_fpfh_feature(S, P.KDTreeSearchParamHybrid(1.0, 200))._dev
ft = P.registration.compute_fpfh_feature(T, P.KDTreeSearchParamHybrid(1.0, 200))._dev
ctx = P._lib.Context.current()
o10 = torch.empty(len(T), dtype=torch.int32, device="cuda"); o01 = torch.empty(len(S), dtype=torch.int32, device="cuda")
def run():
    ctx.check(ctx.lib.pcr_debug_feature_nn(ctx.handle, C.c_void_p(fs.data_ptr()), C.c_int64(len(S)), C.c_void_p(ft.data_ptr()), C.c_int64(len(T)),
                                           C.c_void_p(o10.data_ptr()), C.c_void_p(o01.data_ptr()), C.c_int(mode)), "nn")
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
print(f"featnn n={n} mode={mode} variant={os.environ.get('PCR_FEATNN_VARIANT', '0')}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms for both directions")
if os.environ.get("FEATNN_SUM"):      # order-independent digest of both answers (to compare paths / switches across processes)
    print("featnn digest", int(o10.to(torch.int64).sum().item()), int(o01.to(torch.int64).sum().item()), int((o10.to(torch.int64) * torch.arange(len(T), device="cuda")).sum().item()))
