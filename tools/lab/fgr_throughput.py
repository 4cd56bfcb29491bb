"""Diagnostic: registro_FGR (script-1 parameters) throughput on NCLT-size clouds with N host threads, one stream each."""
import importlib, os, sys, threading, time, copy
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "nclt_pair_500.npz"))
S, T = P.PointCloud(g["source"]), P.PointCloud(g["target"])
def loop(n, out, k):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        P.script1.registro_FGR(copy.deepcopy(S), copy.deepcopy(T), 0.1, seed=1)
        s.synchronize(); t0 = time.perf_counter()
        for i in range(n): P.script1.registro_FGR(copy.deepcopy(S), copy.deepcopy(T), 0.1, seed=1 + i)
        s.synchronize(); out[k] = time.perf_counter() - t0
for N in (1, 2, 4, 6):
    out = [0.0] * N; n = 12
    th = [threading.Thread(target=loop, args=(n, out, k)) for k in range(N)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
    print(f"{N} threads: {N * n / max(out):.1f} FGR pairs/s ({1e3 * max(out) / n:.2f} ms per pair per thread)")
