"""Diagnostic: per-wavefront timeline of the GICP iteration kernels (PCR_ICP_STAMPS) for the config-2 pair."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
OUT = "/tmp/icp_stamps.bin"
if os.path.exists(OUT): os.remove(OUT)
os.environ["PCR_ICP_STAMPS"] = OUT
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
p = syn.make_pair(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)
from importlib import import_module
reg = import_module("point-cloud-registration-with-global-refinement_amd.registration")
r = reg.multiscale_gicp(P.PointCloud(p.source), P.PointCloud(p.target), p.voxel_sizes, p.max_distances_script, p.T_init,
                        estimation_method=reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()), criteria=reg.ICPConvergenceCriteria(1e-6, 1e-6, 100))
raw = np.fromfile(OUT, dtype=np.uint64)
pos = 0; scale = 0
while pos < len(raw):
    assert raw[pos] == 0x49435053
    wn, wi, L, launches, ns = (int(x) for x in raw[pos + 1: pos + 6]); pos += 6
    nn = raw[pos: pos + L * wn * 2].reshape(L, wn, 2); pos += L * wn * 2
    it = raw[pos: pos + L * wi * 12].reshape(L, wi, 12); pos += L * wi * 12
    print(f"== scale {scale}: ns {ns}, launches {launches}")
    scale += 1
    for l in range(1, min(L, launches), 3):
        a = nn[l]; live = a[:, 1] != 0; a = a[live].astype(np.int64)
        b = it[l]; liveb = b[:, 0] != 0; b = b[liveb].astype(np.int64)
        t0 = a[:, 0].min()
        nb, ne = a[:, 0] - t0, a[:, 1] - t0
        ib, il, ir = b[:, 0] - t0, b[:, 1] - t0, b[:, 2] - t0
        fin = b[:, 3].max() - t0
        life = ne - nb
        print(f"  launch {l}: NN waves {len(a)} first start 0, last start {nb.max() * 0.01:.1f} us, end p50 {np.percentile(ne, 50) * 0.01:.1f} p99 {np.percentile(ne, 99) * 0.01:.1f} max {ne.max() * 0.01:.1f} us; life mean {life.mean() * 0.01:.1f} p99 {np.percentile(life, 99) * 0.01:.1f} max {life.max() * 0.01:.1f} us")
        lastw = b[np.argmax(b[:, 3])]
        print(f"            ITER phases (us): wave-sums done max {(b[:, 4].max() - t0) * 0.01:.1f}; stores drained max {(b[:, 5].max() - t0) * 0.01:.1f}; ticket returned max {ir.max() * 0.01:.1f} | last wg: partials summed {(lastw[6] - t0) * 0.01:.1f}, sums in LDS {(lastw[7] - t0) * 0.01:.1f}, solved {(lastw[8] - t0) * 0.01:.1f}, done {(lastw[3] - t0) * 0.01:.1f}")
        print(f"            ITER waves {len(b)} start min {ib.min() * 0.01:.1f} max {ib.max() * 0.01:.1f}; loop end p50 {np.percentile(il, 50) * 0.01:.1f} max {il.max() * 0.01:.1f}; published max {ir.max() * 0.01:.1f}; iteration finished {fin * 0.01:.1f} us")
    if scale > 8: break
