"""Diagnostic: per-wavefront timeline of the k-NN kernel (PCR_KNN_STAMPS).  Prints occupancy over time, wave life
distribution and placement per XCD for the SOR search of each scale of the config-2 target cloud."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
OUT = "/tmp/knn_stamps.bin"
os.environ["PCR_KNN_STAMPS"] = OUT
import torch
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
syn = importlib.import_module("point-cloud-registration-with-global-refinement_amd.synthetic")
p = syn.make_pair(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)


def analyse(tag):
    raw = np.fromfile(OUT, dtype=np.uint64); os.remove(OUT)
    pos = 0
    while pos < len(raw):
        assert raw[pos] == 0x5354414d50
        mode, k, nw = int(raw[pos + 1]), int(raw[pos + 2]), int(raw[pos + 3])
        w = raw[pos + 4: pos + 4 + 24 * nw].reshape(nw, 24); pos += 4 + 24 * nw
        livew = w[:, 0] != 0
        w = w[livew]
        t0 = w[:, 0].min()
        b = (w[:, 0] - t0).astype(np.float64); e = (w[:, 1] - t0).astype(np.float64); cyc = w[:, 2].astype(np.float64)
        life = e - b
        tick_per_cycle = (life.sum() / cyc.sum())
        span = e.max()
        xcc = (w[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
        hw = (w[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
        cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
        print(f"[{tag}] mode {mode} k {k}: {len(w)} live waves of {nw}; span {span:.0f} ticks; life mean {life.mean():.0f} p50 {np.percentile(life,50):.0f} "
              f"p99 {np.percentile(life,99):.0f} max {life.max():.0f} ticks; cycles/wave mean {cyc.mean():.0f}; ticks per cycle {tick_per_cycle:.4f}")
        edges = np.linspace(0, span, 11)
        conc = [(np.minimum(e, edges[i + 1]) - np.maximum(b, edges[i])).clip(0).sum() / (edges[i + 1] - edges[i]) for i in range(10)]
        print("    mean resident waves per decile of the span:", " ".join(f"{c:.0f}" for c in conc))
        print("    last wave start at", f"{b.max() / span:.2f} of span; waves starting in each decile:", np.histogram(b, bins=edges)[0])
        scan = (w[:, 4] >> np.uint64(32)).astype(np.int64); rounds = (w[:, 4] & np.uint64(0xffffffff)).astype(np.int64); pops = w[:, 7].astype(np.int64); loose = (pops >> 30) & 1; pops = pops & 0xfffff
        qx = (w[:, 5] >> np.uint64(32)).astype(np.uint32).view(np.float32); qy = (w[:, 5] & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32)
        qz = (w[:, 6] >> np.uint64(32)).astype(np.uint32).view(np.float32); worst = (w[:, 6] & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32)
        print(f"    scan steps mean {scan.mean():.1f} p99 {np.percentile(scan,99):.0f} max {scan.max()} | insertion rounds mean {rounds.mean():.1f} p99 {np.percentile(rounds,99):.0f} max {rounds.max()} | node pops mean {pops.mean():.1f} max {pops.max()}")
        print(f"    loose waves {loose.sum()} ({100.0 * loose.mean():.1f}%), life mean loose {life[loose == 1].mean() if loose.any() else 0:.0f} tight {life[loose == 0].mean():.0f}")
        print(f"    corr(life, scan) {np.corrcoef(life, scan)[0,1]:.2f} corr(life, rounds) {np.corrcoef(life, rounds)[0,1]:.2f} corr(life, pops) {np.corrcoef(life, pops)[0,1]:.2f} corr(life, begin) {np.corrcoef(life, b)[0,1]:.2f}")
        for i in np.argsort(-life)[:5]:
            print(f"      slow wave: life {life[i]:.0f} begin {b[i]:.0f} scan {scan[i]} rounds {rounds[i]} pops {pops[i]} q ({qx[i]:.1f},{qy[i]:.1f},{qz[i]:.1f}) loose {loose[i]} kth-dist {np.sqrt(worst[i]):.2f}")
        f32 = lambda a: a.astype(np.uint32).view(np.float32)
        seedb = np.sqrt(f32(w[:, 8:16] >> np.uint64(32))); finb = np.sqrt(f32(w[:, 8:16] & np.uint64(0xffffffff)))
        gx = f32(w[:, 16:24] >> np.uint64(32)); gy = f32(w[:, 16:24] & np.uint64(0xffffffff))
        ext = np.hypot(gx.max(1) - gx.min(1), gy.max(1) - gy.min(1))
        ratio = (seedb / np.maximum(finb, 1e-6)).max(1)
        print(f"    seed/final bound ratio (max over group): mean {ratio.mean():.2f} p50 {np.percentile(ratio,50):.2f} p90 {np.percentile(ratio,90):.2f} p99 {np.percentile(ratio,99):.2f}; corr(life, ratio) {np.corrcoef(life, ratio)[0,1]:.2f} corr(life, ext/final) {np.corrcoef(life, ext / finb.max(1))[0,1]:.2f} corr(life, final) {np.corrcoef(life, finb.max(1))[0,1]:.2f}")
        for i in np.argsort(-life)[:5]:
            print(f"      slow wave {i}: life {life[i]:.0f} xy-extent {ext[i]:.2f} seed bounds {np.round(seedb[i],2)} final {np.round(finb[i],2)}")
        print("    waves per XCC:", np.bincount(xcc, minlength=8), " distinct (xcc,se,cu):", len(set(zip(xcc, se, cu))))


ctx = P._lib.Context.current()
for voxel in p.voxel_sizes:
    pc = P.PointCloud(p.target).voxel_down_sample(voxel)
    if os.path.exists(OUT): os.remove(OUT)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pc2, _ = pc.remove_statistical_outlier(30, 1.0)
    torch.cuda.synchronize()
    print(f"voxel {voxel}: n {len(pc)} -> {len(pc2)}  call {1e3 * (time.perf_counter() - t0):.2f} ms")
    analyse(f"v{voxel}")
