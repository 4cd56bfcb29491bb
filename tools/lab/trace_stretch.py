"""From a rocprofv3 kernel trace of the in-flight bench: mean duration of the small kernels as a function of what else was
resident when they started (number of k_knn kernels, number of k_icp_fused kernels, number of other kernels).
usage: python tools/trace_stretch.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections, bisect
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split('(')[0]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * 0.4
rows = [r for r in rows if r[0] > lo]
def cls(n):
    if "k_knn<0" in n: return "knn_sor"
    if "k_knn<1" in n: return "knn_nrm"
    if "k_icp_fused" in n: return "fused"
    return "small"
iv = collections.defaultdict(list)
for s, e, n in rows: iv[cls(n)].append((s, e))
def resident(c, t):
    return sum(1 for s, e in iv[c] if s <= t < e) if len(iv[c]) < 3000 else None
# fast resident counters via sorted starts/ends
cnt = {}
for c, l in iv.items():
    cnt[c] = (np.sort([s for s, e in l]), np.sort([e for s, e in l]))
def res(c, t):
    st, en = cnt[c]
    return int(np.searchsorted(st, t, side="right") - np.searchsorted(en, t, side="right"))
for c in ("knn_sor", "knn_nrm", "fused"):
    d = np.array([e - s for s, e in iv[c]]) / 1e3
    print(f"{c}: n {len(d)} mean {d.mean():.1f} us")
wall = (max(r[1] for r in rows) - min(r[0] for r in rows)) / 1e6
print(f"window {wall:.1f} ms")
targets = sys.argv[2].split(",") if len(sys.argv) > 2 else ("k_rs_scatter", "k_oct_apply", "k_oct_level_boxes", "k_voxel_mean", "k_scan_tile_apply", "k_icp_fused")
for target in targets:
    tab = collections.defaultdict(list)
    for s, e, n in rows:
        if target in n:
            k = (min(res("knn_sor", s), 2), min(res("knn_nrm", s), 1), min(res("fused", s) - (1 if target == "k_icp_fused" else 0), 3))
            tab[k].append((e - s) / 1e3)
    print(target)
    for k in sorted(tab):
        v = np.array(tab[k])
        if len(v) >= 20: print(f"   knn_sor {k[0]} knn_nrm {k[1]} fused {k[2]}: n {len(v):5d} mean {v.mean():6.1f} us p50 {np.percentile(v, 50):6.1f}")
