"""Per-boundary gaps of the GICP iteration chain from a rocprofv3 --kernel-trace: for consecutive k_icp_fused
dispatches of the same stream, the kernel duration, the gap end -> next start, and the period start -> start; no-op launches (after
'done', < 6 us) are reported separately.  usage: python tools/icp_gap_hist.py <dir with *_kernel_trace.csv> [out.txt]"""
import collections, csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
by_stream = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    by_stream[(r["Queue_Id"], r["Stream_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"])))
dur, gap, period, noop, gap_by_prev = [], [], [], [], collections.defaultdict(list)
for rows in by_stream.values():
    rows.sort()
    for (s0, e0, n0, g0), (s1, e1, n1, g1) in zip(rows, rows[1:]):
        if "k_icp_fused" not in n1:
            continue
        d1 = (e1 - s1) / 1e3
        if d1 < 6.0:
            noop.append(d1); continue
        dur.append((d1, g1))
        gap_by_prev[n0 if "k_icp" in n0 else "other"].append((s1 - e0) / 1e3)
        if "k_icp_fused" in n0 and (e0 - s0) / 1e3 >= 6.0:
            gap.append((s1 - e0) / 1e3); period.append((s1 - s0) / 1e3)
out = []
def pct(a, name):
    a = np.array(a)
    if len(a) == 0:
        out.append(f"{name}: none"); return
    out.append(f"{name}: n {len(a)} mean {a.mean():.1f} p10 {np.percentile(a, 10):.1f} p50 {np.percentile(a, 50):.1f} p90 {np.percentile(a, 90):.1f} p99 {np.percentile(a, 99):.1f} us")
pct([d for d, g in dur], "k_icp_fused live duration")
for lo, hi in ((0, 70000), (70000, 130000), (130000, 10**9)):
    pct([d for d, g in dur if lo <= g < hi], f"  grid threads in [{lo}, {hi})")
pct(gap, "gap end -> next start (live -> live, same stream)")
pct(period, "period start -> start (live -> live)")
pct(noop, "no-op launches after convergence")
for k, v in gap_by_prev.items():
    pct(v, f"gap before a live k_icp_fused when the previous kernel on the stream was {k}")
h, edges = np.histogram(np.array(gap), bins=[0, 1, 2, 3, 5, 8, 12, 20, 40, 80, 1e9]) if gap else ([], [])
out.append("gap histogram (us): " + ", ".join(f"[{edges[i]:g},{edges[i + 1]:g}): {h[i]}" for i in range(len(h))))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
