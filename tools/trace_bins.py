"""Reads a rocprofv3 kernel-trace CSV and prints, in bins of the last part of the run, the share of time with no kernel resident, the mean number
resident and the busy share of the fat preprocessing kernels against the iteration kernels.  usage: trace_bins.py <dir> [bin_ms=2] [last_ms=160]"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
binms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
last = float(sys.argv[3]) if len(sys.argv) > 3 else 160.0
t1 = max(r[1] for r in rows); t0 = t1 - int(last * 1e6)
nb = int(last / binms); res = 20000          # 20 us resolution
n = int(last * 1e6 / res)
tot = np.zeros(n); fat = np.zeros(n); icp = np.zeros(n)
for s, e, name in rows:
    if e <= t0: continue
    a = max(0, (s - t0) // res); b = min(n - 1, (e - t0) // res)
    tot[a:b + 1] += 1
    if "knn" in name or "normals_from" in name or "sor_" in name or "rs_" in name or "voxel" in name or "oct_" in name: fat[a:b + 1] += 1
    if "icp_" in name: icp[a:b + 1] += 1
per = n // nb
for k in range(nb):
    sl = slice(k * per, (k + 1) * per)
    print(f"{k * binms:6.0f} ms  idle {100 * np.mean(tot[sl] == 0):5.1f} %  resident {tot[sl].mean():4.1f}  prep kernels {fat[sl].mean():4.1f}  icp kernels {icp[sl].mean():4.1f}")
