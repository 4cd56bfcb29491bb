"""Reads a rocprofv3 kernel-trace CSV and prints, for the densest part of the run (the timed steps of bench.py), how many
kernels are resident over time: fraction of wall time with 0 / 1 / 2 ... kernels running, and per-kernel-name busy time.
usage: python tools/trace_concurrency.py <dir with *_kernel_trace.csv> [skip_fraction]"""
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
lo = t0 + (t1 - t0) * skip                    # the second half of the run = timed steps (warm-up and setup are before)
ev = []
for s, e, n in rows:
    if e <= lo: continue
    ev.append((max(s, lo), 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = lo
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print(f"window {tot / 1e6:.1f} ms, kernels {sum(1 for r in rows if r[1] > lo)}")
acc = 0
for k in sorted(hist):
    print(f"  {k:2d} kernels resident: {100.0 * hist[k] / tot:5.1f} % of wall time")
print(f"  mean concurrency {sum(k * v for k, v in hist.items()) / tot:.2f}")
busy = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    if e > lo: busy[n.split('(')[0][:60]] += e - max(s, lo); cnt[n.split('(')[0][:60]] += 1
for n, b in busy.most_common(14):
    print(f"  {n:60s} {b / 1e6:8.2f} ms  {cnt[n]:6d} launches  avg {b / cnt[n] / 1e3:7.1f} us")

# ---- per-queue dependent-launch gaps: time between the end of a kernel and the start of the next one in the same queue
hdr = next(csv.reader(open(f)))
qcol = "Queue_Id" if "Queue_Id" in hdr else None
if qcol:
    byq = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if int(r["End_Timestamp"]) > lo: byq[r[qcol]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split('(')[0][:40]))
    gaps = []; after = collections.defaultdict(list)
    for q, ks in byq.items():
        ks.sort()
        for a, b in zip(ks, ks[1:]):
            g = b[0] - a[1]
            if 0 <= g < 200_000: gaps.append(g); after[a[2]].append(g)       # > 200 us: the queue was simply empty
    gaps = np.array(gaps, dtype=np.float64) / 1e3
    print(f"queues {len(byq)}; back-to-back gaps (<200 us): n {len(gaps)} mean {gaps.mean():.1f} us p50 {np.percentile(gaps,50):.1f} p90 {np.percentile(gaps,90):.1f}; sum {gaps.sum() / 1e3:.1f} ms")
    for n, g in sorted(after.items(), key=lambda kv: -sum(kv[1]))[:10]:
        g = np.array(g) / 1e3
        print(f"  gap after {n:40s} n {len(g):6d} mean {g.mean():6.1f} us p50 {np.percentile(g,50):6.1f}  sum {g.sum() / 1e3:7.1f} ms")
