"""GPU-side overview of a rocprofv3 --kernel-trace: busy fraction (union of kernel intervals), mean concurrency, kernels per second,
per-queue gap statistics (end -> next start on the same queue).  usage: trace_overview.py <dir> [skip_fraction]"""
import collections, csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Stream_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
wall = (max(r[1] for r in rows) - rows[0][0]) / 1e3
ev = sorted([(s, 1) for s, e, q, st in rows] + [(e, -1) for s, e, q, st in rows])
busy = 0.0; conc_area = 0.0; cur = 0; last = ev[0][0]
for t, d in ev:
    if cur > 0: busy += t - last
    conc_area += cur * (t - last); last = t; cur += d
print(f"window {wall / 1e3:.2f} ms, {len(rows)} kernels = {len(rows) / wall * 1e3:.0f} k/s, busy {busy / 1e3 / wall:.3f}, mean concurrency {conc_area / 1e3 / wall:.2f}, sum of kernel time {sum(e - s for s, e, q, st in rows) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for s, e, q, st in rows: byq[q].append((s, e))
gaps = []
for q, l in byq.items():
    l.sort()
    gaps += [(b[0] - a[1]) / 1e3 for a, b in zip(l, l[1:])]
g = np.array(gaps)
print(f"queues {len(byq)}; same-queue gaps: n {len(g)} mean {g.mean():.1f} p50 {np.percentile(g, 50):.1f} p90 {np.percentile(g, 90):.1f} p99 {np.percentile(g, 99):.1f} us; share of gaps > 20 us: {(g > 20).mean():.3f}, their sum {g[g > 20].sum() / 1e3:.1f} ms of {wall / 1e3 * len(byq):.1f} queue-ms")
d = np.array([(e - s) / 1e3 for s, e, q, st in rows])
print(f"kernel durations: mean {d.mean():.1f} p50 {np.percentile(d, 50):.1f} p90 {np.percentile(d, 90):.1f} us")
