"""Sum rocprofv3 --pmc counters per kernel name.  usage: pmc_sum.py <dir> [name filter]"""
import csv, glob, sys, collections
f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"][:60]
    if flt and flt not in name: continue
    d[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], name)
    if key not in seen:
        seen.add(key); calls[name] += 1; dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name in sorted(d, key=lambda k: -sum(d[k].values())):
    print(f"{name:60s} calls {calls[name]:5d} us {dur[name]:10.1f} " + " ".join(f"{c} {v:.4g}" for c, v in sorted(d[name].items())))
