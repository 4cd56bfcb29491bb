"""Diagnostic: kernel launches per registered pair (run under rocprofv3; prints the per-kernel table of a stats CSV)."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0])))
tot = 0
for r in rows:
    tot += int(r["Calls"])
    print("%4d %8.1f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:70]))
print("total launches", tot)
