#!/bin/bash
# usage: tools/prof.sh TAG [bench args...]  -- kernel-trace stats of one bench run, top rows to gpurun_out/TAG_stats.txt
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$ROOT/gpurun_out/bench_$TAG.log" 2> "$ROOT/gpurun_out/bench_$TAG.err"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" > "$ROOT/gpurun_out/${TAG}_stats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:16]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:7.2f} ms {float(r["Percentage"]):5.1f}%')
PY
grep '^{"metric' "$ROOT/gpurun_out/bench_$TAG.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pairs/s', round(d['value'],2), 'ms/pair', round(d['ms_per_step'],3))" >> "$ROOT/gpurun_out/${TAG}_stats.txt"
cat "$ROOT/gpurun_out/${TAG}_stats.txt"
