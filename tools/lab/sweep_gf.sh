#!/bin/bash
# group x inflight sweep of the default bench (pairs/s): sweep_gf.sh "6x4 4x6 ..."
for gf in $1; do
  g=${gf%x*}; f=${gf#*x}
  v=$(python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 --group $g --inflight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['config']['iterations_per_pair_mean'],1))")
  echo "group $g x $f in flight: $v"
done
