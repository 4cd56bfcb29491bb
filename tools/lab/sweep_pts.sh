#!/bin/bash
# points x tile x group sweep of the bench's GICP form (pairs/s at fixed iterations): sweep_pts.sh "50000 100000" "512 1024" "8 12 24"
for n in $1; do for t in $2; do for g in $3; do
  v=$(PCR_ICP_TILE=$t python bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 2 --points $n --pairs-per-step 96 --group $g --inflight 4 --fixed-iterations 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "points $n tile $t group $g x 4: $v"
done; done; done
