#!/bin/bash
# usage: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- A/B bench variants inside one GPU call (same box, interleaved twice)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for rep in $(seq 1 ${AB_REPS:-2}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    for k in ${AB_INFLIGHT:-1 3}; do
      env $v python bench.py --no-cpu-baseline --steps ${AB_STEPS:-4} --inflight $k > gpurun_out/ab_${i}_$k.log 2>&1
      grep "^{" gpurun_out/ab_${i}_$k.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$v] inflight $k:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms', [s['iterations'] for s in d['config']['scales']])"
    done
  done
done
