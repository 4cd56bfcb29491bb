#!/bin/bash
# interleaved A/B of environments on the default bench (pairs/s): ab.sh rounds "ENV_A=.. ENV_A2=.." "ENV_B=.." ... -- [extra bench args]
R=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq 1 $R); do
  for e in "${envs[@]}"; do
    v=$(env $e python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['config']['iterations_per_pair_mean'],1))")
    echo "round $r [$e] $*: $v"
  done
done
