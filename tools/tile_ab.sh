for t in 512 256 128 policy; do
  if [ $t = policy ]; then unset PCR_ICP_TILE; else export PCR_ICP_TILE=$t; fi
  python bench.py --inflight 1 --group 1 --no-extras --no-cpu-baseline --steps 2 --warmup 1 --pairs-per-step 16 --fixed-iterations 25 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('solo fixed25 tile $t: %.1f pairs/s, us/launch events %.1f in-kernel %.1f'%(d['value'], r['us_per_launch_hip_events'], r['us_per_launch_in_kernel_clock']))"
done
for t in 1024 512 256; do
  export PCR_ICP_TILE=$t
  python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --fixed-iterations 25 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('2x4 fixed25 tile $t: %.1f pairs/s'%(d['value']))"
done
