# Diagnostic: one pair alone / default bench / FETCH_SIZE per launch of the fused iteration kernel (used for the XCD-aware tile order).
ROOT=$(pwd)
python bench.py --inflight 1 --group 1 --no-extras --no-cpu-baseline --steps 2 --warmup 1 --pairs-per-step 16 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('solo: %.1f pairs/s, us/launch events %.1f in-kernel %.1f'%(d['value'], r['us_per_launch_hip_events'], r['us_per_launch_in_kernel_clock']))"
python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2x4: %.1f pairs/s'%d['value'])"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmx
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmx -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --pairs-per-step 4 --inflight 1 --group 1 --fixed-iterations 20 > /tmp/pmx.log 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob("/tmp/pmx/**/*counter_collection.csv", recursive=True)[0]
v=[]
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("void k_icp_fused<") and (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))>6000: v.append(float(r["Counter_Value"]))
print(f"fused live launches {len(v)}, FETCH_SIZE avg {sum(v)/len(v):.0f} KB -> HBM-corrected {2*sum(v)/len(v)/1024:.2f} MB per launch")
PY
