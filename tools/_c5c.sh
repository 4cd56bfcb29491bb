set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--config5 --pairs-per-step 2 --base-pairs 2 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --inflight 1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/c5s -o c5 --output-format csv -- python bench.py $A > gpurun_out/c5s.log 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/c5s/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'k_icp' in n: d[n.split('(')[0]+' grid '+r.get('Grid_Size_X', r.get('Grid_Size','?'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    v2=sorted(x for x in v if x>6)
    if v2: print(k.ljust(40), len(v), len(v2), 'live: min %.1f p25 %.1f med %.1f p75 %.1f max %.1f mean %.1f'%(v2[0], v2[len(v2)//4], v2[len(v2)//2], v2[3*len(v2)//4], v2[-1], sum(v2)/len(v2)))
PY
rm -f gpurun_out/c5s/*kernel_trace.csv gpurun_out/c5s/*/*kernel_trace.csv
