set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--config5 --pairs-per-step 8 --base-pairs 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --inflight 4"
timeout -k 10 300 python bench.py $A > gpurun_out/c5_plain.log 2>&1
tail -1 gpurun_out/c5_plain.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/c5prof -o c5 --output-format csv -- python bench.py $A > gpurun_out/c5_prof.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c5prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(7), r['TotalDurationNs'].rjust(12), r['AverageNs'][:9].rjust(10), r['Percentage'][:5])
PY
