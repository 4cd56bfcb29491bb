set -e
A="--config5 --pairs-per-step 8 --base-pairs 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --inflight 4"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', round(d['value'],2), 'it', d['config']['iterations_per_pair_mean'], d['config']['err_vs_planted'], 'us/launch', round(r['us_per_launch_hip_events'],1), 'frac', round(r['frac'],4))"; }
PCR_ICP_STREAM_MIN=0 timeout -k 10 300 python bench.py $A > gpurun_out/c5_fused.log 2>&1; show gpurun_out/c5_fused.log fused
timeout -k 10 300 python bench.py $A > gpurun_out/c5_stream.log 2>&1; show gpurun_out/c5_stream.log stream
PCR_ICP_STREAM_MIN=50000 timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or config2_reference" > gpurun_out/t_stream.log 2>&1 || { tail -30 gpurun_out/t_stream.log; exit 1; }
tail -2 gpurun_out/t_stream.log
for g in 0 1 2; do timeout -k 10 300 python bench.py --variant fgr --no-cpu-baseline --no-extras --steps 2 --warmup 1 --group $g > gpurun_out/fgr_g$g.log 2>&1; tail -1 gpurun_out/fgr_g$g.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fgr group $g', round(d['value'],1), d['config']['lockstep_group'])"; done
