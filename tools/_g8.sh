set -e
timeout -k 10 600 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_groups.py -x -q -m gpu > gpurun_out/t_g8.log 2>&1 || { tail -40 gpurun_out/t_g8.log; exit 1; }
tail -2 gpurun_out/t_g8.log
tools/ab.sh 2 "PCR_X=1" -- 
A="--config5 --pairs-per-step 8 --base-pairs 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --inflight 4"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', round(d['value'],2), 'it', d['config']['iterations_per_pair_mean'], d['config']['err_vs_planted'], 'us/launch', round(r['us_per_launch_hip_events'],1), 'frac', round(r['frac'],4))"; }
PCR_ICP_STREAM_MIN=0 timeout -k 10 300 python bench.py $A > gpurun_out/c5_fused.log 2>&1; show gpurun_out/c5_fused.log c5-fused
timeout -k 10 300 python bench.py $A > gpurun_out/c5_stream.log 2>&1; show gpurun_out/c5_stream.log c5-stream
tools/ab.sh 1 "PCR_X=1" -- --points 20000 --pairs-per-step 192
tools/ab.sh 1 "PCR_X=1" -- --inflight 1 --group 1 --pairs-per-step 8
