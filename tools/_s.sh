{
tools/ab.sh 1 "PCR_X=1" -- --inflight 4 --group 6
tools/ab.sh 1 "PCR_X=1" -- --inflight 4 --group 8
tools/ab.sh 1 "PCR_X=1" -- --inflight 5 --group 6 --pairs-per-step 60
tools/ab.sh 1 "PCR_X=1" -- --inflight 6 --group 4
tools/ab.sh 1 "PCR_X=1" -- --inflight 3 --group 8
tools/ab.sh 1 "PCR_X=1" -- --inflight 6 --group 8 --pairs-per-step 96
tools/ab.sh 1 "PCR_X=1" -- --inflight 4 --group 12 --pairs-per-step 96
} > gpurun_out/sweep2.log 2>&1
cat gpurun_out/sweep2.log
