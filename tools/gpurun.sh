#!/bin/bash
# gpurun with a pre-flight: rebuild the in-tree library (the GPU box runs the snapshot's .so) and refuse to go when it does not build.
# usage: tools/gpurun.sh <tag> <steps-file> [timeout]
set -e -o pipefail
cd "$(dirname "$0")/.."
bash point-cloud-registration-with-global-refinement_amd/csrc/build.sh 2>&1 | grep -v "warning\|^ \|^$\|generated" | tail -3
make -s -C oracle > /dev/null 2>&1 || true
mkdir -p gpurun_out/$1
gpurun --timeout ${3:-1200} -- "bash tools/gpu_steps.sh $1 $2" > gpurun_out/$1/call.log 2>&1
echo finished >> gpurun_out/$1/call.log
