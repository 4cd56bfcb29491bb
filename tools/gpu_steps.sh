#!/bin/bash
# Runs the lines of a step file one after the other on the GPU box (each under its own `timeout -k 10`), logging to gpurun_out/<tag>/.
# An ordinary failure (non-zero exit) is recorded and the next step runs; a step that TIMES OUT or is killed ends the whole call
# (no further GPU step after a hang).  usage: tools/gpu_steps.sh <tag> <steps-file>   with lines "<seconds> <name> <command...>"
TAG=$1; STEPS=$2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
while IFS= read -r line; do
  [ -z "$line" ] && continue
  case "$line" in \#*) continue;; esac
  secs=${line%% *}; rest=${line#* }; name=${rest%% *}; cmd=${rest#* }
  echo "[$(date +%H:%M:%S)] step $name (limit ${secs}s)"
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/$name.log" 2> "$OUT/$name.err" < /dev/null
  rc=$?
  echo "[$(date +%H:%M:%S)] step $name rc=$rc"; tail -n 3 "$OUT/$name.log" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
done < "$STEPS"
exit 0
