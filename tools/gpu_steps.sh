#!/usr/bin/env bash
# Runs the given GPU steps one after another on the gpurun box.  A step that fails normally (non-zero exit)
# does not stop the session; a step killed by its timeout does (no further GPU work after a hang).
# usage: tools/gpu_steps.sh "<name>|<timeout s>|<command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== step $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name rc=$rc in $(( $(date +%s) - start ))s"; tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== step $name timed out: stopping"; exit 1; fi
done
exit 0
