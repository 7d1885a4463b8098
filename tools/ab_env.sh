#!/usr/bin/env bash
# A/B of bench.py under different environments: tools/ab_env.sh <bench args> -- "NAME=ENV1=V1 ENV2=V2" "NAME2=" ...
# (each spec: a label, '=', then space-separated VAR=value assignments; an empty list = the defaults)
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%=*}"; envs="${spec#*=}"
  ( for kv in $envs; do export "$kv"; done
    timeout -k 10 200 python bench.py "${args[@]}" > gpurun_out/ab_$name.log 2>&1 ); rc=$?
  echo "$name rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/ab_$name.log | head -2 | tr '\n' ' ') $(grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/ab_$name.log | tr '\n' ' ') $(grep -o '"ms_trace_closest": [0-9.]*, "ms_trace_shadow": [0-9.]*, "ms_shade": [0-9.]*' gpurun_out/ab_$name.log | tr '\n' ' ') $(grep -o '"retried_closest": [0-9.e-]*, "retried_shadow": [0-9.e-]*' gpurun_out/ab_$name.log | head -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
