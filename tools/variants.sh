#!/usr/bin/env bash
# A/B runs of tuning builds: tools/variants.sh <bench args> -- name1 name2 ...   (libs in take_amd/variants/lib_<name>.so)
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
mkdir -p gpurun_out
for v in base "$@"; do
  if [ "$v" = base ]; then unset TAKE_HIP_LIB; else export TAKE_HIP_LIB="$PWD/take_amd/variants/lib_$v.so"; fi
  timeout -k 10 150 python bench.py "${args[@]}" > gpurun_out/var_$v.log 2>&1; rc=$?
  echo "$v rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/var_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/var_$v.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
