#!/usr/bin/env bash
# primitives per leaf vs throughput (16 spp)
mkdir -p gpurun_out
for ml in 0 1 3 4; do
  TAKE_HIP_MAX_LEAF=$ml timeout -k 10 200 python bench.py --spp 16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/leaf_$ml.log 2>&1; rc=$?
  echo "max_leaf=$ml rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/leaf_$ml.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/leaf_$ml.log) $(grep -o '"bytes_per_ray": [0-9.]*' gpurun_out/leaf_$ml.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
