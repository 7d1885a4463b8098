#!/usr/bin/env bash
# host SAH vs device LBVH: scene_create time and render throughput (1M and 10M triangles, 16 spp)
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python bench.py --spp 16 --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bs_$name.log 2>&1; rc=$?
  echo "$name rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/bs_$name.log | head -1) $(grep -o '"setup_s": [0-9.]*' gpurun_out/bs_$name.log) $(grep -o '"nodes": [0-9]*, "prims": [0-9]*, "depth": [0-9]*' gpurun_out/bs_$name.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi; }
run host_1m
run dev_1m_leaf2 --builder device --max-leaf 2
run dev_1m_leaf1 --builder device --max-leaf 1
run dev_1m_leaf4 --builder device --max-leaf 4
run host_10m --tris 10000000
run dev_10m_leaf1 --tris 10000000 --builder device --max-leaf 1
