#!/usr/bin/env bash
# batch-size sweep: paths in flight per batch vs throughput (64 spp total)
mkdir -p gpurun_out
for spb in 8 16 32 64; do
  timeout -k 10 200 python bench.py --spp 64 --spb $spb --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/spb_$spb.log 2>&1; rc=$?
  echo "spb=$spb rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/spb_$spb.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/spb_$spb.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
