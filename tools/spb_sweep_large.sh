for spb in 64 128 256; do
  timeout -k 10 400 python bench.py --spp 256 --spb $spb --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/spb2_$spb.log 2>&1; rc=$?
  echo "spb=$spb rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/spb2_$spb.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/spb2_$spb.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
