#!/bin/bash
cd "$(dirname "$0")/.."
for m in 4 0; do
  SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=$m timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_dist1_final_mode$m.json 2> gpurun_out/r4_dist1_final_mode$m.err
  echo "one rank, SBA_DP_REPLAY=$m: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_dist1_final_mode$m.json) $(grep -o '"launch": "[a-z0-9-]*"' gpurun_out/r4_dist1_final_mode$m.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_dist1_final_mode$m.json)"
done
timeout -k 10 300 python bench.py --child --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_single_final.json 2> gpurun_out/r4_single_final.err; echo "single GPU: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_single_final.json) $(grep 'launch probe' gpurun_out/r4_single_final.err)"
