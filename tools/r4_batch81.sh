#!/bin/bash
# final build: the single-GPU step and the data-parallel step with one rank (all host-call nodes live), same box, alternating
set -o pipefail
mkdir -p gpurun_out
for r in 1 2; do
  timeout -k 10 300 python bench.py --child --steps 30 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b81_single_$r.json 2> gpurun_out/r4_b81_single_$r.err || exit 1
  echo "single  $r: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b81_single_$r.json)"
  SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=4 timeout -k 10 300 python bench.py --child --steps 30 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b81_dist1_$r.json 2> gpurun_out/r4_b81_dist1_$r.err || exit 1
  echo "dist1   $r: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b81_dist1_$r.json) $(grep -o '"launch": "[a-z0-9-]*"' gpurun_out/r4_b81_dist1_$r.json)"
done
