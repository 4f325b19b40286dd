#!/bin/bash
# round-4 GPU batch: DAMSM matrix-core kernels, determinism, two-rank data-parallel test, precision split, one-rank cost of the DP launch modes
cd "$(dirname "$0")/.."
set -x
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "damsm" 2>&1 | tail -4
timeout -k 10 120 python tools/bench_damsm.py 2>&1 | grep -E "launches"
timeout -k 10 600 python -m pytest tests/test_determinism_gpu.py -x -q 2>&1 | tail -6
timeout -k 10 800 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -12
timeout -k 10 300 python tools/precision_split.py model_b20 > gpurun_out/r4_precision_split_model_b20.txt 2>&1; tail -8 gpurun_out/r4_precision_split_model_b20.txt
for m in 3 2 0; do
  SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=$m timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_dist1_mode$m.json 2> gpurun_out/r4_dist1_mode$m.err
  grep -E "launch probe|unavailable" gpurun_out/r4_dist1_mode$m.err; tail -n 1 gpurun_out/r4_dist1_mode$m.json | cut -c1-160
done
timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_single_v2.json 2> gpurun_out/r4_single_v2.err; grep "launch probe" gpurun_out/r4_single_v2.err; tail -n 1 gpurun_out/r4_single_v2.json | cut -c1-160
