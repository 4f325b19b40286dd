#!/bin/bash
# experiment: the real half of the discriminator loss, backward included, at the start of the step (SBA_REAL_BWD_EARLY=1)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/real_bwd_early_check.py > gpurun_out/r4_b87_check.txt 2>&1 || { tail -n 30 gpurun_out/r4_b87_check.txt; exit 1; }
grep -v "Warning\|warn\|^  super\|amdgpu.ids" gpurun_out/r4_b87_check.txt | tail -n 40
for r in 1 2; do
  timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b87_default_$r.json 2> gpurun_out/r4_b87_default_$r.err || exit 1
  echo "default (grouped real|fake pass)            $r: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b87_default_$r.json)"
  SBA_REAL_FIRST=1 SBA_REAL_BWD_EARLY=1 timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b87_early_$r.json 2> gpurun_out/r4_b87_early_$r.err || exit 1
  echo "real half first, backward included          $r: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b87_early_$r.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b87_early_$r.json)"
done
SBA_REAL_BWD_EARLY=1 SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=4 timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b87_dist1_early.json 2> gpurun_out/r4_b87_dist1_early.err || exit 1
echo "data-parallel, one rank, early real backward : $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b87_dist1_early.json) $(grep -o '"launch": "[a-z0-9-]*"' gpurun_out/r4_b87_dist1_early.json)"
SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=4 timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b87_dist1.json 2> gpurun_out/r4_b87_dist1.err || exit 1
echo "data-parallel, one rank                      : $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b87_dist1.json)"
