#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "adam" 2>&1 | tail -3
for nt in 1 0 1 0; do
  SBA_ADAM_NT=$nt timeout -k 10 300 python bench.py --child --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_ab_adam_nt$nt.json 2> gpurun_out/r4_ab_adam_nt$nt.err
  echo "SBA_ADAM_NT=$nt $(grep 'launch probe' gpurun_out/r4_ab_adam_nt$nt.err) $(tail -n 1 gpurun_out/r4_ab_adam_nt$nt.json | cut -c60-140)"
done
