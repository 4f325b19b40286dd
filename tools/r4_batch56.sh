#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_fwd_dgrad_wgrad and 4x4s2" 2>&1 | tail -3
BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep -E "4x4s2"
rm -f gpurun_out/r4_pmc_wgrad_row2.txt; bash tools/pmc_wgrad.sh r4_pmc_wgrad_row2 "128->256 @64"; grep -A1 "BANK_CONFLICT\|WAVE_CYCLES" gpurun_out/r4_pmc_wgrad_row2.txt | grep -v "^--"
