#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_fwd_dgrad_wgrad" 2>&1 | tail -4
for a in 0 1; do echo "SBA_WGRAD_ALLROWS=$a"; SBA_WGRAD_ALLROWS=$a BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep -E " 3x3up | 3x3 .*M=(81920|327680)"; done
for w in 256 512; do echo "SBA_WGRAD_ALL9_WGS=$w"; SBA_WGRAD_ALL9_WGS=$w BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep -E " 3x3up | 3x3 .*M=(81920|327680)"; done
