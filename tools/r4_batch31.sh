#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_fwd_dgrad_wgrad" 2>&1 | tail -4
for s1 in 0 1 2; do echo "SBA_WGRAD_S1=$s1"; SBA_WGRAD_S1=$s1 BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep -E " 3x3 |4x4s2  M=(16|40)"; done
