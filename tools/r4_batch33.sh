#!/bin/bash
cd "$(dirname "$0")/.."
SBA_WGRAD_S1=3 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_fwd_dgrad_wgrad" 2>&1 | tail -4
for s1 in 2 3; do echo "SBA_WGRAD_S1=$s1"; SBA_WGRAD_S1=$s1 BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep -E " 3x3up "; done
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b33_$tag.json 2> gpurun_out/r4_b33_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b33_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b33_$tag.json)"; }
run s1_2 A=1
run s1_3 SBA_WGRAD_S1=3
run s1_2b A=1
run s1_3b SBA_WGRAD_S1=3
