#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b32_$tag.json 2> gpurun_out/r4_b32_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b32_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b32_$tag.json)"; }
run s1_2 A=1
run s1_0 SBA_WGRAD_S1=0
run s1_1 SBA_WGRAD_S1=1
run s1_2b A=1
run s1_0b SBA_WGRAD_S1=0
