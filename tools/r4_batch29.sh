#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b29_$tag.json 2> gpurun_out/r4_b29_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b29_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b29_$tag.json)"; }
run s2_on A=1
run s2_off SBA_WGRAD_S2=0
run s2_on2 A=1
run s2_off2 SBA_WGRAD_S2=0
