#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b23_$tag.json 2> gpurun_out/r4_b23_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b23_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b23_$tag.json)"; }
run base A=1
run fused SBA_SPLITK_FUSED=1
run base2 A=1
run fused2 SBA_SPLITK_FUSED=1
