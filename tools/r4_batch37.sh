#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b37_$tag.json 2> gpurun_out/r4_b37_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b37_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b37_$tag.json)"; }
run base A=1
run early_zero SBA_EARLY_ZERO=1
run merge0 SBA_D_MERGE=0
run base2 A=1
run early_zero2 SBA_EARLY_ZERO=1
run s5 SBA_REPLAY_STREAMS=5
