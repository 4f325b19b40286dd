#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b80_$tag.json 2> gpurun_out/r4_b80_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b80_$tag.json)"; }
for i in 1 2; do run base_$i A=1; run merge0_$i SBA_D_MERGE=0; run s5_$i SBA_REPLAY_STREAMS=5; run merge0_s5_$i SBA_D_MERGE=0 SBA_REPLAY_STREAMS=5; done
