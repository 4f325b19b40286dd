#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b15_$tag.json 2> gpurun_out/r4_b15_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b15_$tag.json)"; }
run base A=1
run merge SBA_D_MERGE=1
run merge_s3 SBA_D_MERGE=1 SBA_REPLAY_STREAMS=3
run merge_s5 SBA_D_MERGE=1 SBA_REPLAY_STREAMS=5
run base2 A=1
