#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b16_$tag.json 2> gpurun_out/r4_b16_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b16_$tag.json)"; }
run merge SBA_D_MERGE=1
run merge_noearly SBA_D_MERGE=1 SBA_EARLY_D=0
run merge_pol1 SBA_D_MERGE=1 SBA_REPLAY_POLICY=1
run merge_pol2 SBA_D_MERGE=1 SBA_REPLAY_POLICY=2
run merge_wgd SBA_D_MERGE=1 SBA_OVERLAP_WGRAD_D=1
run merge_wgd_s5 SBA_D_MERGE=1 SBA_OVERLAP_WGRAD_D=1 SBA_REPLAY_STREAMS=5
run merge_nofork SBA_D_MERGE=1 SBA_FORK_MAPPING=0
run merge2 SBA_D_MERGE=1
