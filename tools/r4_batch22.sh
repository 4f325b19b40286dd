#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b22_$tag.json 2> gpurun_out/r4_b22_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b22_$tag.json)"; grep -A 8 "sba_replay_prioritize: [0-9]* streams" gpurun_out/r4_b22_$tag.err | cut -c1-250; }
run m1 SBA_D_MERGE=1 SBA_REPLAY_PRIO=c:4:1:0.05
run m0 SBA_D_MERGE=0 SBA_REPLAY_PRIO=c:4:1:0.05
run m1p1 SBA_D_MERGE=1 SBA_REPLAY_PRIO=c:4:1:0.05 SBA_REPLAY_POLICY=1
run m1p2 SBA_D_MERGE=1 SBA_REPLAY_PRIO=c:4:1:0.05 SBA_REPLAY_POLICY=2
run m1s5 SBA_D_MERGE=1 SBA_REPLAY_PRIO=c:5:1:0.05
