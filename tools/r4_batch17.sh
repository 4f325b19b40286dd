#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b17_$tag.json 2> gpurun_out/r4_b17_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b17_$tag.json)  $(grep 'sba_replay_prioritize' gpurun_out/r4_b17_$tag.err | head -1 | cut -c1-200)"; }
run merge SBA_D_MERGE=1
run mask50_s10 SBA_D_MERGE=1 SBA_REPLAY_PRIO=4:5:3:0.10
run mask50_s15 SBA_D_MERGE=1 SBA_REPLAY_PRIO=4:5:3:0.15
run mask75_s10 SBA_D_MERGE=1 SBA_REPLAY_PRIO=4:5:3:0.10 SBA_REPLAY_CUMASK=77777777
run mask75_s05 SBA_D_MERGE=1 SBA_REPLAY_PRIO=4:5:3:0.05 SBA_REPLAY_CUMASK=77777777
run nomerge_mask75_s10 SBA_REPLAY_PRIO=4:5:3:0.10 SBA_REPLAY_CUMASK=77777777
run maskff_s10 SBA_D_MERGE=1 SBA_REPLAY_PRIO=4:5:3:0.10 SBA_REPLAY_CUMASK=ffffffff
