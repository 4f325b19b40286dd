#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b38_$tag.json 2> gpurun_out/r4_b38_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b38_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b38_$tag.json)"; grep "no contention" gpurun_out/r4_b38_$tag.err | cut -c1-160; }
run base SBA_REPLAY_PRIO=c:4:1:0.05
run bucket_adam SBA_BUCKET_ADAM=1 SBA_REPLAY_PRIO=c:4:1:0.05
run wgd SBA_OVERLAP_WGRAD_D=1 SBA_REPLAY_PRIO=c:4:1:0.05
run bucket_adam_s5 SBA_BUCKET_ADAM=1 SBA_REPLAY_STREAMS=5
run base2 A=1
