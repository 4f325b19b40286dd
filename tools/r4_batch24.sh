#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b24_$tag.json 2> gpurun_out/r4_b24_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b24_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b24_$tag.json)"; grep "no contention" gpurun_out/r4_b24_$tag.err | cut -c1-200; }
run base SBA_REPLAY_PRIO=c:4:1:0.05
run realfirst SBA_REAL_FIRST=1 SBA_REPLAY_PRIO=c:4:1:0.05
run realfirst_m0 SBA_REAL_FIRST=1 SBA_D_MERGE=0
run realfirst_s5 SBA_REAL_FIRST=1 SBA_REPLAY_STREAMS=5
run base2 A=1
run realfirst2 SBA_REAL_FIRST=1
