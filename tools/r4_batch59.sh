#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b59_$tag.json 2> gpurun_out/r4_b59_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b59_$tag.json)"; }
for i in 1 2 3; do run base_$i A=1; run notickets_$i SBA_LIB_PATH=$PWD/tools/_ab/libsbagan_notickets.so; done
