#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -k "resize_backward or stem_backward_matrix or image_encoder_hip" 2>&1 | tail -6
for m in 1 0; do echo "SBA_RESIZE_BWD_TAB=$m"; SBA_RESIZE_BWD_TAB=$m timeout -k 10 200 python tools/bench_encoder_hip.py 2>&1 | grep -E "graph"; done
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b61_$tag.json 2> gpurun_out/r4_b61_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b61_$tag.json)"; }
for i in 1 2; do run tab_$i A=1; run notab_$i SBA_RESIZE_BWD_TAB=0; done
