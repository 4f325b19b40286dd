#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py -x -q -k "dense_layers or init_stage or generator_forward or relaxations" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b78_$tag.json 2> gpurun_out/r4_b78_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b78_$tag.json)"; }
for i in 1 2 3; do run new_$i A=1; run prev_$i SBA_LIB_PATH=$PWD/tools/_ab/libsbagan_prev.so; done
