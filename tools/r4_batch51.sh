#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b51_$tag.json 2> gpurun_out/r4_b51_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b51_$tag.json)"; }
for i in 1 2; do run new_$i A=1; run old_$i SBA_LIB_PATH=$PWD/tools/_ab/libsbagan_old_epilogue.so; done
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_bn_act_blocks or conv_fwd_dgrad_wgrad" 2>&1 | tail -3
