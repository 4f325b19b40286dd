#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "resblock_batchnorm_sums or conv_bn_act_blocks or conv_fwd_dgrad_wgrad" 2>&1 | tail -12
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b47_$tag.json 2> gpurun_out/r4_b47_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b47_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b47_$tag.json)"; }
run fuse1 A=1
run fuse0 SBA_FUSE_BN_RED=0
run fuse1b A=1
run fuse0b SBA_FUSE_BN_RED=0
