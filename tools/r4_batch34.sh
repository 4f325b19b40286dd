#!/bin/bash
cd "$(dirname "$0")/.."
SBA_WGRAD_ROW_OW4=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_fwd_dgrad_wgrad" 2>&1 | tail -4
for o in 0 1; do echo "SBA_WGRAD_ROW_OW4=$o"; SBA_WGRAD_ROW_OW4=$o BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep -E "M=(320|304|640) "; done
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b34_$tag.json 2> gpurun_out/r4_b34_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b34_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b34_$tag.json)"; }
run ow4_0 A=1
run ow4_1 SBA_WGRAD_ROW_OW4=1
run ow4_0b A=1
run ow4_1b SBA_WGRAD_ROW_OW4=1
