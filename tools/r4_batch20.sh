#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b20_$tag.json 2> gpurun_out/r4_b20_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b20_$tag.json)"; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -k "stem or maxpool or image_encoder" 2>&1 | tail -3
timeout -k 10 200 python tools/bench_encoder_hip.py 2>&1 | grep -E "fwd\+bwd"
run a A=1
run b A=1
run c A=1
