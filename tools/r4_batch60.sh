#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -k "stem_backward_matrix or image_encoder_hip" 2>&1 | tail -6
for m in 1 0; do echo "SBA_ENC_STEM_BWD_MFMA=$m"; SBA_ENC_STEM_BWD_MFMA=$m timeout -k 10 200 python tools/bench_encoder_hip.py 2>&1 | grep -E "graph"; done
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b60_$tag.json 2> gpurun_out/r4_b60_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b60_$tag.json)"; }
for i in 1 2; do run mfma_$i A=1; run valu_$i SBA_ENC_STEM_BWD_MFMA=0; done
