#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or inception" 2>&1 | tail -6
timeout -k 10 400 python -m pytest tests/test_step_gpu.py -x -q -k "image_encoder_hip" 2>&1 | tail -4
for f in 1 0; do echo "SBA_ENC_FRAG_STEM=$f"; SBA_ENC_FRAG_STEM=$f timeout -k 10 200 python tools/bench_encoder_hip.py 2>&1 | grep -E "fwd\+bwd"; done
timeout -k 10 300 python bench.py --child --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_single_v5.json 2> gpurun_out/r4_single_v5.err; grep "launch probe" gpurun_out/r4_single_v5.err; tail -n 1 gpurun_out/r4_single_v5.json | cut -c1-160
SBA_ENC_FRAG_STEM=0 timeout -k 10 300 python bench.py --child --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_single_v5_nostem.json 2> gpurun_out/r4_single_v5_nostem.err; grep "launch probe" gpurun_out/r4_single_v5_nostem.err; tail -n 1 gpurun_out/r4_single_v5_nostem.json | cut -c1-160
