#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py -x -q -k "damsm or generator_loss or relaxations or reproducible_in_every" 2>&1 | tail -4
timeout -k 10 300 python bench.py --child --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_single_v4.json 2> gpurun_out/r4_single_v4.err; grep "launch probe" gpurun_out/r4_single_v4.err; tail -n 1 gpurun_out/r4_single_v4.json | cut -c1-160
