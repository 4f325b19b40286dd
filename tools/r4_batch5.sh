#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -6
echo "== frag weights ON"; ONLY=G timeout -k 10 200 python tools/bench_conv.py 2>&1 | tail -12
echo "== frag weights OFF"; SBA_FRAG_WEIGHTS=0 ONLY=G timeout -k 10 200 python tools/bench_conv.py 2>&1 | tail -12
timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_single_v3.json 2> gpurun_out/r4_single_v3.err; grep "launch probe" gpurun_out/r4_single_v3.err; tail -n 1 gpurun_out/r4_single_v3.json | cut -c1-160
SBA_FRAG_WEIGHTS=0 timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_single_v3_nofrag.json 2> gpurun_out/r4_single_v3_nofrag.err; grep "launch probe" gpurun_out/r4_single_v3_nofrag.err; tail -n 1 gpurun_out/r4_single_v3_nofrag.json | cut -c1-160
