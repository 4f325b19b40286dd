#!/bin/bash
# round-4 GPU batch 3: the whole GPU suite, the default bench line, PMC of the halo-tile conv
cd "$(dirname "$0")/.."
timeout -k 10 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout -k 10 400 python bench.py > gpurun_out/r4_bench_v2.json 2> gpurun_out/r4_bench_v2.err; grep "launch probe" gpurun_out/r4_bench_v2.err | head -2; tail -n 1 gpurun_out/r4_bench_v2.json | cut -c1-200
rm -f gpurun_out/r4_pmc_halo_res128.txt
timeout -k 10 600 bash tools/pmc_shape.sh r4_pmc_halo_res128 20 128 128 64 64 3 3
cat gpurun_out/r4_pmc_halo_res128.txt
