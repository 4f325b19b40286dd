#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default_v6.json 2> gpurun_out/r4_bench_default_v6.err; tail -n 1 gpurun_out/r4_bench_default_v6.json | cut -c1-170
bash tools/prof_step.sh r4step_v7 | tail -1 | cut -c1-160
bash tools/r4_batch36.sh | tail -9 | cut -c1-200
