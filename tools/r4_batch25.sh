#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default_v3.json 2> gpurun_out/r4_bench_default_v3.err; tail -n 1 gpurun_out/r4_bench_default_v3.json | cut -c1-300
bash tools/prof_step.sh r4step_v5
