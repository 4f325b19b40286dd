#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default_v4.json 2> gpurun_out/r4_bench_default_v4.err; tail -n 1 gpurun_out/r4_bench_default_v4.json | cut -c1-200
