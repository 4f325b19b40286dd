#!/bin/bash
# last call of the round: the whole GPU suite, the default bench line, and the kernel trace of the benched step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_b86_pytest.txt 2>&1; rc=$?
tail -n 3 gpurun_out/r4_b86_pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r4_b86_bench.json 2> gpurun_out/r4_b86_bench.err || exit 1
tail -n 1 gpurun_out/r4_b86_bench.json | cut -c1-220
timeout -k 10 300 bash tools/prof_step.sh r04_step_final || exit 1
