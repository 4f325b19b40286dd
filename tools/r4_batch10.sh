#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "register_weight_halo" 2>&1 | tail -15
bash tools/prof_step.sh r4step_v3
