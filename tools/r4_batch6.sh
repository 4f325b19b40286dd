#!/bin/bash
# round-4 GPU batch 6: committed evidence -- PMC of the time-dominant kernel, step profile, HBM-bound kernels (event time + PMC), default-mode golden spread
cd "$(dirname "$0")/.."
timeout -k 10 500 bash tools/pmc_dominant.sh r04 2>&1 | tail -30
timeout -k 10 300 bash tools/prof_step.sh r4step_v2
timeout -k 10 200 python tools/bench_hbm_kernels.py > gpurun_out/r04_hbm_bound_kernels.txt 2>&1; tail -30 gpurun_out/r04_hbm_bound_kernels.txt
timeout -k 10 300 bash tools/pmc_hbm_kernels.sh r04_pmc_hbm_kernels 2>&1 | tail -30
timeout -k 10 400 python tools/golden_spread.py model_b20 bfloat16 graph 8 default > gpurun_out/r04_golden_spread_default_model_b20.txt 2>&1; tail -22 gpurun_out/r04_golden_spread_default_model_b20.txt
timeout -k 10 200 python tools/golden_spread.py model_b20 bfloat16 graph 1 det > gpurun_out/r04_golden_det_model_b20.txt 2>&1; tail -22 gpurun_out/r04_golden_det_model_b20.txt
