#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_step_gpu.py -x -q -k "b20_other_variants" 2>&1 | tail -15
for t in 0 128 192 256 384; do echo "SBA_ENC_GROUP_T7_MIN=$t"; SBA_ENC_GROUP_T7_MIN=$t timeout -k 10 200 python tools/bench_encoder_hip.py 2>&1 | grep -E "fwd\+bwd"; done
timeout -k 10 500 python tools/tune_dgrad4.py > gpurun_out/r4_tune_dgrad4_v2.txt 2>&1; tail -28 gpurun_out/r4_tune_dgrad4_v2.txt; cp sba-gan_amd/sbagan/igemm_table.json gpurun_out/r4_igemm_table_v2.json
