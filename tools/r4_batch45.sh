#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_step_gpu.py tests/test_damsm_gpu.py -x -q -k "image_encoder_hip or damsm_update" 2>&1 | tail -4
for c in 1 0; do echo "SBA_ENC_POOL_COMMUTE=$c"; SBA_ENC_POOL_COMMUTE=$c timeout -k 10 200 python tools/bench_encoder_hip.py 2>&1 | grep -E "fwd\+bwd"; done
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b45_$tag.json 2> gpurun_out/r4_b45_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b45_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b45_$tag.json)"; }
run commute1 A=1
run commute0 SBA_ENC_POOL_COMMUTE=0
run commute1b A=1
run commute0b SBA_ENC_POOL_COMMUTE=0
