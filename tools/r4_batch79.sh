#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests/test_determinism_gpu.py tests/test_stream_audit_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b79_$tag.json 2> gpurun_out/r4_b79_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b79_$tag.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b79_$tag.json)"; }
for i in 1 2 3; do run side_$i A=1; run main_$i SBA_LOG_STREAM=0; done
