#!/bin/bash
# image heads of the 64 / 128 px images on their discriminators' streams (SBA_FORK_HEADS): same-box A/B + the tests that guard it
set -o pipefail
mkdir -p gpurun_out
for r in 1 2; do
  for v in 1 0; do
    SBA_FORK_HEADS=$v timeout -k 10 300 python bench.py --child --steps 60 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b83_heads${v}_$r.json 2> gpurun_out/r4_b83_heads${v}_$r.err || exit 1
    echo "SBA_FORK_HEADS=$v  $r: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b83_heads${v}_$r.json) $(grep -o '"launch": "[a-z0-9-]*"' gpurun_out/r4_b83_heads${v}_$r.json)"
  done
done
timeout -k 10 600 python -m pytest tests/test_determinism_gpu.py tests/test_stream_audit_gpu.py -x -q -m gpu 2>&1 | tail -n 5
