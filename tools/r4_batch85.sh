#!/bin/bash
# data-parallel single recording: the deferred generator update as recorded launches (SBA_DP_RECORD_UPDATE=1) against the
# eager update issued from the host-call node (=0): two-rank bit-equality tests, then one rank (RCCL, all host calls live) A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -n 6 || exit 1
for r in 1 2; do
  for v in 1 0; do
    SBA_DP_RECORD_UPDATE=$v SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=4 timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b85_rec${v}_$r.json 2> gpurun_out/r4_b85_rec${v}_$r.err || exit 1
    echo "SBA_DP_RECORD_UPDATE=$v  $r: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b85_rec${v}_$r.json) $(grep -o '"launch": "[a-z0-9-]*"' gpurun_out/r4_b85_rec${v}_$r.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b85_rec${v}_$r.json)"
  done
done
timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b85_single.json 2> gpurun_out/r4_b85_single.err || exit 1
echo "single GPU: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b85_single.json)"
