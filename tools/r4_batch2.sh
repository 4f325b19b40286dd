#!/bin/bash
# round-4 GPU batch 2: stream audit + stress, two-rank data-parallel test (single recording with host-call exchanges), one-rank cost of mode 4
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests/test_stream_audit_gpu.py -x -q 2>&1 | tail -25
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -15
for m in 4; do
  SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=$m timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_dist1_mode$m.json 2> gpurun_out/r4_dist1_mode$m.err
  grep -E "launch probe|unavailable|sba_replay" gpurun_out/r4_dist1_mode$m.err; tail -n 1 gpurun_out/r4_dist1_mode$m.json | cut -c1-160
done
