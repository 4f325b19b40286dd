#!/bin/bash
# calibration report (every launch alone, longest path, stream composition) of the single-GPU and the one-rank data-parallel recording
set -o pipefail
mkdir -p gpurun_out
export SBA_REPLAY_PRIO=c:4:1:0.05 SBA_REPLAY_PRIO_VERBOSE=2
timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b84_single.json 2> gpurun_out/r4_b84_single.err || exit 1
SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=4 timeout -k 10 300 python bench.py --child --steps 10 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b84_dist1.json 2> gpurun_out/r4_b84_dist1.err || exit 1
grep -c . gpurun_out/r4_b84_single.err gpurun_out/r4_b84_dist1.err
