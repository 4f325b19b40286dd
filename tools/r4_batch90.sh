#!/bin/bash
# data-parallel single recording, one rank: generator exchange deferred behind the next step's real-image forwards (default,
# two-pass discriminator loss) against exposed where it is issued (SBA_DP_OVERLAP_G=0, grouped real|fake pass)
set -o pipefail
mkdir -p gpurun_out
for r in 1; do
  for v in 1 0; do
    SBA_DP_OVERLAP_G=$v SBA_BENCH_FORCE_DIST=1 SBA_DP_REPLAY=4 timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b90_og${v}_$r.json 2> gpurun_out/r4_b90_og${v}_$r.err || { tail -n 20 gpurun_out/r4_b90_og${v}_$r.err; exit 1; }
    echo "SBA_DP_OVERLAP_G=$v  $r: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b90_og${v}_$r.json) $(grep -o '"launch": "[a-z0-9-]*"' gpurun_out/r4_b90_og${v}_$r.json) $(grep -o 'losses_finite[^,}]*' gpurun_out/r4_b90_og${v}_$r.json) $(grep -o 'sba_replay: [0-9]* nodes' gpurun_out/r4_b90_og${v}_$r.err)"
  done
done
timeout -k 10 300 python bench.py --child --steps 40 --warmup 6 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b90_single.json 2> gpurun_out/r4_b90_single.err || exit 1
echo "single GPU: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b90_single.json)"
