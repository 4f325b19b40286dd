#!/bin/bash
# A/B of the persistent halo-tile 3x3 kernel against the per-tile one on the generator's big convs (B = 20)
cd "$(dirname "$0")/.."
for v in 1 0; do
  echo "== SBA_CONV_HALO2=$v"
  SBA_CONV_HALO2=$v python tools/bench_conv.py 2>&1 | grep -v amdgpu | tail -12
done
