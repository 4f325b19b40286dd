#!/bin/bash
cd "$(dirname "$0")/.."
for d in 4 3 6; do echo "SBA_WGRAD_ROW_D=$d"; SBA_WGRAD_ROW_D=$d BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep -E " 3x3up | 3x3 .*M=(81920|327680)|4x4s2  M=(16|40|10|25)"; done
