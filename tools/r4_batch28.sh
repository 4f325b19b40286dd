#!/bin/bash
cd "$(dirname "$0")/.."
for s2 in 0 1; do echo "SBA_WGRAD_S2=$s2"; SBA_WGRAD_S2=$s2 BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep "4x4s2"; done
for w in 256 384 768 1024; do echo "SBA_WGRAD_S2_WGS=$w"; SBA_WGRAD_S2_WGS=$w BENCH_FIRST_WRITE=1 timeout -k 10 200 python tools/bench_wgrad.py 2>&1 | grep "4x4s2"; done
