#!/bin/bash
cd "$(dirname "$0")/.."
SBA_REPLAY_PRIO=c:4:1:0.05 SBA_REPLAY_PRIO_VERBOSE=3 timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b36_calib.json 2> gpurun_out/r4_b36_calib.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b36_calib.json
grep "^  node" gpurun_out/r4_b36_calib.err > gpurun_out/r4_nodes_alone_v2.txt
grep "^  cp" gpurun_out/r4_b36_calib.err > gpurun_out/r4_cp_v2.txt
grep "sba_replay_prioritize\|^  stream" gpurun_out/r4_b36_calib.err | cut -c1-220
