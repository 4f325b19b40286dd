#!/bin/bash
# stream priorities for the longest path: A/B over SBA_REPLAY_PRIO = mode:streams:n_high:slack
cd "$(dirname "$0")/.."
first=1
for spec in 0 1:6:2:0.08 2:6:2:0.08 3:6:2:0.08 2:5:1:0.04 2:6:2:0.2 2:8:3:0.1 1:8:4:0.3; do
  v=1; [ $first = 1 ] || true
  [ "$spec" = "2:6:2:0.08" ] && v=2
  SBA_REPLAY_PRIO=$spec SBA_REPLAY_PRIO_VERBOSE=$v timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_prio_$spec.json 2> gpurun_out/r4_prio_$spec.err || echo "FAILED $spec"
  echo "SBA_REPLAY_PRIO=$spec: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_prio_$spec.json)"
  grep "sba_replay_prioritize" gpurun_out/r4_prio_$spec.err | head -3
done
