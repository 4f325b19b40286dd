#!/bin/bash
# A/B of environment knobs on the benched step: each argument is one "NAME=VALUE[,NAME=VALUE...]" setting ("base" = none);
# prints ms per step (short child runs of bench.py, all launch modes probed).   bash tools/ab_env.sh base SBA_SPLITK_FUSED=1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for setting in "$@"; do
    envs=()
    if [ "$setting" != "base" ]; then IFS=',' read -ra envs <<< "$setting"; fi
    out=$(env "${envs[@]}" python3 $ROOT/bench.py --child --steps 12 --warmup 4 --no-cpu-baseline --no-also --no-roofline 2>/dev/null | tail -n 1)
    echo "$setting: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("launch"))')"
done
