#!/bin/bash
# rocprofv3 kernel trace of the benched command (child process of bench.py, all launch modes probed) -> per-kernel summary.
#   bash tools/prof_step.sh <tag>   writes gpurun_out/<tag>_summary.txt, <tag>_kernel_stats.csv, <tag>_bench.json
set -e
tag=${1:-r3step}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $ROOT/bench.py --child --steps 6 --warmup 4 --no-cpu-baseline --no-also --no-roofline > $ROOT/gpurun_out/${tag}_bench.json 2> $ROOT/gpurun_out/${tag}_bench.err
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -n 1)
python3 $ROOT/tools/prof_summary.py $f 3 > $ROOT/gpurun_out/${tag}_summary.txt
python3 $ROOT/tools/prof_timeline.py $f > $ROOT/gpurun_out/${tag}_timeline.txt
k=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -n 1)
[ -n "$k" ] && cp $k $ROOT/gpurun_out/${tag}_kernel_stats.csv
tail -n 1 $ROOT/gpurun_out/${tag}_bench.json | cut -c1-200
