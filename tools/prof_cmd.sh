#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool -> top-kernel table.   bash tools/prof_cmd.sh <tag> <frac> <script> [args]
set -e
tag=$1; frac=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o p -- python3 $ROOT/"$@" > $ROOT/gpurun_out/${tag}_run.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -n 1)
python3 $ROOT/tools/prof_top.py $f $frac 60 > $ROOT/gpurun_out/${tag}_top.txt
python3 $ROOT/tools/prof_top.py $f $frac 40 nogrid > $ROOT/gpurun_out/${tag}_top_nogrid.txt
tail -n 3 $ROOT/gpurun_out/${tag}_run.log
