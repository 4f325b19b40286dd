#!/bin/bash
# rocprofv3 kernel trace of tools/phase_times.py (every phase graph replayed alone) -> per-phase kernel tables.
#   bash tools/prof_phases.sh <tag>      writes gpurun_out/<tag>_phases.txt and gpurun_out/<tag>_phase_times.log
set -e
tag=${1:-r3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o ph -- python3 $ROOT/tools/phase_times.py > $ROOT/gpurun_out/${tag}_phase_times.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -n 1)
python3 $ROOT/tools/prof_phases.py $f > $ROOT/gpurun_out/${tag}_phases.txt
grep -E "^phase|^whole" $ROOT/gpurun_out/${tag}_phase_times.log
