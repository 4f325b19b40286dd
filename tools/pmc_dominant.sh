#!/bin/bash
# PMC traffic of the time-dominant kernel: bench.py dumps its shapes, two rocprofv3 --pmc passes (one counter each,
# --kernel-trace only), collect -> gpurun_out/<tag>_pmc_dominant_kernel.json.   bash tools/pmc_dominant.sh <tag>
set -e
tag=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --child --graph 0 --steps 2 --warmup 2 --no-cpu-baseline --no-also --dump-shapes $ROOT/gpurun_out/${tag}_dominant_shapes.json > $ROOT/gpurun_out/${tag}_bench_roofline.json 2> $ROOT/gpurun_out/${tag}_bench_roofline.err
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $ROOT/tools/pmc_dominant.py $ROOT/gpurun_out/${tag}_dominant_shapes.json > /tmp/pmc_$c.log 2>&1
done
python3 $ROOT/tools/pmc_dominant.py --collect $ROOT/gpurun_out/${tag}_dominant_shapes.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $ROOT/gpurun_out/${tag}_pmc_dominant_kernel.json
# kernel-trace stats of the same launches: the average duration must agree with bench.py's live events
rm -rf /tmp/pmc_trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_trace -o p -- python3 $ROOT/tools/pmc_dominant.py $ROOT/gpurun_out/${tag}_dominant_shapes.json > /tmp/pmc_trace.log 2>&1
k=$(find /tmp/pmc_trace -name "*kernel_stats.csv" | head -n 1)
[ -n "$k" ] && cp $k $ROOT/gpurun_out/${tag}_dominant_kernel_stats.csv
cat $ROOT/gpurun_out/${tag}_pmc_dominant_kernel.json
tail -n 1 $ROOT/gpurun_out/${tag}_bench_roofline.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['roofline'], indent=1))"
