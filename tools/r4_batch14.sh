#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b14_$tag.json 2> gpurun_out/r4_b14_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b14_$tag.json)"; }
run newtable A=1
run prevtable SBA_IGEMM_TABLE_FILE=$PWD/tools/_ab/igemm_table_prev.json
run newtable2 A=1
run calib SBA_REPLAY_PRIO=c:4:1:0.05 SBA_REPLAY_PRIO_VERBOSE=3
grep "^  node" gpurun_out/r4_b14_calib.err > gpurun_out/r4_nodes_alone.txt; wc -l gpurun_out/r4_nodes_alone.txt
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "tile" 2>&1 | tail -3
