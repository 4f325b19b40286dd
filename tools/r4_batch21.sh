#!/bin/bash
cd "$(dirname "$0")/.."
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --child --graph 3 --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-roofline > gpurun_out/r4_b21_$tag.json 2> gpurun_out/r4_b21_$tag.err || echo "FAILED $tag"; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_b21_$tag.json)"; }
run merge1 SBA_D_MERGE=1
run merge0 SBA_D_MERGE=0
run merge1b SBA_D_MERGE=1
run merge0b SBA_D_MERGE=0
run merge1_prevtable SBA_D_MERGE=1 SBA_IGEMM_TABLE_FILE=$PWD/tools/_ab/igemm_table_prev.json
run merge1_nostem SBA_D_MERGE=1 SBA_ENC_FRAG_STEM=0
