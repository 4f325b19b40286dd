#!/bin/bash
# hipGraph runtime knobs vs step time (ROCm 7.2): does multi-queue graph execution recover branch concurrency?
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --child --graph 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E '"ms_per_step"|failed|Error' | sed -E 's/.*"ms_per_step": ([0-9.]+).*"launch": "([a-z]+)".*/   ms_per_step \1 launch \2/'; }
run SBA_X=0
run SBA_GRAPH_SINGLE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 SBA_GRAPH_SINGLE=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4 SBA_GRAPH_SINGLE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 SBA_GRAPH_SINGLE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 SBA_GRAPH_SINGLE=1
run GPU_MAX_HW_QUEUES=8 SBA_GRAPH_SINGLE=1
run SBA_IGEMM_DMA=0
