#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_fwd_dgrad_wgrad and (case6 or case7 or case10 or case25 or case26 or case27 or case28)" 2>&1 | tail -12
