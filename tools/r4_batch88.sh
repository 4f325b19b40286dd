#!/bin/bash
# the whole GPU suite on the last commit of the round
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_b88_pytest.txt 2>&1; rc=$?
tail -n 3 gpurun_out/r4_b88_pytest.txt
exit $rc
