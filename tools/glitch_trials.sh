#!/bin/bash
# usage: glitch_trials.sh <label> <trials> [debug_finite args...]  (env toggles inherited)
label=$1; n=$2; shift 2
for t in $(seq 1 $n); do
  REPLAY=${REPLAY:-sync} N_EAGER=5 N_STEPS=8 timeout -k 10 120 python tools/debug_finite.py --no-cpu-baseline --no-roofline "$@" > gpurun_out/gl_${label}_$t.log 2>&1
  a=$(grep "^5 " gpurun_out/gl_${label}_$t.log | sed 's/.*errG_total=\([^ ]*\).*/\1/')
  b=$(grep "^6 " gpurun_out/gl_${label}_$t.log | sed 's/.*errG_total=\([^ ]*\).*/\1/')
  c=$(grep "^7 " gpurun_out/gl_${label}_$t.log | sed 's/.*errG_total=\([^ ]*\).*/\1/')
  echo "$label trial $t: errG_total step5=$a step6=$b step7=$c"
done
