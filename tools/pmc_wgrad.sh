#!/bin/bash
# PMC passes (rocprofv3, one counter group per run, --kernel-trace only) over ONE weight-gradient shape of tools/bench_wgrad.py
# under the environment given:   SBA_WGRAD_S2=0 bash tools/pmc_wgrad.sh <out_prefix> "<substring of the shape's name>"
cd "$(dirname "$0")/.."
ROOT=$PWD
out=$1; export BENCH_WGRAD_ONLY="$2"; export BENCH_FIRST_WRITE=1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmcw_$tag
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcw_$tag -o p -- python3 $ROOT/tools/bench_wgrad.py > /tmp/pmcw_$tag.log 2>&1
  f=$(find /tmp/pmcw_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$grp" <<'PY' >> $ROOT/gpurun_out/$out.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'wgrad' not in k: continue
    acc[k[:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print('    %-34s mean %.4g over %d dispatches' % (c, sum(v) / len(v), len(v)))
PY
  else
    echo "no counter csv for $grp" >> $ROOT/gpurun_out/$out.txt; tail -3 /tmp/pmcw_$tag.log >> $ROOT/gpurun_out/$out.txt
  fi
done
