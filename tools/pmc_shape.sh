#!/bin/bash
# PMC passes (rocprofv3, one counter group per run, --kernel-trace only) over tools/bench_shape.py for one conv shape.
# usage: tools/pmc_shape.sh <out_prefix> N H W Cin Cout KH KW
cd "$(dirname "$0")/.."
ROOT=$PWD
out=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python3 $ROOT/tools/bench_shape.py "$@" > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$grp" <<'PY' >> $ROOT/gpurun_out/$out.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'igemm' not in k and 'halo' not in k and 'wgrad' not in k: continue
    acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print('    %-34s mean %.4g over %d dispatches' % (c, sum(v) / len(v), len(v)))
PY
  else
    echo "no counter csv for $grp" >> $ROOT/gpurun_out/$out.txt; tail -3 /tmp/pmc_$tag.log >> $ROOT/gpurun_out/$out.txt
  fi
done
