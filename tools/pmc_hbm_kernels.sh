#!/bin/bash
# HBM traffic of the HBM-bound kernels from the PMC counters (rocprofv3, one counter per pass, --kernel-trace only) over
# tools/bench_hbm_kernels.py; corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
# bytes = 2 x FETCH_SIZE (KiB) x 1024 + WRITE_SIZE (KiB) x 1024.  usage: tools/pmc_hbm_kernels.sh <out_name>
cd "$(dirname "$0")/.."
ROOT=$PWD
out=${1:-pmc_hbm_kernels}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_hbm_$grp
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_hbm_$grp -o p -- python3 $ROOT/tools/bench_hbm_kernels.py > /tmp/pmc_hbm_$grp.log 2>&1
done
python3 - "$ROOT/gpurun_out/$out.txt" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for grp in ('FETCH_SIZE', 'WRITE_SIZE'):
    fs = glob.glob('/tmp/pmc_hbm_%s/**/*counter_collection.csv' % grp, recursive=True)
    if not fs:
        print('no counter csv for', grp); continue
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if not any(s in k for s in ('img_head', 'd_stem', 'word_attn', 'adam_step')):
            continue
        name = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].replace('unsigned short', 'bf16')[:48]
        grid = r.get('Grid_Size') or r.get('Grid_Size_X') or '?'
        acc[(name, grid)][r['Counter_Name']].append(float(r['Counter_Value']))
with open(sys.argv[1], 'w') as f:
    f.write('# HBM traffic per launch from rocprofv3 PMC (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; gfx950 correction of\n'
            '# MI355X_MICROARCH.md); grid = total work-items of the launch.  Compare with the algorithmic bytes of\n'
            '# profiles/r02_hbm_bound_kernels.txt\n')
    f.write('%-50s %12s %10s %10s %10s %6s\n' % ('kernel', 'grid', 'fetch MB', 'write MB', 'total MB', 'n'))
    for (name, grid), d in sorted(acc.items()):
        fe = d.get('FETCH_SIZE', [0.0]); wr = d.get('WRITE_SIZE', [0.0])
        fmb = 2 * sum(fe) / len(fe) * 1024 / 1e6
        wmb = sum(wr) / len(wr) * 1024 / 1e6
        f.write('%-50s %12s %10.1f %10.1f %10.1f %6d\n' % (name, grid, fmb, wmb, fmb + wmb, len(fe)))
print(open(sys.argv[1]).read())
PY
