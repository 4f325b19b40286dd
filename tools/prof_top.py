#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace CSV over the LAST `frac` of the run (steady state): calls, total, average.
    python tools/prof_top.py trace.csv [frac=0.5] [top=40]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*$', '', n).replace('unsigned short', 'bf16')
    return n[:90]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = rows[int(len(rows) * (1 - frac)):]
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
tot, cnt = collections.Counter(), collections.Counter()
for r in rows:
    k = short(r['Kernel_Name']) + '  grid=%s' % r.get('Grid_Size_X', '?')
    if len(sys.argv) > 4:
        k = short(r['Kernel_Name'])
    tot[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    cnt[k] += 1
busy = sum(tot.values())
print('# window %.3f ms wall, %.3f ms kernel-busy, %d dispatches' % ((t1 - t0) / 1e6, busy / 1e6, len(rows)))
for k, v in tot.most_common(top):
    print('%-112s %6d x %8.1f us = %8.3f ms  %4.1f%%' % (k, cnt[k], v / cnt[k] / 1e3, v / 1e6, 100.0 * v / busy))
