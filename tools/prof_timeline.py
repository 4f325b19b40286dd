#!/usr/bin/env python3
"""Timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV: per HIP stream (queue)
the kernels in launch order with start offset, duration and the gap since the previous kernel of
the same stream; plus per-stream busy/idle totals.  Steps are delimited like prof_summary.py.

    python tools/prof_timeline.py trace.csv [step_from_end=1] > timeline.txt
"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    return name.replace('unsigned short', 'bf16')[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'adam_prepare_kernel' in r['Kernel_Name']]
    n = len(marks) // 4
    lo = marks[(n - back) * 4 - 1] + 2
    hi = marks[(n - back + 1) * 4 - 1] + 2
    win = rows[lo:hi]
    t0 = int(win[0]['Start_Timestamp'])
    qkey = sys.argv[3] if len(sys.argv) > 3 else ('Stream_Id' if 'Stream_Id' in win[0] and len(set(r['Stream_Id'] for r in win)) > 1 else 'Queue_Id')
    by_q = collections.defaultdict(list)
    for r in win:
        by_q[r[qkey]].append(r)
    print('# step window %.3f ms, %d dispatches, %d streams (%s)' % (
        (int(win[-1]['End_Timestamp']) - t0) / 1e6, len(win), len(by_q), qkey))
    for q, rs in sorted(by_q.items(), key=lambda kv: int(kv[1][0]['Start_Timestamp'])):
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
        span = (int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])) / 1e3
        print('\n## stream %s: %d kernels, busy %.1f us over a span of %.1f us (first at %.1f us)' % (
            q, len(rs), busy, span, (int(rs[0]['Start_Timestamp']) - t0) / 1e3))
        prev = None
        for r in rs:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            gap = (s - prev) / 1e3 if prev is not None else 0.0
            g = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
            print('%9.1f %8.1f %8.1f  %-60s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(r['Kernel_Name']), g))
            prev = e


if __name__ == '__main__':
    main()
