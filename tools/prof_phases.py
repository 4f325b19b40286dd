#!/usr/bin/env python3
"""Per-phase kernel table from a rocprofv3 --kernel-trace of tools/phase_times.py: the run replays each phase
graph alone 10 times, groups separated by a torch.cumsum marker kernel.

    python tools/prof_phases.py <..._kernel_trace.csv> > profiles/rNN_phases.txt
"""
import collections
import csv
import re
import sys

NAMES = ['A  generator forward', 'D0 update of D_NET64', 'D1 update of D_NET128', 'D2 update of D_NET256',
         'B  generator loss + backward + Adam', 'whole step']


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*$', '', n).replace('unsigned short', 'bf16')
    return n[:78]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'scan' in r['Kernel_Name'].lower() or 'cumsum' in r['Kernel_Name'].lower()]
    # the last len(NAMES)+1 markers delimit the timed groups (the final group = wall loop has no marker before it)
    marks = marks[-len(NAMES):]
    bounds = marks + [len(rows)]
    reps = 10
    for gi, name in enumerate(NAMES):
        win = rows[bounds[gi] + 1:bounds[gi + 1]]
        if gi == len(NAMES) - 1:
            win = win[:len(win) // 2]       # whole-step group is followed by the 10 wall-clock replays
        if not win:
            continue
        t0, t1 = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
        by = collections.defaultdict(lambda: [0, 0.0])
        for r in win:
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            k = short(r['Kernel_Name'])
            by[k][0] += 1
            by[k][1] += d
        busy = sum(v[1] for v in by.values())
        print('## phase %s: wall %.3f ms, kernel-busy %.3f ms, %.0f launches per replay'
              % (name, (t1 - t0) / 1e6 / reps, busy / 1e3 / reps, len(win) / reps))
        for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
            print('  %-80s %6.1f x %7.1f us = %7.3f ms' % (k, v[0] / reps, v[1] / v[0], v[1] / 1e3 / reps))
        print()


if __name__ == '__main__':
    main()
