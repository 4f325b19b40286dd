#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel (and per launch shape) time over the
steady-state steps of bench.py.  Steps are delimited by the fused-Adam prepare kernel
(4 launches per 3-stage step: D64, D128, D256, G).

    python tools/prof_summary.py gpurun_out/prof/x_kernel_trace.csv [steps_to_keep] > profiles/rNN_step.txt
"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('unsigned short', 'bf16')
    return name[:86]


def main():
    path = sys.argv[1]
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'adam_prepare_kernel' in r['Kernel_Name']]
    per_step = 4
    nsteps = len(marks) // per_step
    if nsteps <= keep:
        keep = max(1, nsteps - 1)
    # window: from the end of step (nsteps-keep) to the end of the last step (G's adam_step follows its prepare)
    start = marks[(nsteps - keep) * per_step - 1] + 2 if nsteps > keep else 0
    end = marks[nsteps * per_step - 1] + 2
    win = rows[start:end]
    t0, t1 = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
    by_name = collections.defaultdict(lambda: [0, 0.0])
    by_shape = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for r in win:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        n = short(r['Kernel_Name'])
        by_name[n][0] += 1
        by_name[n][1] += d
        g = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        by_shape[(n, g)][0] += 1
        by_shape[(n, g)][1] += d
        busy += d
    print('# %s' % path)
    print('# steady-state window: %d steps, %d dispatches, wall %.3f ms/step, kernel-busy %.3f ms/step'
          % (keep, len(win), (t1 - t0) / 1e6 / keep, busy / 1e3 / keep))
    print('\n## per kernel (per step)')
    print('%-88s %7s %10s %9s %6s' % ('kernel', 'calls', 'ms/step', 'avg_us', '%'))
    for n, (c, d) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%-88s %7.1f %10.3f %9.1f %6.1f' % (n, c / keep, d / 1e3 / keep, d / c, 100 * d / busy))
    # wall-time attribution: every instant of the window is split equally among the kernels
    # running at that instant (idle gaps are reported separately) -- this is what a kernel costs
    # the step when streams overlap
    ev = []
    for i, r in enumerate(win):
        ev.append((int(r['Start_Timestamp']), 1, i))
        ev.append((int(r['End_Timestamp']), 0, i))
    ev.sort()
    active = set()
    share = collections.defaultdict(float)
    idle = 0.0
    conc_hist = collections.defaultdict(float)
    prev = ev[0][0]
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            if active:
                for j in active:
                    share[short(win[j]['Kernel_Name'])] += dt / len(active)
            else:
                idle += dt
            conc_hist[min(len(active), 8)] += dt
        prev = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    print('\n## wall-time attribution (ms/step; time split equally among concurrently running kernels)')
    print('idle (no kernel running): %.3f ms/step' % (idle / 1e6 / keep))
    print('concurrency histogram (ms/step): ' + ', '.join('%d%s: %.2f' % (k, '+' if k == 8 else '', v / 1e6 / keep)
                                                            for k, v in sorted(conc_hist.items())))
    for n, d in sorted(share.items(), key=lambda kv: -kv[1])[:30]:
        print('%-88s %10.3f' % (n, d / 1e6 / keep))
    print('\n## per kernel + launch grid (blocks), top 40 (per step)')
    for (n, g), (c, d) in sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%-70s %-18s %6.1f %9.3f %9.1f' % (n[:70], str(g), c / keep, d / 1e3 / keep, d / c))


if __name__ == '__main__':
    main()
