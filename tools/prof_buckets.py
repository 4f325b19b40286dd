#!/usr/bin/env python3
"""0.5 ms buckets of a prof_timeline.py dump: kernel-busy time (concurrency) and the top kernels."""
import collections
import re
import sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)\s+(\S.*?)\s+\((\d+), (\d+), (\d+)\)\s*$', l)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), m.group(4).strip()))
rows.sort()
end = max(s + d for s, d, _ in rows)
B = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
for b in range(int(end // B) + 1):
    lo, hi = b * B, (b + 1) * B
    busy = collections.Counter()
    tot = 0
    for s, d, n in rows:
        o = max(0, min(s + d, hi) - max(s, lo))
        if o > 0:
            busy[n.split('<')[0][:28]] += o
            tot += o
    print('%5.1f-%5.1f ms  busy %4.0f us (%.1fx)  %s' % (lo / 1e3, hi / 1e3, tot, tot / B,
                                                       ', '.join('%s %.0f' % kv for kv in busy.most_common(4))))
