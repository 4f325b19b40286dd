#!/usr/bin/env python3
"""DAMSM words loss at the benched shape (B = 20 captions x 20 images, 17x17 regions, nef 256, 18 words): forward +
backward repeated; run under `rocprofv3 --kernel-trace --stats` for the per-kernel times.
python tools/bench_damsm.py [--steps 30]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 30
    from miscc import losses
    dev = torch.device('cuda:0')
    B, nef, L = 20, 256, 18
    torch.manual_seed(1)
    feat = torch.randn(B, nef, 17, 17, device=dev, requires_grad=True)
    words = torch.randn(B, nef, L, device=dev)
    lens = torch.tensor([18] * 4 + [15] * 6 + [12] * 6 + [9] * 4, device=dev)
    labels = torch.arange(B, device=dev)
    cids = np.arange(B)
    from torch.profiler import profile, ProfilerActivity
    for it in range(5):
        w0, w1, _ = losses.words_loss(feat, words, labels, lens, cids, B)
        (w0 + w1).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for it in range(steps):
            w0, w1, _ = losses.words_loss(feat, words, labels, lens, cids, B)
            (w0 + w1).backward()
        torch.cuda.synchronize()
    print('lib %s' % os.environ.get('SBA_LIB_PATH', '(in-tree)'))
    for ev in prof.key_averages():
        if 'damsm' in ev.key:
            print('%-40s %4d launches  %8.1f us avg' % (ev.key.split('(')[0][-40:], ev.count, ev.device_time_total / ev.count))


if __name__ == '__main__':
    main()
