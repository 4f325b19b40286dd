#!/usr/bin/env python3
"""Measure (tile, K split) of the GROUPED launch of the four parity classes of the 4x4 / stride-2 data gradient
(ops.conv_dgrad, kind '4x4s2'; sba_conv_igemm_group_splitk) on the discriminators' down-block shapes at B (real | fake
pass: N = 2B, generator-term pass: N = B), next to the four single launches it replaces, and write the fastest plan per
shape to the 'dgrad4' section of sba-gan_amd/sbagan/igemm_table.json.  Run on an MI355X: `python tools/tune_dgrad4.py`.
Each candidate is timed as 20 back-to-back launches replayed from a hipGraph."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

from sbagan import ops  # noqa: E402

B = int(os.environ.get('B', 20))
# (Cin of the forward conv, Cout, input H) of every downBlock / encode_image_by_16times conv with Cin >= 64
# (model.py:550-577, 611-674) at ndf = 64
SHAPES = [(64, 128, 128), (128, 256, 64), (256, 512, 32), (512, 1024, 16), (1024, 2048, 8),
          (64, 128, 64), (128, 256, 32), (256, 512, 16), (512, 1024, 8),
          (64, 128, 32), (128, 256, 16), (256, 512, 8)]


def graph_time(fn, reps=20, iters=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(iters):
            e0.record(s)
            g.replay()
            e1.record(s)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    path = os.path.join(ROOT, 'sba-gan_amd', 'sbagan', 'igemm_table.json')
    with open(path) as f:
        table = json.load(f)
    out = {}
    tot_old = tot_new = 0.0
    print('%-28s %9s | %9s  %-10s' % ('shape (N cin->cout @H)', '4 single', 'grouped', 'plan'))
    for n in (2 * B, B):
        for cin, cout, h in SHAPES:
            oh = h // 2
            dy = torch.randn((n, cout, oh, oh), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            w = torch.nn.Parameter((torch.randn((cout, cin, 4, 4), device=dev) / (cin * 16) ** 0.5)
                                   .contiguous(memory_format=torch.channels_last))
            pw = ops.PackedWeight(w)
            ops.conv_dgrad(dy, pw, '4x4s2', (h, h))
            ops.DGRAD4_GROUP = False
            t_old = graph_time(lambda: ops.conv_dgrad(dy, pw, '4x4s2', (h, h)))
            ops.DGRAD4_GROUP = True
            g0 = ops._geom(('4x4s2_dgrad', n, oh, oh, cout, cin, (0, 0)))
            ns64 = 4 * (cout // 64)
            best = None
            for tile in (1, 3, 5, 7):
                for split in (1, 2, 3, 4, 6, 8):
                    if split > 1 and (ns64 // split < 4 or n * oh * oh > 4096):
                        continue
                    ops._DGRAD4_FORCE = '%d,%d' % (tile, split)
                    t = graph_time(lambda: ops.conv_dgrad(dy, pw, '4x4s2', (h, h)))
                    if best is None or t < best[0]:
                        best = (t, tile, split)
            ops._DGRAD4_FORCE = None
            out[ops.geom_key(g0)] = [best[1], best[2]]
            tot_old += t_old
            tot_new += best[0]
            print('%-28s %9.1f | %9.1f  tile %d split %d' % ('%d %d->%d @%d' % (n, cin, cout, h), t_old, best[0], best[1],
                                                               best[2]), flush=True)
    print('total: four single launches %.0f us, grouped %.0f us' % (tot_old, tot_new))
    table['dgrad4'] = out
    with open(path, 'w') as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print('wrote', path)


if __name__ == '__main__':
    main()
