#!/usr/bin/env python3
"""Which launches of one eager G+D step are PyTorch's own kernels (autograd glue, fills, copies) rather than
this library's, and which line of this package asked for each: one step under torch.profiler with Python stacks,
every device kernel attributed to the innermost frame inside sba-gan_amd/.
python tools/glue_inventory.py [--batch 20] [--image-encoder inception]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    sys.argv += ['--child']
    args = bench.parse()
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    from sbagan.synth import synthetic_batch
    step = bench.build(args, dev)
    b = synthetic_batch(args.batch, branch_num=args.branch, device=dev, seed=100)
    noise = torch.empty((2, args.batch, 100) if args.variant == 'mix' else (args.batch, 100), device=dev)
    a = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    for _ in range(3):
        noise.normal_(0, 1)
        step.step(*a)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step.step(*a)
        torch.cuda.synchronize()
    ours = collections.Counter()
    glue = collections.Counter()
    glue_us = collections.Counter()
    pkg = os.path.join(ROOT, 'sba-gan_amd')
    for ev in prof.events():
        ks = list(getattr(ev, 'kernels', None) or [])
        if not ks or any(getattr(c, 'kernels', None) for c in (ev.cpu_children or [])):
            continue                    # attribute a launch to the innermost operator that owns it
        where = '?'
        for fr in (ev.stack or []):
            if pkg in fr:
                where = fr.replace(pkg + '/', '').strip()
                break
        for k in ks:
            name = k.name
            if 'at::' in name or 'elementwise' in name or 'Cat' in name:
                short = name.split('<')[0].replace('void ', '')
                inner = name[name.find('<') + 1:][:70]
                glue[(where, ev.name, short + '<' + inner)] += 1
                glue_us[(where, ev.name, short + '<' + inner)] += k.duration
            else:
                ours[name.split('(')[0][:60]] += 1
    n_glue = sum(glue.values())
    print('library kernels: %d launches; PyTorch kernels: %d launches, %.1f us' %
          (sum(ours.values()), n_glue, sum(glue_us.values())))
    by_where = collections.Counter()
    for (where, op, kern), n in glue.items():
        by_where[where] += n
    print('\n## PyTorch launches by source line')
    for where, n in by_where.most_common():
        print('%4d  %s' % (n, where))
        for (w, op, kern), m in sorted(glue.items(), key=lambda kv: -kv[1]):
            if w == where:
                print('        %3d  %-28s %s' % (m, op[:28], kern[:90]))


if __name__ == '__main__':
    main()
