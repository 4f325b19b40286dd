#!/usr/bin/env python3
"""Where the replayed step spends its time: each captured phase graph (generator forward, the three
discriminator updates, generator loss/backward/Adam) replayed ALONE, back to back, timed with events;
then the whole step.  python tools/phase_times.py [--batch 20] [--image-encoder inception]"""
import os
import sys
import time

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
    from sbagan.trainer import GraphedStep
    step = bench.build(args, dev)
    b = synthetic_batch(args.batch, branch_num=args.branch, device=dev, seed=100)
    noise = torch.empty((2, args.batch, 100) if args.variant == 'mix' else (args.batch, 100), device=dev)
    a = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    for _ in range(4):
        noise.normal_(0, 1)
        step.step(*a)
    g = GraphedStep(step, *a, prologue=lambda: noise.normal_(0, 1))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()

    mark_buf = torch.ones(8, device=dev)

    def timeit(fn, n=10):
        torch.cumsum(mark_buf, 0)           # marker kernel between the groups (tools/prof_phases.py splits on it)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    # replaying a phase alone repeats it on stale inputs of the last full step: same launches, same sizes
    print('phase A (generator forward)      %.3f ms' % timeit(g.gA.replay))
    for i, gd in enumerate(g.gD):
        print('phase D%d (update of D_NET%d)     %.3f ms' % (i, 64 * 2 ** i, timeit(gd.replay)))
    print('phase B (G loss, backward, Adam) %.3f ms' % timeit(g.gB.replay))
    print('whole step (replay)              %.3f ms' % timeit(g.replay))
    t = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print('whole step (wall)                %.3f ms' % ((time.perf_counter() - t) * 100))


if __name__ == '__main__':
    main()
