#!/usr/bin/env python3
"""GANStep.real_bwd_early (the real half of the discriminator loss, backward pass included, at the start of the step)
against the two-pass layout it reorders (real_first): one step from the same state, deterministic mode.  The two differ in
the ORDER in which the two halves' gradients are added up (and in the order of the conditional head's BatchNorm
running-statistic updates), not in any term: losses equal to an ulp, everything else to rounding.

    python tools/real_bwd_early_check.py        (one MI355X)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
sys.path.insert(0, ROOT)


def main():
    from helpers import rel_l2
    from test_step_gpu import _build_step
    from test_determinism_gpu import _OrderedStandIn, _state
    from sbagan import ops
    from sbagan.synth import synthetic_batch
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    sm = cfg.TRAIN.SMOOTH
    sm.GAMMA1, sm.GAMMA2, sm.GAMMA3, sm.LAMBDA = 4.0, 5.0, 10.0, 5.0
    dev = torch.device('cuda:0')
    ops.set_deterministic(True)
    worst = 0.0
    for encoder, dt in (('standin', torch.float32), ('inception', torch.bfloat16)):
        ops.set_compute_dtype(dt)
        B = 20
        b = synthetic_batch(B, device=dev, seed=100)
        gen = torch.Generator(device='cpu')
        gen.manual_seed(4321)
        noise = torch.randn((B, 100), generator=gen).to(dev)
        eps = torch.randn((B, 100), generator=gen).to(dev)
        args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
        st = _build_step(dev, B, encoder=encoder)
        if encoder == 'standin':
            st.image_encoder = _OrderedStandIn(256, device=dev)
        orig = st.phase_a
        st.phase_a = lambda se, we, m, nz, e=None: orig(se, we, m, nz, eps)
        for _ in range(2):
            st.step(*args)
        torch.cuda.synchronize()
        snap = st.snapshot()

        def run(early):
            st.restore(snap)
            st.real_first, st.real_bwd_early = True, early
            out = st.step(*args)
            torch.cuda.synchronize()
            return _state(st, out)

        a, c = run(False), run(True)
        again = run(True)
        rep = [k for k in c if not torch.equal(c[k], again[k])]
        rows = []
        for k in a:
            x, y = a[k].float(), c[k].float()
            if not torch.equal(x, y):
                rows.append((float(rel_l2(y, x)), k))
        rows.sort(reverse=True)
        print('%s %s: %d of %d entries differ; not reproducible run to run: %d' % (encoder, dt, len(rows), len(a), len(rep)))
        for r, k in rows[:8]:
            print('   %-60s rel L2 %.3e' % (k, r))
        for k in a:
            if k.startswith('loss/errD'):
                print('   %-30s %r %r' % (k, a[k].flatten()[:1].tolist(), c[k].flatten()[:1].tolist()))
        big = [r for r, k in rows if 'running' not in k and 'num_batches' not in k]
        worst = max([worst] + big)
    print('worst rel L2 outside the BatchNorm running statistics: %.3e' % worst)


if __name__ == '__main__':
    main()
