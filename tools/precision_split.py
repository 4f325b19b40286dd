#!/usr/bin/env python3
"""Which network owns the bf16 step's distance to the reference?  On the reference's golden fixture (default: model_b20)
in the deterministic mode, the step-0 discriminator losses with
   G bf16 + D bf16  (the benched path)
   G bf16 + D f32   (the generator's roundings only: bf16 fake images judged by exact discriminators)
   G f32  + D bf16  (the discriminators' roundings only)
   G f32  + D f32
each as the relative deviation from the reference's own number.  `python tools/precision_split.py [case]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'model_b20'
    import test_step_gpu as T
    from helpers import FULL, load_golden, make_inputs
    from miscc.config import cfg, reset_cfg
    from miscc.losses import discriminator_loss
    from sbagan import ops
    dev = torch.device('cuda:0')
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    fname, variant, B, branch, slim = T.GOLDEN_STEPS[case]
    Gs = load_golden(os.path.join(ROOT, 'tests', 'golden'), fname)
    x = make_inputs(FULL, B, 18, branch=branch, lmax=18, tag=500)
    imgs = [i.to(dev) for i in x['imgs']]
    sent, words, mask = x['sent'].to(dev), x['words'].to(dev), x['mask'].to(dev)
    from oracle import fill
    nshape = (2, B, 100) if variant == 'mix' else (B, 100)
    noise = fill.unit(nshape, 550).to(dev)
    eps = torch.from_numpy(Gs['step0/eps']).to(dev)
    ops.set_deterministic(True)
    res = {}
    for gdt in (torch.bfloat16, torch.float32):
        ops.set_compute_dtype(gdt)
        st = T._build_step(dev, B, variant, branch)
        st.netG.ca_net.eps = eps
        with torch.no_grad():
            fakes, _, _, _ = st.netG(noise, sent, words, mask)
        fakes = [f.detach().float().clone() for f in fakes]
        for ddt in (torch.bfloat16, torch.float32):
            ops.set_compute_dtype(ddt)
            sd = T._build_step(dev, B, variant, branch)
            for i, d in enumerate(sd.netsD):
                with torch.no_grad():
                    e = discriminator_loss(d, imgs[i], fakes[i], sent, sd.real_labels, sd.fake_labels)
                ref = float(Gs['step0/errD%d' % i])
                res[(str(gdt)[6:], str(ddt)[6:], i)] = (float(e) - ref) / abs(ref)
            del sd
        del st
    ops.set_deterministic(False)
    print('# %s: (errD_i - reference) / reference, step 0, deterministic mode' % case)
    print('%-10s %-10s %12s %12s %12s' % ('G', 'D', 'errD0', 'errD1', 'errD2'))
    for gdt in ('bfloat16', 'float32'):
        for ddt in ('bfloat16', 'float32'):
            print('%-10s %-10s %+12.3e %+12.3e %+12.3e' % ((gdt, ddt) + tuple(res[(gdt, ddt, i)] for i in range(branch))))


if __name__ == '__main__':
    main()
