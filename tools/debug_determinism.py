#!/usr/bin/env python3
"""Run-to-run noise of one discriminator loss + backward on identical inputs (per parameter tensor), f32 and bf16:
legitimate noise is f32 atomic order (~1e-6 relative in f32); anything larger points at a race."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sba-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from helpers import FULL, d_shapes, rel_l2
    from miscc.config import cfg
    from miscc.losses import discriminator_loss
    from oracle import fill
    from sbagan import ops
    import model
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    dev = torch.device('cuda:0')
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for dt in (torch.float32, torch.bfloat16):
        ops.set_compute_dtype(dt)
        for which in (0, 1, 2):
            S = 64 * 2 ** which
            net = [model.D_NET64, model.D_NET128, model.D_NET256][which]()
            net.load_state_dict(fill.fill_state_dict(d_shapes(FULL, which), salt=which))
            net.to(dev).train()
            real, fake = fill.uniform((B, 3, S, S), 950).to(dev), fill.uniform((B, 3, S, S), 951).to(dev)
            sent = fill.unit((B, 256), 952).to(dev)
            runs = []
            for r in range(3):
                for p in net.parameters():
                    p.grad = None
                err = discriminator_loss(net, real, fake, sent, torch.ones(B, device=dev), torch.zeros(B, device=dev))
                err.backward()
                torch.cuda.synchronize()
                runs.append((float(err), {n: p.grad.clone() for n, p in net.named_parameters()}))
            print('== %s D%d B=%d: errD %s' % (dt, which, B, ['%.7f' % r[0] for r in runs]))
            for n in runs[0][1]:
                a = max(rel_l2(runs[1][1][n], runs[0][1][n]), rel_l2(runs[2][1][n], runs[0][1][n]))
                if a > (1e-5 if dt == torch.float32 else 1e-3):
                    print('   %-40s run-to-run rel L2 %.2e' % (n, a))


if __name__ == '__main__':
    main()
