"""Where does the bf16 step's deviation of errD_i from the reference come from?  CPU emulation on the golden fixture
(step_full_model_b20: B = 20, bird_style dims): the oracle's discriminator with bf16 ROUNDING inserted at one storage
point at a time -- packed weights, stem output, raw conv outputs, activations, the heads' tensors, the fake image --
and the resulting relative deviation of the step-0 discriminator loss.  Rounding points mirror what the HIP path
stores in bf16.

    python tools/bf16_bisect.py [which=0]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from helpers import FULL, d_shapes, g_shapes, make_inputs  # noqa: E402
from oracle import fill  # noqa: E402
from oracle import sbagan_oracle as O  # noqa: E402


def r16(x):
    return x.to(torch.bfloat16).float()


class Emu(object):
    def __init__(self, P, on):
        self.P, self.on = P, set(on)

    def q(self, name, x):
        return r16(x) if (name in self.on or 'all' in self.on) else x

    def w(self, k):
        return self.q('W', self.P[k])

    def bn(self, y, prefix, tag):
        """statistics from the unrounded accumulators (conv epilogue), normalisation applied to the stored tensor"""
        w, b = self.P[prefix + '.weight'], self.P[prefix + '.bias']
        stat_src = y if 'stats_rounded' not in self.on else self.q('Y' + tag, y)
        mean = stat_src.mean((0, 2, 3))
        var = (stat_src ** 2).mean((0, 2, 3)) - mean ** 2
        ys = self.q('Y' + tag, y)
        sc = w / torch.sqrt(var + 1e-5)
        sh = b - mean * sc
        return ys * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)

    def trunk(self, x):
        P = self.P
        x = F.leaky_relu(F.conv2d(x, P['img_code_s16.0.weight'], None, 2, 1), 0.2)
        x = self.q('A0', x)
        for n, (iw, ib) in enumerate(((2, 3), (5, 6), (8, 9))):
            y = F.conv2d(x, self.w('img_code_s16.%d.weight' % iw), None, 2, 1)
            x = self.q('A%d' % (n + 1), F.leaky_relu(self.bn(y, 'img_code_s16.%d' % ib, str(n + 1)), 0.2))
        k = 4
        for pre, kind in (('img_code_s32', 'down'), ('img_code_s64', 'down'), ('img_code_s64_1', '3x3'),
                          ('img_code_s64_2', '3x3'), ('img_code_s32_1', '3x3')):
            if pre + '.0.weight' not in P:
                continue
            y = F.conv2d(x, self.w(pre + '.0.weight'), None, 2 if kind == 'down' else 1, 1)
            x = self.q('A%d' % k, F.leaky_relu(self.bn(y, pre + '.1', str(k)), 0.2))
            k += 1
        return x

    def logits(self, prefix, h, c=None):
        P = self.P
        if c is not None:
            cc = self.q('C', c).view(c.size(0), -1, 1, 1).repeat(1, 1, 4, 4)
            y = F.conv2d(torch.cat((h, cc), 1), self.w(prefix + '.jointConv.0.weight'), None, 1, 1)
            h = self.q('AJ', F.leaky_relu(self.bn(y, prefix + '.jointConv.1', 'J'), 0.2))
        o = F.conv2d(h, P[prefix + '.outlogits.0.weight'], P[prefix + '.outlogits.0.bias'], 4)
        return torch.sigmoid(o).view(-1)

    def loss(self, real, fake, cond):
        B = real.size(0)
        ones, zeros = torch.ones(B), torch.zeros(B)
        rf, ff = self.trunk(real), self.trunk(fake)
        cr = O.bce(self.logits('COND_DNET', rf, cond), ones)
        cf = O.bce(self.logits('COND_DNET', ff, cond), zeros)
        cw = O.bce(self.logits('COND_DNET', rf[:B - 1], cond[1:B]), zeros[1:B])
        r = O.bce(self.logits('UNCOND_DNET', rf), ones)
        f = O.bce(self.logits('UNCOND_DNET', ff), zeros)
        return float((r + cr) / 2. + (f + cf + cw) / 3.)


def main():
    which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    torch.set_num_threads(8)
    B = 20
    G = np.load(os.path.join(ROOT, 'tests', 'golden', 'step_full_model_b20.npz'), allow_pickle=True)
    x = make_inputs(FULL, B, 18, branch=3, lmax=18, tag=500)
    PG = fill.fill_state_dict(g_shapes(FULL, which + 1, 'model'))
    PD = fill.fill_state_dict(d_shapes(FULL, which), salt=which)
    noise = fill.unit((B, 100), 550)
    eps = torch.from_numpy(G['step0/eps'])
    with torch.no_grad():
        fakes = O.g_net(PG, noise, x['sent'], x['words'], x['mask'], eps, which + 1, 'model')[0]
        fake, real, cond = fakes[which], x['imgs'][which], x['sent']
        ref = Emu(PD, ()).loss(real, fake, cond)
        gold = float(G['step0/errD%d' % which])
        print('oracle errD%d %.8f   golden %.8f   rel %.2e' % (which, ref, gold, abs(ref - gold) / gold))
        names = ['W', 'A0'] + ['Y%d' % k for k in range(1, 8)] + ['A%d' % k for k in range(1, 8)] + ['C', 'YJ', 'AJ']
        for n in names + ['all']:
            v = Emu(PD, (n,)).loss(real, fake, cond)
            if v != ref or n == 'all':
                print('  round %-4s  errD %.8f  rel dev %+.3e' % (n, v, (v - ref) / ref))
        v = Emu(PD, ('all', 'stats_rounded')).loss(real, fake, cond)
        print('  all + statistics of the ROUNDED conv output: rel dev %+.3e' % ((v - ref) / ref))
        allbut = lambda drop: Emu(PD, [n for n in names if n not in drop]).loss(real, fake, cond)      # noqa: E731
        for drop in (('W',), ('A0',), ('Y1', 'Y2', 'Y3', 'Y4', 'Y5', 'Y6', 'Y7', 'YJ'), ('A3', 'AJ', 'C'),
                     ('A1', 'A2', 'A3', 'A4', 'A5', 'A6', 'A7', 'AJ')):
            v = allbut(drop)
            print('  all but %-40s rel dev %+.3e' % (','.join(drop), (v - ref) / ref))


if __name__ == '__main__':
    main()
