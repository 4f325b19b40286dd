#!/usr/bin/env python3
"""Where do the 8-17 % per-tensor bf16 errors of the discriminators' weight gradients come from?
CPU emulation on a D_NET64-shaped trunk (conv4x4/s2 -> BatchNorm(train) -> LeakyReLU, B = 3 and B = 20, float64 reference):
the gradient entering each BatchNorm backward is rounded to bf16 as the HIP path stores it, and the BatchNorm backward

    dy = gamma rstd (dz - mean(dz) - xhat mean(dz xhat))

is evaluated three ways:
    A  sums AND the per-element dz from the bf16 tensor            (the HIP path)
    B  sums from the UNROUNDED dz (what "statistics from the f32 dgrad accumulators" would give), per-element dz bf16
    C  everything unrounded                                          (only the activations / weights are bf16)
    D  as A, but the FORWARD pass unrounded (f64 activations and weights): the gradient roundings alone
Prints the relative L2 error of every conv weight gradient against the float64 reference.   python tools/bn_bwd_rounding.py"""
import torch
import torch.nn.functional as F


def r16(x):
    return x.to(torch.bfloat16).to(x.dtype)


class BN(torch.autograd.Function):
    mode = 'C'

    @staticmethod
    def forward(ctx, y, gamma, beta):
        mean = y.mean((0, 2, 3), keepdim=True)
        var = y.var((0, 2, 3), unbiased=False, keepdim=True)
        rstd = (var + 1e-5).rsqrt()
        xhat = (y - mean) * rstd
        ctx.save_for_backward(xhat, rstd, gamma)
        return xhat * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dz):
        xhat, rstd, gamma = ctx.saved_tensors
        dzq = r16(dz) if BN.mode in ('A', 'B') else dz          # the stored tensor
        src = dzq if BN.mode == 'A' else dz                     # what the sums are taken from
        m1 = src.mean((0, 2, 3), keepdim=True)
        m2 = (src * xhat).mean((0, 2, 3), keepdim=True)
        dy = gamma.view(1, -1, 1, 1) * rstd * (dzq - m1 - xhat * m2)
        return dy, (src * xhat).sum((0, 2, 3)), src.sum((0, 2, 3))


def run(B, mode, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    chans = [3, 64, 128, 256, 512]
    x = torch.rand((B, 3, 64, 64), generator=g, dtype=torch.float64) * 2 - 1
    ws, gs, bs = [], [], []
    for i in range(4):
        w = torch.randn((chans[i + 1], chans[i], 4, 4), generator=g, dtype=torch.float64) / (chans[i] * 16) ** 0.5
        ws.append(w.to(dtype).requires_grad_(True))
        gs.append((1 + 0.1 * torch.randn(chans[i + 1], generator=g, dtype=torch.float64)).to(dtype).requires_grad_(True))
        bs.append((0.1 * torch.randn(chans[i + 1], generator=g, dtype=torch.float64)).to(dtype).requires_grad_(True))
    wl = (torch.randn((1, 512, 4, 4), generator=g, dtype=torch.float64) / 90).to(dtype)
    BN.mode = mode
    low = mode not in ('ref', 'D')
    q = r16 if low else (lambda t: t)
    if mode == 'D':
        BN.mode = 'A'
    h = x.to(dtype)
    for i in range(4):
        y = F.conv2d(q(h), q(ws[i]), None, 2, 1)
        if i == 0:
            h = q(F.leaky_relu(y, 0.2))
        else:
            h = q(F.leaky_relu(BN.apply(q(y), gs[i], bs[i]), 0.2))
    p = torch.sigmoid(F.conv2d(h, wl, None, 4).view(-1))
    loss = F.binary_cross_entropy(p, torch.ones_like(p))
    loss.backward()
    return [w.grad.double() for w in ws]


def main():
    for B in (3, 20):
        BN.mode = 'C'
        ref = run(B, 'ref', torch.float64)
        print('B = %d: relative L2 error of d(conv weight) per layer' % B)
        for mode, what in (('A', 'A  bf16 dz everywhere (HIP path)      '), ('B', 'B  f32 sums, bf16 per-element dz      '),
                           ('C', 'C  unrounded dz (bf16 activations only)'), ('D', 'D  bf16 dz, UNROUNDED forward pass    ')):
            got = run(B, mode, torch.float64)
            errs = [float((a - b).norm() / b.norm()) for a, b in zip(got, ref)]
            print('   %s  %s' % (what, '  '.join('%.3f' % e for e in errs)))


if __name__ == '__main__':
    main()
