#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernels on the layer shapes of the B=20 step
(tuning aid; SBA_IGEMM_CFG=A..E forces one tile configuration).  Prints one line per shape:
fwd / dgrad / wgrad microseconds and TFLOP/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

from sbagan import ops  # noqa: E402

B = int(os.environ.get('B', 20))
SHAPES = [
    # kind, Cin, Cout, H(in), tag
    ('3x3up', 512, 512, 4, 'G1.up1'), ('3x3up', 256, 256, 8, 'G1.up2'), ('3x3up', 128, 128, 16, 'G1.up3'),
    ('3x3up', 64, 64, 32, 'G1.up4'), ('3x3', 64, 128, 64, 'G2.res.a'), ('3x3', 64, 64, 64, 'G2.res.b'),
    ('3x3up', 64, 64, 64, 'G2.up'), ('3x3', 64, 128, 128, 'G3.res.a'), ('3x3', 64, 64, 128, 'G3.res.b'),
    ('3x3up', 64, 64, 128, 'G3.up'),
    ('4x4s2', 64, 128, 128, 'D256.c2'), ('4x4s2', 128, 256, 64, 'D256.c3'), ('4x4s2', 256, 512, 32, 'D256.c4'),
    ('4x4s2', 512, 1024, 16, 'D256.s32'), ('4x4s2', 1024, 2048, 8, 'D256.s64'), ('3x3', 2048, 1024, 4, 'D256.s64_1'),
    ('3x3', 1024, 512, 4, 'D256.s64_2'), ('3x3', 768, 512, 4, 'D.joint'),
    ('4x4s2', 64, 128, 32, 'D64.c2'), ('4x4s2', 128, 256, 16, 'D64.c3'), ('4x4s2', 256, 512, 8, 'D64.c4'),
]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    dt = torch.bfloat16 if os.environ.get('DT', 'bf16') == 'bf16' else torch.float32
    print('cfg=%s B=%d dtype=%s' % (os.environ.get('SBA_IGEMM_CFG', 'auto'), B, dt))
    print('%-12s %-7s %5s %5s %4s %8s | %8s %7s | %8s %7s | %8s %7s' % (
        'layer', 'kind', 'Cin', 'Cout', 'H', 'M', 'fwd_us', 'TF/s', 'dgrad_us', 'TF/s', 'wgrad_us', 'TF/s'))
    tot = [0.0, 0.0, 0.0]
    only = os.environ.get('ONLY')
    for kind, cin, cout, h, tag in SHAPES:
        if only and not any(o in tag for o in only.split(',')):
            continue
        k = 4 if kind == '4x4s2' else 3
        x = torch.randn((B, cin, h, h), device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter((torch.randn((cout, cin, k, k), device=dev) / (cin * k * k) ** 0.5)
                               .contiguous(memory_format=torch.channels_last))
        pw = ops.PackedWeight(w)
        y, _ = ops.conv_forward(x, pw, kind)
        dy = torch.randn_like(y)
        oh = y.shape[2]
        flops = 2.0 * B * oh * oh * cout * cin * k * k
        t_f = timeit(lambda: ops.conv_forward(x, pw, kind))
        t_d = timeit(lambda: ops.conv_dgrad(dy, pw, kind, (h, h)))
        t_w = timeit(lambda: ops.conv_wgrad(x, dy, w, kind))
        tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
        print('%-12s %-7s %5d %5d %4d %8d | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f' % (
            tag, kind, cin, cout, h, B * oh * oh, t_f, flops / t_f / 1e6, t_d, flops / t_d / 1e6, t_w,
            flops / t_w / 1e6))
    print('total us: fwd %.0f dgrad %.0f wgrad %.0f' % tuple(tot))


if __name__ == '__main__':
    main()
