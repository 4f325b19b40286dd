#!/usr/bin/env python3
"""Weight-gradient launches of the benched step, one shape at a time (B = 20): time per launch under the library's
current dispatch.  Run once per setting of SBA_WGRAD_DMA (0 = register-staged, 3 / 4 = LDS-DMA ring depth); the
library reads it once.  python tools/bench_wgrad.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

# kind, N, Cin, Cout, H, W  (input map)
SHAPES = [
    ('3x3', 20, 768, 512, 4, 4, 'D jointConv'),
    ('3x3', 19, 768, 512, 4, 4, 'D jointConv (wrong pairs)'),
    ('4x4s2', 40, 256, 512, 8, 8, 'D s16 last down block (real|fake)'),
    ('4x4s2', 40, 512, 1024, 8, 8, 'D128 s32'),
    ('4x4s2', 40, 1024, 2048, 8, 8, 'D256 s64'),
    ('3x3', 40, 1024, 512, 4, 4, 'D128 s32_1'),
    ('3x3', 40, 2048, 1024, 4, 4, 'D256 s64_1'),
    ('3x3', 40, 1024, 512, 4, 4, 'D256 s64_2'),
    ('3x3up', 20, 1024, 1024, 4, 4, 'G upsample1'),
    ('3x3up', 20, 512, 512, 8, 8, 'G upsample2'),
    ('3x3up', 20, 256, 256, 16, 16, 'G upsample3'),
    ('3x3up', 20, 128, 128, 32, 32, 'G upsample4'),
    ('4x4s2', 40, 128, 256, 16, 16, 'D s16 third down block'),
    ('4x4s2', 40, 64, 128, 128, 128, 'D256 down 64->128 @128'),
    ('4x4s2', 40, 128, 256, 64, 64, 'D256 down 128->256 @64'),
    ('4x4s2', 40, 256, 512, 32, 32, 'D256 down 256->512 @32'),
    ('4x4s2', 40, 64, 128, 64, 64, 'D128 down 64->128 @64'),
    ('4x4s2', 40, 64, 128, 32, 32, 'D64 down 64->128 @32'),
    ('3x3', 20, 64, 64, 64, 64, 'G ResBlock 64x64'),
    ('3x3', 20, 64, 128, 128, 128, 'G ResBlock 128x128 conv1 (64->128)'),
    ('3x3', 20, 64, 64, 128, 128, 'G ResBlock 128x128 conv2 (64->64)'),
    ('3x3up', 20, 64, 64, 128, 128, 'G upBlock -> 256x256 (64->64)'),
    ('3x3up', 20, 64, 128, 64, 64, 'G upBlock -> 128x128 (64->128)'),
    ('3x3', 20, 64, 128, 64, 64, 'G ResBlock 64x64 conv1 (64->128)'),
]


def main():
    from sbagan import ops
    dev = torch.device('cuda:0')
    ops.set_compute_dtype(torch.bfloat16)
    print('SBA_WGRAD_DMA=%s SBA_WGRAD_GEN_DMA=%s' % (os.environ.get('SBA_WGRAD_DMA', '(default)'), os.environ.get('SBA_WGRAD_GEN_DMA', '(default)')))
    only = os.environ.get('BENCH_WGRAD_ONLY')           # substring of the shape's name: that shape alone (PMC runs)
    for kind, N, Cin, Cout, H, W, name in SHAPES:
        if only and only not in name:
            continue
        k = 4 if kind == '4x4s2' else 3
        w = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device=dev).contiguous(memory_format=torch.channels_last))
        x = torch.randn(N, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        OH, OW = ops._conv_out_hw(kind, H, W)
        dy = torch.randn(N, Cout, OH, OW, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if os.environ.get('BENCH_FIRST_WRITE', '0') == '1':     # every launch as the first write of a cleared gradient
            w._sba_gepoch = [0]
            _orig = ops.conv_wgrad

            def _fw(x_, dy_, w_, kind_):
                w_._sba_gepoch[0] += 1
                return _orig(x_, dy_, w_, kind_)
            call_wgrad = _fw
        else:
            call_wgrad = ops.conv_wgrad
        for _ in range(3):
            call_wgrad(x, dy, w, kind)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            call_wgrad(x, dy, w, kind)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        M = N * OH * OW
        fl = 2.0 * M * Cout * Cin * k * k
        print('%-36s %-6s M=%-6d %5dx%-5d  %8.1f us  %6.1f TFLOP/s  dW %6.1f MB' %
              (name, kind, M, Cin, Cout, us, fl / us * 1e-6, Cout * Cin * k * k * 4 / 1e6))


if __name__ == '__main__':
    main()
