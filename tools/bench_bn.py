#!/usr/bin/env python3
"""BatchNorm + activation passes of the benched step at their large shapes (B = 20), one at a time: forward
(normalise + activation), backward reduce + apply.  Algorithmic bytes / time against the achievable HBM rate.
python tools/bench_bn.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

# name, act, N, C (BatchNorm channels), H, W
SHAPES = [
    ('G upBlock -> 256x256 (GLU)', 'glu', 20, 64, 256, 256),
    ('G upBlock -> 128x128 (GLU)', 'glu', 20, 128, 128, 128),
    ('G upBlock -> 64x64 (GLU)', 'glu', 20, 64, 64, 64),
    ('G ResBlock 128x128 conv1 (GLU)', 'glu', 20, 128, 128, 128),
    ('G ResBlock 128x128 conv2 (none)', 'none', 20, 64, 128, 128),
    ('D256 down 64x64 (LeakyReLU), real|fake', 'lrelu', 40, 128, 64, 64),
    ('D256 down 32x32 (LeakyReLU), real|fake', 'lrelu', 40, 256, 32, 32),
]


def main():
    from sbagan import ops
    from sbagan._lib import ACT_GLU, ACT_LRELU, ACT_NONE
    dev = torch.device('cuda:0')
    ops.set_compute_dtype(torch.bfloat16)
    acts = {'glu': ACT_GLU, 'lrelu': ACT_LRELU, 'none': ACT_NONE}
    print('SBA_BN_RED_BLOCKS=%s' % os.environ.get('SBA_BN_RED_BLOCKS', '(default)'))

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    for name, act, N, C, H, W in SHAPES:
        a = acts[act]
        Co = C // 2 if act == 'glu' else C
        groups = 2 if 'real|fake' in name else 1
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        y = torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dout = torch.randn(N, Co, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        stats = ops.bn_stats(y, groups)
        out, st = ops.bn_act_forward(y, stats, bn, a, None, groups)
        ey, eo = y.numel() * 2, out.numel() * 2
        t_f = timeit(lambda: ops.bn_act_forward(y, stats, bn, a, None, groups))
        red = torch.zeros((groups, ops.BN_STAT_SLOTS, 2 * C), dtype=torch.float32, device=dev)
        dy = torch.empty_like(y)
        rows = (N // groups) * H * W

        def reduce_():
            ops.call('sba_bn_act_bwd_reduce', ops._dt(y), ops._p(y), ops._p(dout), ops._p(st.aux), ops._p(red), rows,
                     groups, C, a, Co, 0, ops._stream())

        def apply_():
            ops.call('sba_bn_act_bwd_apply', ops._dt(y), ops._p(y), ops._p(dout), ops._p(st.aux), ops._p(red),
                     ops._p(dy), None, None, rows, groups, C, a, Co, 0, ops._stream())
        t_r, t_a = timeit(reduce_), timeit(apply_)
        print('%-42s fwd %6.1f us %5.2f TB/s | reduce %6.1f us %5.2f TB/s | apply %6.1f us %5.2f TB/s' %
              (name, t_f, (ey + eo) / t_f * 1e-6, t_r, (ey + eo) / t_r * 1e-6, t_a, (2 * ey + eo) / t_a * 1e-6))


if __name__ == '__main__':
    main()
