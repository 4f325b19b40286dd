#!/usr/bin/env python3
"""The HBM-bound kernels of the step at their B = 20 shapes: microseconds, algorithmic bytes, GB/s and
the fraction of the achievable HBM rate (6.3 TB/s measured streaming rate of MI355X, MI355X_MICROARCH.md).
    python tools/bench_hbm_kernels.py            # events on the launch stream
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/bench_hbm_kernels.py   # traffic"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

from sbagan import ops  # noqa: E402
from sbagan._lib import call  # noqa: E402

ACHIEVABLE = 6.3e12


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
    ops.set_compute_dtype(torch.bfloat16)
    ops.reduce_scratch(dev)          # the ring of the two-stage weight-gradient reductions, as the operators set it
    B, C = 20, 32
    st = torch.cuda.current_stream().cuda_stream
    rows = []

    def rec(name, us, nbytes):
        rows.append((name, us, nbytes))
        print('%-44s %8.1f us  %8.1f MB  %7.0f GB/s  %5.1f %% of achievable HBM' % (
            name, us, nbytes / 1e6, nbytes / us / 1e3, 100 * nbytes / (us * 1e-6) / ACHIEVABLE), flush=True)
    for S in (128, 256):
        h = torch.randn((B, C, S, S), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn((3, C, 3, 3), device=dev) / 17).contiguous(memory_format=torch.channels_last)
        img = torch.empty((B, 3, S, S), device=dev)
        dimg = torch.randn((B, 3, S, S), device=dev)
        dh = torch.empty_like(h)
        dw = torch.zeros_like(w)
        px = B * S * S
        rec('img_head_fwd @%d' % S, timeit(lambda: call('sba_img_head_fwd', 1, h.data_ptr(), w.data_ptr(), img.data_ptr(),
                                                        B, S, S, C, st)), px * (C * 2 + 12))
        rec('img_head_bwd @%d' % S, timeit(lambda: call('sba_img_head_bwd', 1, h.data_ptr(), w.data_ptr(), img.data_ptr(),
                                                        dimg.data_ptr(), dh.data_ptr(), dw.data_ptr(), B, S, S, C, 0, st)),
            px * (C * 2 + 12 + 12 + C * 2))
        # discriminator stem: conv4x4 s2 3 -> 64 + LeakyReLU on [real | fake] (2B images)
        x = torch.randn((2 * B, 3, S, S), device=dev)
        ws = (torch.randn((64, 3, 4, 4), device=dev) / 7).contiguous(memory_format=torch.channels_last)
        out = torch.empty((2 * B, 64, S // 2, S // 2), device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dout = torch.randn_like(out)
        dx = torch.empty_like(x)
        dws = torch.zeros_like(ws)
        opx = 2 * B * (S // 2) ** 2
        rec('d_stem_fwd @%d (2B images)' % S, timeit(lambda: call('sba_d_stem_fwd', 1, x.data_ptr(), ws.data_ptr(), out.data_ptr(),
                                                                  2 * B, S, 64, st)), 2 * B * 3 * S * S * 4 + opx * 128)
        rec('d_stem_wgrad @%d' % S, timeit(lambda: call('sba_d_stem_bwd', 1, x.data_ptr(), ws.data_ptr(), out.data_ptr(),
                                                        dout.data_ptr(), None, dws.data_ptr(), 2 * B, S, 64, st)),
            2 * B * 3 * S * S * 4 + 2 * opx * 128)
        xb = x[:B].contiguous()
        ob, dob = out[:B].contiguous(memory_format=torch.channels_last), dout[:B].contiguous(memory_format=torch.channels_last)
        dxb = torch.empty_like(xb)
        rec('d_stem_dgrad @%d (B images)' % S, timeit(lambda: call('sba_d_stem_bwd', 1, xb.data_ptr(), ws.data_ptr(), ob.data_ptr(),
                                                                   dob.data_ptr(), dxb.data_ptr(), None, B, S, 64, st)),
            B * 3 * S * S * 4 + 2 * (opx // 2) * 128)
    # word attention at stage 3 (128 x 128 queries), through the fused entry of NEXT_STAGE_G
    for S in (64, 128):
        L = 18
        h = torch.randn((B, C, S, S), device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        style = torch.randn((B, 2 * C), device=dev)
        words = torch.randn((B, 256, L), device=dev)
        wctx = torch.nn.Parameter(torch.randn((C, 256, 1, 1), device=dev) / 16)
        mask = torch.zeros((B, L), dtype=torch.bool, device=dev)
        out, _ = ops.AttnAdainCatFn.apply(h, style, words, wctx, mask, False, 0)
        dout = torch.randn_like(out)
        px = B * S * S
        src = torch.randn((B, C, L), device=dev)
        m8 = mask.to(torch.uint8)
        o2 = torch.empty_like(out)
        rec('word_attn_fwd %dx%d' % (S, S), timeit(lambda: call('sba_word_attn_fwd', 1, h.data_ptr(), src.data_ptr(), m8.data_ptr(),
                                                               o2.data_ptr(), None, B, S * S, C, L, 0, 2 * C, C, st)),
            px * C * 2 * 2)
        dh, dsrc = torch.empty_like(h), torch.zeros((B, C, L), device=dev)
        rec('word_attn_bwd %dx%d' % (S, S), timeit(lambda: call('sba_word_attn_bwd', 1, h.data_ptr(), src.data_ptr(), m8.data_ptr(),
                                                               dout.data_ptr(), dh.data_ptr(), dsrc.data_ptr(), B, S * S, C, L, 0,
                                                               2 * C, C, 0, st)), px * C * 2 * 3)
    # fused Adam + EMA on a D_NET256-sized buffer
    n = 71_860_000 // 4 * 4
    p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
    v.abs_()
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    call('sba_adam_prepare', state.data_ptr(), 2e-4, 0.5, 0.999, st)
    rec('adam_step 71.9 M parameters', timeit(lambda: call('sba_adam_step', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                           None, None, state.data_ptr(), n, 0.5, 0.999, 1e-8, 1.0, st)), n * 28)
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'hbm_kernels.txt'), 'w') as f:
            for name, us, nb in rows:
                f.write('%-44s %8.1f us  %8.1f MB  %7.0f GB/s  %5.1f %% of achievable HBM (6.3 TB/s)\n' % (
                    name, us, nb / 1e6, nb / us / 1e3, 100 * nb / (us * 1e-6) / ACHIEVABLE))


if __name__ == '__main__':
    main()
