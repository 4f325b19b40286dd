#!/usr/bin/env python3
"""Micro-benchmark of sba_conv_igemm on arbitrary (kh x kw, stride 1, same-padded) shapes, e.g. the
Inception trunk's 17x17 factorised convs (tuning aid; SBA_IGEMM_CFG=A..E forces one configuration).
usage: bench_shape.py N H W Cin Cout KH KW [N H W Cin Cout KH KW ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

from sbagan import ops  # noqa: E402
from sbagan._lib import ConvGeom, call  # noqa: E402

if os.environ.get('BENCH_LIB'):           # experimental build of the library (tuning aid)
    _alt = ctypes.CDLL(os.path.join(ROOT, os.environ['BENCH_LIB']))
    _alt.sba_conv_igemm_bias.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7 + [
        ctypes.POINTER(ConvGeom), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]

    def call(name, *a):    # noqa: F811
        rc = getattr(_alt, name)(*a)
        assert rc == 0, rc

DEFAULT = [20, 17, 17, 192, 192, 1, 7, 20, 17, 17, 768, 192, 1, 1, 20, 17, 17, 128, 128, 7, 1,
           20, 35, 35, 288, 64, 1, 1, 20, 35, 35, 64, 96, 3, 3, 20, 35, 35, 64, 64, 5, 5,
           20, 8, 8, 1280, 320, 1, 1, 20, 8, 8, 384, 384, 1, 3, 20, 8, 8, 2048, 448, 1, 1, 1, 8, 8, 32, 64, 1, 1]


def main():
    a = [int(v) for v in sys.argv[1:]] or DEFAULT
    dev = torch.device('cuda:0')
    print('cfg=%s d2f=%s' % (os.environ.get('SBA_IGEMM_CFG', 'auto'), os.environ.get('SBA_IGEMM_D2F', '0')))
    for i in range(0, len(a), 7):
        N, H, W, Cin, Cout, KH, KW = a[i:i + 7]
        g = ConvGeom()
        g.N, g.IH, g.IW, g.Cin, g.Cout = N, H, W, Cin, Cout
        g.OH = g.OHs = H
        g.OW = g.OWs = W
        g.sy = g.sx = g.osy = g.osx = 1
        g.ntaps = KH * KW
        for t in range(KH * KW):
            g.ty[t], g.tx[t] = t // KW - KH // 2, t % KW - KW // 2
        x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
        w = (torch.randn(Cout, KH * KW, Cin, device=dev) / (Cin * KH * KW) ** 0.5).bfloat16()
        y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        bias = torch.zeros(Cout, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(side)
        ws = ops.workspace(dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            call('sba_conv_igemm_bias', 1, x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None,
                 bias.data_ptr() if os.environ.get('BIAS', '1') == '1' else None, None,
                 ctypes.byref(g), ws.data_ptr(), ops.WORKSPACE_BYTES, st)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us_eager = e0.elapsed_time(e1) / 20 * 1e3
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(20):
                run()
        gr.replay()
        torch.cuda.synchronize()
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * N * H * W * Cout * Cin * KH * KW
        print('N%d %dx%d Cin%-5d Cout%-5d %dx%d  M=%-6d K=%-6d graph %8.1f us %7.1f TF/s | eager %8.1f us' % (
            N, H, W, Cin, Cout, KH, KW, N * H * W, Cin * KH * KW, us, fl / us / 1e6, us_eager))


if __name__ == '__main__':
    main()
