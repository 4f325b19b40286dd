#!/usr/bin/env python3
"""Where a stage of the LDS-DMA implicit GEMM spends its cycles: wave 0 of the first 32 workgroups stamps
s_memtime at five points of every stage (trace build of csrc/igemm.hip: tools/_ab/libtrace.so, -DSBA_DMA_TRACE).
usage: trace_dma.py TILE N H W Cin Cout KH KW"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sbagan import ops  # noqa: E402
from sbagan._lib import ConvGeom  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_ab', 'libtrace.so'))
lib.sba_conv_igemm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.POINTER(ConvGeom), ctypes.c_void_p,
                                                                          ctypes.c_int64, ctypes.c_void_p]
lib.sba_set_dma_trace.argtypes = [ctypes.c_void_p]


def main():
    tile, N, H, W, Cin, Cout, KH, KW = [int(v) for v in sys.argv[1:9]]
    dev = torch.device('cuda:0')
    g = ConvGeom()
    g.N, g.IH, g.IW, g.Cin, g.Cout = N, H, W, Cin, Cout
    g.OH = g.OHs = H
    g.OW = g.OWs = W
    g.sy = g.sx = g.osy = g.osx = 1
    g.ntaps = KH * KW
    g.tile, g.ksplit = tile, 1
    for t in range(KH * KW):
        g.ty[t], g.tx[t] = t // KW - KH // 2, t % KW - KW // 2
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, KH * KW, Cin, device=dev) / (Cin * KH * KW) ** 0.5).bfloat16()
    y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
    trace = torch.zeros(32 * 32 * 8, dtype=torch.int64, device=dev)
    ws = ops.workspace(dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.sba_conv_igemm(1, x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, ctypes.byref(g),
                                ws.data_ptr(), ops.WORKSPACE_BYTES, st)
        assert rc == 0, rc
    for _ in range(3):
        run()
    lib.sba_set_dma_trace(trace.data_ptr())
    run()
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(32, 32, 8)
    nst = int((t[0, :, 0] != 0).sum())
    print('tile %d  M=%d N=%d K=%d: %d stages traced' % (tile, N * H * W, Cout, Cin * KH * KW, nst))
    names = ['wait vmcnt', 'barrier', 'issue DMA | (gen2) read frags', 'ds_read + MFMA | (gen2) MFMA + DMA']
    for wg in (0, 1, 8, 17):
        seg = np.diff(t[wg, :nst, :5].astype(np.int64), axis=1)          # [stage][4]
        tot = t[wg, nst - 1, 4] - t[wg, 0, 0]
        print(' workgroup slot %2d: %6d cycles over %d stages (%.0f per stage); mean per stage: %s' % (
            wg, tot, nst, tot / nst, ', '.join('%s %.0f' % (n, v) for n, v in zip(names, seg[2:].mean(0)))))
        if wg == 0:
            for s in range(min(nst, 8)):
                print('    stage %d: %s' % (s, ' '.join('%5d' % v for v in seg[s])))


if __name__ == '__main__':
    main()
