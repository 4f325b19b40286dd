#!/usr/bin/env python3
"""Per-workgroup timeline of one igemm launch (needs the traced build tools/_trace/libsbagan_trace.so,
which stamps wall_clock64() at entry / loop start / loop end / exit of every workgroup into the
`addend` pointer).  Tuning aid only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402
import numpy as np  # noqa: E402

from sbagan._lib import ConvGeom  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_trace', 'libsbagan_trace.so'))
lib.sba_conv_igemm_bias.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.POINTER(ConvGeom), ctypes.c_void_p,
                                                                            ctypes.c_int64, ctypes.c_void_p]


def main():
    mode = sys.argv[1]            # bias | none | stats
    a = [int(v) for v in sys.argv[2:]]
    dev = torch.device('cuda:0')
    for i in range(0, len(a), 7):
        N, H, W, Cin, Cout, KH, KW = a[i:i + 7]
        g = ConvGeom()
        g.N, g.IH, g.IW, g.Cin, g.Cout = N, H, W, Cin, Cout
        g.OH = g.OHs = H
        g.OW = g.OWs = W
        g.sy = g.sx = g.osy = g.osx = 1
        g.ntaps = KH * KW
        if os.environ.get('UPS') == '1':
            g.ups = 1
            g.IH, g.IW = H // 2, W // 2
        for t in range(KH * KW):
            g.ty[t], g.tx[t] = t // KW - KH // 2, t % KW - KW // 2
        x = torch.randn(N, g.IH, g.IW, Cin, device=dev).bfloat16()
        w = (torch.randn(Cout, KH * KW, Cin, device=dev) / (Cin * KH * KW) ** 0.5).bfloat16()
        y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        bias = torch.zeros(Cout, device=dev)
        stats = torch.zeros(2 * Cout, device=dev)
        tr = torch.zeros(8 * 65536, device=dev, dtype=torch.int64)
        ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            rc = lib.sba_conv_igemm_bias(1, x.data_ptr(), w.data_ptr(), y.data_ptr(), tr.data_ptr(),
                                         stats.data_ptr() if mode == 'stats' else None,
                                         bias.data_ptr() if mode == 'bias' else None, None, ctypes.byref(g), ws.data_ptr(), ws.numel(), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        t = tr.cpu().numpy().reshape(-1, 8)[:, :7]
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        t = (t - t0) * 0.01          # us (100 MHz constant clock)
        order = np.argsort(t[:, 0])
        print(mode, 'shape N%d %dx%d Cin%d Cout%d %dx%d: %d workgroups, span %.2f us' % (N, H, W, Cin, Cout, KH, KW, len(t),
                                                                                 t[:, 6].max()))
        print('  start: p10 %.2f p50 %.2f p90 %.2f max %.2f' % tuple(np.percentile(t[:, 0], [10, 50, 90, 100])))
        d = t[:, 1:] - t[:, :-1]
        for k, nm in enumerate(('phase 0->1', 'phase 1->2', 'phase 2->3', 'phase 3->4', 'phase 4->5', 'phase 5->6')):
            print('  %-9s: p10 %.2f p50 %.2f p90 %.2f max %.2f' % ((nm,) + tuple(np.percentile(d[:, k], [10, 50, 90, 100]))))
        print('  end  : p10 %.2f p50 %.2f p90 %.2f max %.2f' % tuple(np.percentile(t[:, 6], [10, 50, 90, 100])))
        print('  first 8 starts:', np.round(t[order[:8], 0], 2), ' last 4 starts:', np.round(t[order[-4:], 0], 2))


if __name__ == '__main__':
    main()
