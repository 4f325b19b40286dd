#!/usr/bin/env python3
"""Micro-benchmark of the DAMSM words-loss kernels at the training shapes (B=20, nef=256, 17x17
regions, 18 words).  BENCH_LIB=<path> times an experimental build of the library (tuning aid)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

from sbagan import _lib  # noqa: E402

lib = _lib.LIB if hasattr(_lib, 'LIB') else None
if os.environ.get('BENCH_LIB'):
    lib = ctypes.CDLL(os.path.join(ROOT, os.environ['BENCH_LIB']))
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def main():
    dev = torch.device('cuda:0')
    B, nef, R, L = 20, 256, 289, 18
    torch.manual_seed(0)
    feat = torch.randn(B, nef, R, device=dev)
    words = torch.randn(B, nef, L, device=dev)
    lens = torch.randint(5, L + 1, (B,), device=dev, dtype=torch.int64)
    sim = torch.zeros(B * B, device=dev)
    attn = torch.zeros(B * B * L * R, device=dev)
    attn1 = torch.zeros(B * B * L * R, device=dev)
    wctx = torch.zeros(B * B * L * nef, device=dev)
    dsim = torch.randn(B * B, device=dev)
    dfeat = torch.zeros(B, nef, R, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    if lib is None:
        fwd = lambda: _lib.call('sba_damsm_words_fwd', feat.data_ptr(), words.data_ptr(), lens.data_ptr(), sim.data_ptr(),
                                attn.data_ptr(), attn1.data_ptr(), wctx.data_ptr(), B, nef, R, L, 5.0, 5.0, st)
        bwd = lambda: _lib.call('sba_damsm_words_bwd', feat.data_ptr(), words.data_ptr(), lens.data_ptr(), sim.data_ptr(),
                                attn.data_ptr(), attn1.data_ptr(), wctx.data_ptr(), dsim.data_ptr(), dfeat.data_ptr(),
                                None, B, nef, R, L, 5.0, 5.0, st)
    else:
        lib.sba_damsm_words_fwd.argtypes = [P] * 7 + [I] * 4 + [F, F, P]
        lib.sba_damsm_words_bwd.argtypes = [P] * 10 + [I] * 4 + [F, F, P]
        fwd = lambda: lib.sba_damsm_words_fwd(feat.data_ptr(), words.data_ptr(), lens.data_ptr(), sim.data_ptr(),
                                              attn.data_ptr(), attn1.data_ptr(), wctx.data_ptr(), B, nef, R, L, 5.0, 5.0, st)
        bwd = lambda: lib.sba_damsm_words_bwd(feat.data_ptr(), words.data_ptr(), lens.data_ptr(), sim.data_ptr(),
                                              attn.data_ptr(), attn1.data_ptr(), wctx.data_ptr(), dsim.data_ptr(),
                                              dfeat.data_ptr(), None, B, nef, R, L, 5.0, 5.0, st)
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print('damsm_words_%s: %.1f us' % (name, e0.elapsed_time(e1) * 100))


if __name__ == '__main__':
    main()
