#!/usr/bin/env python3
"""Time CNN_ENCODER (Inception-v3 via PyTorch-ROCm/MIOpen) forward + backward-data at B=20 under
different dtype / memory-format settings (tuning aid for bench.py's default)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

from miscc.config import cfg  # noqa: E402
import model  # noqa: E402


def run(tag, dtype, cl, bench):
    torch.backends.cudnn.benchmark = bench
    dev = torch.device('cuda:0')
    enc = model.CNN_ENCODER(256).to(dev).eval()
    for p in enc.parameters():
        p.requires_grad = False
    if cl:
        enc = enc.to(memory_format=torch.channels_last)
    x = torch.rand(20, 3, 256, 256, device=dev) * 2 - 1

    def step():
        xi = x.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=dtype, enabled=dtype is not None):
            f, c = enc(xi)
        (f.float().sum() + c.float().sum()).backward()
        return xi.grad

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print('%-32s %.2f ms fwd+bwd' % (tag, (time.perf_counter() - t0) / n * 1e3), flush=True)


if __name__ == '__main__':
    for tag, dt, cl, b in (('bf16 channels_last', torch.bfloat16, True, False), ('bf16 nchw', torch.bfloat16, False, False),
                           ('fp16 nchw', torch.float16, False, False), ('fp32 nchw', None, False, False),
                           ('fp16 channels_last', torch.float16, True, False),
                           ('bf16 nchw benchmark', torch.bfloat16, False, True)):
        try:
            run(tag, dt, cl, b)
        except Exception as e:
            print(tag, 'failed:', type(e).__name__, e)
