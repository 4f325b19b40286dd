#!/usr/bin/env python3
"""Time the HIP Inception-v3 trunk (sbagan.inception_hip) forward + backward-data at B = 20, 256 px: eager launches and
ONE hipGraph replay (the form it takes inside the benched step).  SBA_ENC_GROUP=0/1 switches the grouped per-level launches.
    python tools/bench_encoder_hip.py [bf16|f32]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402

import model  # noqa: E402
from sbagan import ops  # noqa: E402
from sbagan.inception_hip import InceptionHIP  # noqa: E402


def main():
    dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else torch.bfloat16
    dev = torch.device('cuda:0')
    ops.set_compute_dtype(dt)
    torch.manual_seed(101)
    enc = model.CNN_ENCODER(256).to(dev).eval()
    run = InceptionHIP(enc)
    x = torch.rand(20, 3, 256, 256, device=dev) * 2 - 1
    gf, gc = torch.randn(20, 256, 17, 17, device=dev), torch.randn(20, 256, device=dev)
    out = {}

    def step():
        xi = x.detach().requires_grad_(True)
        f, c = run(xi)
        (g,) = torch.autograd.grad([f, c], xi, [gf, gc])
        out['g'] = g
        return g

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print('eager  fwd+bwd %.3f ms' % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
    ref = out['g'].clone()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        step()
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
        res = step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print('graph  fwd+bwd %.3f ms   (replay vs eager gradient rel L2 %.2e)'
          % (e0.elapsed_time(e1) / 20, float((res - ref).norm() / ref.norm())), flush=True)


if __name__ == '__main__':
    main()
