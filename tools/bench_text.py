#!/usr/bin/env python3
"""Times RNN_ENCODER.forward (bird_style.yml size: B=20, T=18, ninput 300, nhidden 256) through the HIP path
(sba_lstm_bidir_fwd, sync-free) and through the module's torch path (MIOpen packed LSTM + cap_lens.tolist())."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch
import model

dev = torch.device('cuda:0')
B, T, ntoken = 20, 18, 5450
net = model.RNN_ENCODER(ntoken, nhidden=256).to(dev).eval()
cap = torch.randint(1, ntoken, (B, T), device=dev)
lens = torch.sort(torch.randint(5, T + 1, (B,), device=dev), descending=True)[0]
for b in range(B):
    cap[b, int(lens[b]):] = 0
hid = net.init_hidden(B)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


with torch.no_grad():
    net.use_hip = True
    t_hip = timeit(lambda: net(cap, lens, hid))
    net.use_hip = False
    t_torch = timeit(lambda: net(cap, lens, hid))
print('RNN_ENCODER forward B=%d T=%d: HIP %.1f us, torch/MIOpen path %.1f us (wall per call, incl. host)' % (B, T, t_hip, t_torch))
