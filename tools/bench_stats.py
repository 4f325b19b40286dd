import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch
from sbagan import ops
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (kind, cin, cout, h) in (('3x3up', 64, 64, 128), ('3x3', 64, 128, 128), ('3x3', 64, 64, 128), ('3x3up', 64, 64, 64), ('4x4s2', 64, 128, 128)):
    x = torch.randn((20, cin, h, h), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter((torch.randn((cout, cin, 3 if kind != '4x4s2' else 4, 3 if kind != '4x4s2' else 4), device=dev) / 24).contiguous(memory_format=torch.channels_last))
    pw = ops.PackedWeight(w)
    a = t(lambda: ops.conv_forward(x, pw, kind, want_stats=False))
    b = t(lambda: ops.conv_forward(x, pw, kind, want_stats=True))
    print('%-6s %3d->%3d @%3d  no stats %7.1f us   with stats %7.1f us' % (kind, cin, cout, h, a, b))
