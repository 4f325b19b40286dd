import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch
import model
from sbagan import ops
from sbagan.inception_hip import InceptionHIP
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
enc = model.CNN_ENCODER(256).to(dev).eval()
run = InceptionHIP(enc)
run.parallel = os.environ.get('PAR', '0') == '1'
x = torch.rand(20, 3, 256, 256, device=dev) * 2 - 1
def step():
    xi = x.clone().requires_grad_(True)
    f, c = run(xi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (f.sum() + c.sum()).backward()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
tb = 0
for _ in range(5): tb += step()
torch.cuda.synchronize(); tt = time.perf_counter() - t0
print('hip encoder (parallel=%s) fwd %.2f ms  bwd %.2f ms' % (run.parallel, (tt - tb) / 5 * 1e3, tb / 5 * 1e3))
