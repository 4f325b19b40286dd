import os, sys, time, faulthandler
faulthandler.enable()
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
run.parallel = (sys.argv[1] == '1') if len(sys.argv) > 1 else True
x = torch.rand(20, 3, 256, 256, device=dev) * 2 - 1
xi = x.clone().requires_grad_(True)
def step():
    xi.grad = None
    f, c = run(xi)
    (f.sum() + c.sum()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print('capturing', flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        step()
    print('captured', flush=True)
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); print('graph replay ms', (time.perf_counter() - t0) / 10 * 1e3)
except Exception as e:
    print('EXC', type(e).__name__, e, flush=True)
