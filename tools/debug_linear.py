import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch
from sbagan._lib import call
dev = torch.device('cuda:0')
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for (B, K, N) in ((20, 256, 400), (20, 100, 256), (5, 256, 64), (3, 200, 16384)):
    x = torch.randn(B, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    y = torch.empty(B, N, device=dev)
    call('sba_linear_fwd', x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, K, N, st)
    ref = (x.double() @ w.double().t() + b.double())
    dy = torch.randn(B, N, device=dev)
    dx = torch.empty(B, K, device=dev); dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    call('sba_linear_bwd', x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, K, N, st)
    torch.cuda.synchronize()
    e = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print('B%d K%d N%d: y %.2e  dx %.2e  dw %.2e  db %.2e' % (B, K, N, e(y, ref), e(dx, dy.double() @ w.double()),
          e(dw, dy.double().t() @ x.double()), e(db, dy.double().sum(0))))
