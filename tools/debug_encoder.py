import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from oracle import fill
from helpers import rel_l2
import model
from sbagan import ops
from sbagan.inception_hip import InceptionHIP
ops.set_compute_dtype(torch.float32)
dev = torch.device('cuda:0')
enc = model.CNN_ENCODER(256).eval()
P = fill.fill_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, gain=1.6)
for k in P:
    if k.endswith('running_mean'): P[k] = 0.1 * fill.uniform(tuple(P[k].shape), fill.tag_of(k))
    elif k.endswith('running_var'): P[k] = 1.0 + 0.3 * fill.uniform(tuple(P[k].shape), fill.tag_of(k) + 1)
enc.load_state_dict(P)
B = 2
img = fill.uniform((B, 3, 256, 256), 77)
acts = {}
def hook(name):
    def h(m, i, o):
        o.retain_grad(); acts[name] = o
    return h
for n in ['Conv2d_1a_3x3','Conv2d_2a_3x3','Conv2d_2b_3x3','Conv2d_3b_1x1','Conv2d_4a_3x3','Mixed_5b','Mixed_5c','Mixed_5d','Mixed_6a','Mixed_6b','Mixed_6c','Mixed_6d','Mixed_6e','Mixed_7a','Mixed_7b','Mixed_7c']:
    getattr(enc, n).register_forward_hook(hook(n))
xr = img.clone().requires_grad_(True)
fr, cr = enc(xr)
gfe, gco = fill.unit(tuple(fr.shape), 78), fill.unit(tuple(cr.shape), 79)
((fr * gfe).sum() + (cr * gco).sum()).backward()
ref_g = {k: v.grad.clone() for k, v in acts.items()}
ref_a = {k: v.detach().clone() for k, v in acts.items()}
enc_g = enc.to(dev)
run = InceptionHIP(enc_g); run.keep_debug = True
xa = img.to(dev).requires_grad_(True)
f, c = run(xa)
((f * gfe.to(dev)).sum() + (c * gco.to(dev)).sum()).backward()
torch.cuda.synchronize()
for n in ref_a:
    a = run.named[n]
    C = ref_a[n].shape[1]
    mine = a.t[..., a.coff:a.coff + C].permute(0, 3, 1, 2).float().cpu()
    g = run._grads[id(a.t)][0][..., a.coff:a.coff + C].permute(0, 3, 1, 2).float().cpu()
    print('%-16s act %.2e   grad %.2e' % (n, rel_l2(mine, ref_a[n]), rel_l2(g, ref_g[n])))
print('dimg', rel_l2(xa.grad, xr.grad))
