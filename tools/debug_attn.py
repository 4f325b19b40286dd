import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch
from sbagan._lib import call
dev = torch.device('cuda:0')
B, idf, L, Q = 3, 32, 7, 144
torch.manual_seed(0)
h = torch.randn(B, Q, idf, device=dev)
src = torch.randn(B, idf, L, device=dev)
ctx = torch.zeros(B, Q, idf, device=dev)
att = torch.full((B, L, Q), -1.0, device=dev)
st = torch.cuda.current_stream().cuda_stream
call('sba_word_attn_fwd', 0, h.data_ptr(), src.data_ptr(), None, ctx.data_ptr(), att.data_ptr(), B, Q, idf, L, 1, idf, 0, st)
torch.cuda.synchronize()
s = torch.bmm(h, src)
a = torch.softmax(s, 2)
cref = torch.bmm(a, src.transpose(1, 2))
print('att err per (b,l):', (att - a.transpose(1, 2)).abs().amax(2))
print('ctx err per b:', (ctx - cref).abs().amax((1, 2)))
print(att[0, :, :4], a.transpose(1, 2)[0, :, :4])
mask = torch.zeros(B, L, dtype=torch.uint8, device=dev)
mask[1, 3:] = 1
mask[2, 5:] = 1
for mode in (0, 1):
    att.fill_(-1)
    call('sba_word_attn_fwd', 0, h.data_ptr(), src.data_ptr(), mask.data_ptr(), ctx.data_ptr(), att.data_ptr(), B, Q, idf, L, mode, idf, 0, st)
    torch.cuda.synchronize()
    if mode == 0:
        rows = (torch.arange(B * Q, device=dev) % B).view(B, Q)
    else:
        rows = torch.arange(B, device=dev).view(B, 1).expand(B, Q)
    m = mask[rows].bool()          # B,Q,L
    a = torch.softmax(s.masked_fill(m, float('-inf')), 2)
    print('mode', mode, 'att err per (b,l):', (att - a.transpose(1, 2)).abs().amax(2))
    print('zero pattern equal:', torch.equal(att == 0, a.transpose(1, 2) == 0))
    bad = ((att == 0) != (a.transpose(1, 2) == 0)).nonzero()
    print(bad[:10])
