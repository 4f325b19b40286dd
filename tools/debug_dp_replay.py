#!/usr/bin/env python3
"""ReplayedStepDP against the eager data-parallel step, one rank (gloo), deterministic mode: which quantities differ after
ONE step from the same state."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sba-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29611')
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
dist.init_process_group('gloo', rank=0, world_size=1)
from dist_worker import build  # noqa: E402
from helpers import rel_l2  # noqa: E402
from miscc.config import cfg, reset_cfg  # noqa: E402
from sbagan import ops  # noqa: E402
from sbagan.synth import synthetic_batch  # noqa: E402
from sbagan.trainer import ReplayedStepDP  # noqa: E402
reset_cfg()
cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
ops.set_compute_dtype(torch.float32)
ops.set_deterministic(True)
B = 4
dp = build(dev, B, True)
b = synthetic_batch(B, device=dev, seed=100)
g = torch.Generator().manual_seed(3)
noise = torch.randn((B, 100), generator=g).to(dev)
eps = torch.randn((B, 100), generator=g).to(dev)
orig = dp.phase_a
dp.phase_a = lambda se, we, m, nz, e=None: orig(se, we, m, nz, eps)
args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
dp.overlap_g = dp.bucket_d = False
dp.step(*args)
torch.cuda.synchronize()
snap = dp.snapshot()


def state(out):
    r = {'loss/' + k: v.detach().float().reshape(1).clone() for k, v in out.items() if torch.is_tensor(v)}
    for i, f in enumerate([dp.flatG] + dp.flatD):
        r['grad/%d' % i], r['data/%d' % i] = f.grad.clone(), f.data.clone()
    return r


nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dp.restore(snap)
for _ in range(nsteps):
    out = dp.step(*args)
torch.cuda.synchronize()
want = state(out)
keep = noise.clone()
rdp = ReplayedStepDP(dp, *args)
rdp.draw = False
noise.copy_(keep)
dp.restore(snap)
rdp.resync()
for _ in range(nsteps):
    rdp.replay()
torch.cuda.synchronize()
got = state(rdp.out)
for k in sorted(want):
    if not torch.equal(want[k], got[k]):
        print('%-22s rel %.3e   %s' % (k, rel_l2(got[k], want[k]), (float(want[k]), float(got[k])) if want[k].numel() == 1 else ''))
print('done')
