#!/usr/bin/env python3
"""Run tests/test_step_gpu.py::test_generator_forward_backward[variant, f32] N times in one process (a rare failure of
the mix variant was seen once in 16 full-suite runs): python tools/stress_generator_test.py [N] [variant]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sba-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import test_step_gpu as T  # noqa: E402
from miscc.config import cfg, reset_cfg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
variant = sys.argv[2] if len(sys.argv) > 2 else 'mix'
dev = torch.device('cuda:0')
fails = 0
for k in range(n):
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3, s.LAMBDA = 4.0, 5.0, 10.0, 5.0
    try:
        T.test_generator_forward_backward(dev, torch.float32, variant)
    except AssertionError as e:
        fails += 1
        print('run %d FAILED: %s' % (k, str(e)[:200]), flush=True)
print('%d / %d failed (SBA_FORK_MAPPING=%s)' % (fails, n, os.environ.get('SBA_FORK_MAPPING', '1')))
