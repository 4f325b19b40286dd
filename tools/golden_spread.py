#!/usr/bin/env python3
"""Run-to-run spread of one golden-step case against the reference's numbers: the GPU test's own code
(tests/test_step_gpu.py::_golden_case) N times with its tolerances opened up, collecting the relative deviations it
reports.  The committed tolerances are set from this spread.
python tools/golden_spread.py bert_b4 bfloat16 eager 12 [det|default]     (det: the deterministic-reduction mode, the
golden tests' own mode; default: f32 atomics / split-K, what bench.py times)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
import torch  # noqa: E402


def main():
    case, dtn, launch, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    det = not (len(sys.argv) > 5 and sys.argv[5] == 'default')
    import test_step_gpu as T
    from miscc.config import cfg, reset_cfg
    dt = getattr(torch, dtn)
    for d in (T.LOSS_TOL, T.LOSS_TOL_B20, T.LOSS_TOL_B20_DEFAULT_MODE, T.GNORM_G_TOL):
        for k in d:
            d[k] = 1e9
    T.check = lambda *a, **k: None
    dev = torch.device('cuda:0')
    golden = os.path.join(ROOT, 'tests', 'golden')
    rows = []
    for it in range(n):
        reset_cfg()
        cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
        cfg.TRAIN.SMOOTH.GAMMA1, cfg.TRAIN.SMOOTH.GAMMA2, cfg.TRAIN.SMOOTH.GAMMA3 = 4.0, 5.0, 10.0
        cfg.TRAIN.SMOOTH.LAMBDA = 5.0
        try:
            T._golden_case(dev, dt, case, launch, golden, det=det)
        except AssertionError as e:
            print('run %d: assertion %s' % (it, str(e)[:200]))
        rep = os.path.join(ROOT, 'gpurun_out', 'parity_report_%s_%s_%s%s.json'
                           % (case, dtn, launch, '' if det else '_default_mode'))
        rows.append(json.load(open(rep)))
    keys = sorted(rows[0])
    print('%-16s %10s %10s %10s' % ('quantity', 'min', 'median', 'max'))
    for k in keys:
        v = sorted(r[k] for r in rows)
        print('%-16s %10.3e %10.3e %10.3e' % (k, v[0], v[len(v) // 2], v[-1]))


if __name__ == '__main__':
    main()
