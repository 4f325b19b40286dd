"""Rank body of tests/test_dist_gpu.py (one process per rank, all ranks on cuda:0, gloo rendezvous on
127.0.0.1).  Checks SURVEY.md 8e's statement through GANStep / FusedAdam(grad_scale = 1/world):

  * rank r's discriminator losses and LOCAL gradients equal a single-process GANStep on rank r's batch;
  * the gradient every rank applies is the SUM over ranks of the local gradients (bit-exact), folded to the
    mean by the fused Adam: post-step parameters = Adam(step 1) of the mean gradient, identical on all ranks;
  * the same holds for the generator, whose local gradient is taken against the UPDATED discriminators.

    python tests/dist_worker.py RANK WORLD PORT OUTDIR
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sba-gan_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(dev, B, distributed):
    import model
    from helpers import FULL, d_shapes, g_shapes
    from oracle import fill
    from sbagan.trainer import GANStep
    netG = model.G_NET()
    netG.load_state_dict(fill.fill_state_dict(g_shapes(FULL, 3, 'model')))
    netsD = [model.D_NET64(), model.D_NET128(), model.D_NET256()]
    for i, d in enumerate(netsD):
        d.load_state_dict(fill.fill_state_dict(d_shapes(FULL, i), salt=i))
    netG.to(dev).train()
    for d in netsD:
        d.to(dev).train()
    netG.set_return_attention(False)
    enc = fill.StandInImageEncoder(256, device=dev)
    return GANStep(netG, netsD, enc, B, lr_g=2e-4, lr_d=2e-4, distributed=distributed)


def adam_first_step(p0, g, lr=2e-4, b1=0.5, b2=0.999, eps=1e-8):
    """torch.optim.Adam step 1 from zero moments (trainer.py:136-143), in float64."""
    g = g.double()
    m, v = (1 - b1) * g, (1 - b2) * g * g
    denom = v.sqrt() / (1 - b2) ** 0.5 + eps
    return p0.double() - (lr / (1 - b1)) * m / denom


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import FULL, make_inputs, rel_l2
    from miscc.config import cfg, reset_cfg
    from oracle import fill
    from sbagan import ops
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3, s.LAMBDA = 4.0, 5.0, 10.0, 5.0
    ops.set_compute_dtype(torch.float32)
    # deterministic reductions: "equal" below then means BIT-equal (two single-process runs from the same state, and the
    # data-parallel rank against the single-process run on the same batch); round 2 ran this in the default mode and had
    # to allow for a bimodal ~2e-3 run-to-run difference of a discriminator gradient
    ops.set_deterministic(True)
    B = 4
    x = make_inputs(FULL, B, 18, lmax=18, tag=500 + 100 * rank)       # every rank its own batch
    imgs = [i.to(dev) for i in x['imgs']]
    sent, words, mask, lens = x['sent'].to(dev), x['words'].to(dev), x['mask'].to(dev), x['cap_lens'].to(dev)
    noise, eps = fill.unit((B, 100), 550 + rank).to(dev), fill.unit((B, 100), 560 + rank).to(dev)
    args = (imgs, sent, words, mask, lens, x['class_ids'], noise, eps)

    dp = build(dev, B, True)
    assert dp.distributed and dp.world == world
    solo = build(dev, B, False)
    p0 = [f.data.clone() for f in [dp.flatG] + dp.flatD]
    local = {}
    orig_start = dp.exchange.start

    def start(flat_grad):                 # keep this rank's gradient as it was BEFORE the exchange
        local[flat_grad.data_ptr()] = flat_grad.clone()
        return orig_start(flat_grad)
    dp.exchange.start = start
    out_dp = dp.step(*args)
    snap = solo.snapshot()
    out_solo = solo.step(*args)
    torch.cuda.synchronize()
    # a second single-process run from the same state: bit-identical in deterministic mode
    solo_grads = [f.grad.clone() for f in [solo.flatG] + solo.flatD]
    solo.restore(snap)
    solo.step(*args)
    torch.cuda.synchronize()
    noise = [rel_l2(f.grad, g) for f, g in zip([solo.flatG] + solo.flatD, solo_grads)]
    fails = []

    def expect(cond, msg):
        if not cond:
            fails.append(msg)

    expect(all(n == 0.0 for n in noise), 'two single-process runs from the same state differ in deterministic mode: %r' % noise)

    flats_dp, flats_solo = [dp.flatG] + dp.flatD, [solo.flatG] + solo.flatD
    for i in range(3):
        a, b = float(out_dp['errD%d' % i]), float(out_solo['errD%d' % i])
        expect(abs(a - b) <= 1e-5 * abs(b), 'errD%d: data-parallel %r vs single-process %r' % (i, a, b))
    for k, (fd, fs) in enumerate(zip(flats_dp, flats_solo)):
        name = 'G' if k == 0 else 'D%d' % (k - 1)
        mine = local[fd.grad.data_ptr()]
        if k > 0:       # discriminators: the local gradient IS the single-process gradient on this batch
            r = rel_l2(mine, solo_grads[k])
            expect(torch.equal(mine, solo_grads[k]), '%s: local gradient differs from the single-process run (rel L2 '
                   '%.2e, single-process run-to-run difference %.2e)' % (name, r, noise[k]))
        gathered = [torch.empty_like(mine).cpu() for _ in range(world)]
        dist.all_gather(gathered, mine.cpu())
        total = gathered[0].clone()
        for t in gathered[1:]:
            total += t
        expect(torch.equal(fd.grad.cpu(), total), '%s: exchanged gradient is not the sum over ranks (max diff %.3e)'
               % (name, float((fd.grad.cpu() - total).abs().max())))
        want = adam_first_step(p0[k].cpu(), total / world)
        err = (fd.data.cpu().double() - want).abs()
        tol = 2e-8 + 1e-6 * want.abs()
        frac = float((err > tol).double().mean())
        # sign-like first update: an element whose mean gradient is at rounding level may flip; nothing else may differ
        expect(frac <= 1e-4 and float(err.max()) <= 2.0 * 2e-4 + 1e-7,
               '%s: parameters are not Adam(mean gradient): %.2e of the elements off, max err %.3e'
               % (name, frac, float(err.max())))
        both = [torch.empty_like(fd.data).cpu() for _ in range(world)]
        dist.all_gather(both, fd.data.cpu())
        expect(all(torch.equal(both[0], t) for t in both[1:]), '%s: replicas diverged after one step' % name)
    # generator: its local gradient is taken against the UPDATED (replica-identical) discriminators, so it
    # differs from the single-process run, whose discriminators moved by the local gradient only
    expect(bool(torch.isfinite(out_dp['errG_total'])), 'errG_total not finite')
    res = {'rank': rank, 'ok': not fails, 'fails': fails, 'solo_run_to_run_noise': noise,
           'errD': [float(out_dp['errD%d' % i]) for i in range(3)], 'errG_total': float(out_dp['errG_total'])}
    with open(os.path.join(outdir, 'rank%d.json' % rank), 'w') as f:
        json.dump(res, f, indent=1)
    dbg = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(dbg):
        with open(os.path.join(dbg, 'dist_worker_rank%d.json' % rank), 'w') as f:
            json.dump(res, f, indent=1)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if not fails else 1)


if __name__ == '__main__':
    main()
