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
    from test_determinism_gpu import _OrderedStandIn      # (torch's adaptive_avg_pool2d backward adds with atomics)
    enc = _OrderedStandIn(256, device=dev)
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
    noise_in = noise
    args = (imgs, sent, words, mask, lens, x['class_ids'], noise, eps)

    dp = build(dev, B, True)
    assert dp.distributed and dp.world == world
    solo = build(dev, B, False)
    # the single-process yardstick runs the data-parallel step's launch decomposition (real-image forwards ahead as
    # their own passes, bucketed backward passes), so that "equal" can mean bit-equal: on this fixture the discriminators'
    # logits are saturated (BCELoss's log clamp), and a 1e-7 difference in summation order between a grouped and a
    # two-pass BatchNorm statistics sum moves D_NET256's gradient by tens of percent
    solo.force_overlap_layout = True
    p0 = [f.data.clone() for f in [dp.flatG] + dp.flatD]
    local = {}
    orig_start = dp.exchange.start

    def start(flat_grad):                 # keep this rank's gradient as it was BEFORE the exchange (bucket by bucket)
        local[flat_grad.data_ptr()] = flat_grad.clone()
        return orig_start(flat_grad)
    dp.exchange.start = start
    assert dp.bucket_d and dp.overlap_g
    out_dp = dp.step(*args)
    # the generator's update is deferred to the next step (its exchange overlaps that step's real-image forwards)
    pG = dp.flatG.data.clone()
    expect_pending = dp._g_pending is not None
    dp.finish()
    assert expect_pending and not torch.equal(pG, dp.flatG.data), 'generator update was not deferred / not applied'

    def local_grad(flat):
        """this rank's pre-exchange gradient of a network, reassembled from its buckets"""
        base, esz = flat.grad.data_ptr(), flat.grad.element_size()
        out = torch.empty_like(flat.grad)
        covered = 0
        for ptr, t in local.items():
            o = (ptr - base) // esz
            if 0 <= o < flat.grad.numel() and (ptr - base) % esz == 0 and o + t.numel() <= flat.grad.numel():
                out[o:o + t.numel()] = t
                covered += t.numel()
        assert covered == flat.grad.numel(), (covered, flat.grad.numel())
        return out
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
        mine = local_grad(fd)
        if k > 0:       # discriminators: the local gradient IS the single-process gradient on this batch
            r = rel_l2(mine, solo_grads[k])
            expect(torch.equal(mine, solo_grads[k]), '%s: local gradient differs from the single-process run (rel L2 '
                   '%.2e, single-process run-to-run difference %.2e)' % (name, r, noise[k]))
            if k > 1:   # D_NET128 / D_NET256 went out in two buckets
                expect(sum(1 for p in local if fd.grad.data_ptr() <= p < fd.grad.data_ptr() + fd.grad.numel() * 4) == 2,
                       '%s: expected two gradient buckets' % name)
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
    # ---- the same data-parallel step replayed from per-phase hipGraphs (the launch mode bench.py uses for N > 1): the
    # real-image forwards as their own graph ahead of the deferred generator update, every bucketed discriminator as
    # two backward graphs with the tail bucket's all-reduce between them -- two replays + finish() must leave every
    # parameter, Adam moment and BatchNorm buffer bit-identical to two eager data-parallel steps from the same state
    from sbagan.trainer import GraphedStep
    orig_a = dp.phase_a
    dp.phase_a = lambda se, we, m, nz, e=None: orig_a(se, we, m, nz, eps)        # fixed eps, eager and captured
    dp.exchange.start = orig_start
    gargs = (imgs, sent, words, mask, lens, x['class_ids'], noise_in)
    snap_dp = dp.snapshot()
    dp.early_damsm = False          # the per-phase graphs keep the ranking terms inside the generator-loss phase

    def state_names():
        names = []
        for k, f in enumerate(flats_dp):
            nm = 'G' if k == 0 else 'D%d' % (k - 1)
            names += ['%s.%s' % (nm, t) for t in ('data', 'm', 'v', 'grad')]
        for k, net in enumerate([dp.netG] + dp.netsD):
            names += ['%s.%s' % ('G' if k == 0 else 'D%d' % (k - 1), n) for n, _ in net.named_buffers()]
        return names

    def state():
        return [t for f in flats_dp for t in (f.data, f.m, f.v, f.grad)] + \
               [b for net in [dp.netG] + dp.netsD for _, b in net.named_buffers()]
    eager_out = None
    for _ in range(2):
        eager_out = dp.step(*gargs)
    dp.finish()
    dp.early_damsm = True
    torch.cuda.synchronize()
    want = [t.clone() for t in state()]
    want_out = {k: float(v) for k, v in eager_out.items()}
    graph = GraphedStep(dp, *gargs)
    expect(graph.gPre is not None and any(g2 is not None for g2 in graph.gD2), 'overlap graphs were not captured')
    dp.restore(snap_dp)
    graph.resync()
    for _ in range(2):
        graph.replay()
    graph.finish()
    torch.cuda.synchronize()
    got = state()
    names = state_names()
    bad = [(names[i], rel_l2(a.float(), b.float())) for i, (a, b) in enumerate(zip(got, want)) if not torch.equal(a, b)]
    got_out = {k: float(v) for k, v in graph.out.items()}
    bad_out = [(k, got_out[k], want_out[k]) for k in want_out if got_out[k] != want_out[k]]
    expect(not bad and not bad_out, 'hipGraph replay of the data-parallel step differs from the eager data-parallel step '
           'in %d tensors: %r; losses %r' % (len(bad), bad[:12], bad_out[:6]))
    # ---- the data-parallel step from three native recordings with the exchange between them (ReplayedStepDP: one grouped
    # pass and one bucket per discriminator, nothing deferred): two replays must leave every parameter, Adam moment and
    # BatchNorm buffer bit-identical to two eager data-parallel steps with the same decomposition
    from sbagan.trainer import ReplayedStepDP
    dp.restore(snap_dp)
    graph.resync()
    dp.overlap_g = dp.bucket_d = False
    for _ in range(2):
        eager_out = dp.step(*gargs)
    dp.finish()
    torch.cuda.synchronize()
    want = [t.clone() for t in state()]
    want_out = {k: float(v) for k, v in eager_out.items()}
    noise_keep = noise_in.clone()
    rdp = ReplayedStepDP(dp, *gargs)        # (its warm-up step draws fresh noise into the static tensor)
    rdp.draw = False
    rdp.eps.copy_(eps)
    noise_in.copy_(noise_keep)
    dp.restore(snap_dp)
    rdp.resync()
    for _ in range(2):
        rdp.replay()
    torch.cuda.synchronize()
    got = state()
    bad = [(names[i], rel_l2(a.float(), b.float())) for i, (a, b) in enumerate(zip(got, want)) if not torch.equal(a, b)]
    got_out = {k: float(v) for k, v in rdp.out.items()}
    bad_out = [(k, got_out[k], want_out[k]) for k in want_out if got_out[k] != want_out[k]]
    expect(not bad and not bad_out, 'ReplayedStepDP differs from the eager data-parallel step in %d tensors: %r; losses %r'
           % (len(bad), bad[:12], bad_out[:6]))
    # ... and with the image encoder + DAMSM terms recorded on their own, launched while the exchange is in flight
    rdp2 = ReplayedStepDP(dp, *gargs, e_beside_exchange=True)
    rdp2.draw = False
    rdp2.eps.copy_(eps)
    noise_in.copy_(noise_keep)
    dp.restore(snap_dp)
    rdp2.resync()
    for _ in range(2):
        rdp2.replay()
    torch.cuda.synchronize()
    got = state()
    bad = [(names[i], rel_l2(a.float(), b.float())) for i, (a, b) in enumerate(zip(got, want)) if not torch.equal(a, b)]
    expect(not bad, 'ReplayedStepDP(e_beside_exchange) differs from the eager data-parallel step in %d tensors: %r'
           % (len(bad), bad[:12]))
    # ---- bench.py's DEFAULT for N > 1: the WHOLE data-parallel step as ONE recording, the exchanges and the deferred generator
    # update as host-call nodes of the replayer (same decomposition as the eager data-parallel step: two-pass discriminator
    # loss, D_NET128 / D_NET256 in two buckets): two replays + finish() bit-identical to two eager steps from the same state
    from sbagan.trainer import ReplayedStep
    dp.restore(snap_dp)
    rdp2.resync()
    dp.overlap_g, dp.bucket_d = True, True
    for _ in range(2):
        eager_out = dp.step(*gargs)
    dp.finish()
    torch.cuda.synchronize()
    want = [t.clone() for t in state()]
    want_out = {k: float(v) for k, v in eager_out.items()}
    rs = ReplayedStep(dp, *gargs)
    expect(rs.info['host_calls'] >= 10, 'the recording holds %d host-call nodes' % rs.info['host_calls'])
    rs.draw = False
    rs.eps.copy_(eps)
    noise_in.copy_(noise_keep)
    dp.restore(snap_dp)
    rs.resync()
    rs.replay()
    expect(dp._g_pending is not None, 'single recording: the generator update was not left pending')
    rs.replay()
    rs.finish()
    torch.cuda.synchronize()
    got = state()
    bad = [(names[i], rel_l2(a.float(), b.float())) for i, (a, b) in enumerate(zip(got, want)) if not torch.equal(a, b)]
    got_out = {k: float(v) for k, v in rs.out.items()}
    bad_out = [(k, got_out[k], want_out[k]) for k in want_out if got_out[k] != want_out[k]]
    expect(not bad and not bad_out, 'ReplayedStep (one recording, host-call exchanges) differs from the eager data-parallel '
           'step in %d tensors: %r; losses %r' % (len(bad), bad[:12], bad_out[:6]))
    # ---- the single recording with the generator's exchange WAITED FOR where it is issued (bench.py: SBA_DP_OVERLAP_G=0; the
    # discriminator loss as one grouped real|fake pass): no update is ever pending, so the recording must not hold the
    # deferred update's launches (it would apply Adam twice), and two replays equal two eager steps of that layout
    dp.restore(snap_dp)
    rs.resync()
    dp.overlap_g, dp.bucket_d = False, True
    for _ in range(2):
        eager_out = dp.step(*gargs)
    torch.cuda.synchronize()
    want = [t.clone() for t in state()]
    want_out = {k: float(v) for k, v in eager_out.items()}
    rs2 = ReplayedStep(dp, *gargs)
    expect(not rs2._rec.recorded_update and dp._g_pending is None,
           'exposed generator exchange: the recording holds a deferred update')
    rs2.draw = False
    rs2.eps.copy_(eps)
    noise_in.copy_(noise_keep)
    dp.restore(snap_dp)
    rs2.resync()
    for _ in range(2):
        rs2.replay()
    torch.cuda.synchronize()
    got = state()
    bad = [(names[i], rel_l2(a.float(), b.float())) for i, (a, b) in enumerate(zip(got, want)) if not torch.equal(a, b)]
    got_out = {k: float(v) for k, v in rs2.out.items()}
    bad_out = [(k, got_out[k], want_out[k]) for k in want_out if got_out[k] != want_out[k]]
    expect(not bad and not bad_out, 'ReplayedStep (one recording, exposed generator exchange) differs from the eager '
           'data-parallel step in %d tensors: %r; losses %r' % (len(bad), bad[:12], bad_out[:6]))
    del rs2
    # ---- ReplayedStepDP with the generator's exchange deferred behind the next step's text encoder + real-image
    # forwards (recording R0), image encoder + DAMSM terms beside the discriminators' exchange.  Same bits as two eager
    # data-parallel steps with the same decomposition (two-pass discriminator loss, one bucket), and the ORDER north_star
    # asks for: in the second replay the real-image forwards are issued BEFORE the pending all-reduce is waited for.
    dp.restore(snap_dp)
    rdp2.resync()
    dp.overlap_g, dp.bucket_d = True, False
    for _ in range(2):
        eager_out = dp.step(*gargs)
    dp.finish()
    torch.cuda.synchronize()
    want = [t.clone() for t in state()]
    rdp3 = ReplayedStepDP(dp, *gargs, e_beside_exchange=True, defer_g=True)
    expect(dp.overlap_g and not dp.bucket_d and len(rdp3.replayers) == 5, 'deferred-update recordings were not captured')
    rdp3.draw = False
    rdp3.eps.copy_(eps)
    noise_in.copy_(noise_keep)
    dp.restore(snap_dp)
    rdp3.resync()
    log = []
    r0 = rdp3.replayers[0]
    orig_r0, orig_wait = r0.replay, dp._allreduce_wait
    r0.replay = lambda: (log.append('real_forwards'), orig_r0())[1]

    def logged_wait(h):
        log.append('wait')
        return orig_wait(h)
    dp._allreduce_wait = logged_wait
    rdp3.replay()
    expect(rdp3._pending is not None, 'the generator update of a deferred replay was applied inside replay()')
    del log[:]
    rdp3.replay()
    rdp3.finish()
    dp._allreduce_wait = orig_wait
    expect(log[:2] == ['real_forwards', 'wait'], 'the real-image forwards were not issued ahead of the wait for the '
           'generator exchange in flight: %r' % (log[:6],))
    torch.cuda.synchronize()
    got = state()
    bad = [(names[i], rel_l2(a.float(), b.float())) for i, (a, b) in enumerate(zip(got, want)) if not torch.equal(a, b)]
    expect(not bad, 'ReplayedStepDP(defer_g) differs from the eager data-parallel step in %d tensors: %r' % (len(bad), bad[:12]))
    # restore()/resync() DROP a pending update instead of applying it to the restored weights
    rdp3.replay()
    expect(rdp3._pending is not None, 'no pending update after a deferred replay')
    dp.restore(snap_dp)
    rdp3.resync()
    expect(rdp3._pending is None and dp._g_pending is None, 'restore / resync left a stale generator update pending')
    torch.cuda.synchronize()
    expect(torch.equal(flats_dp[0].data, snap_dp[0]['data']) and torch.equal(dp.optG.state, snap_dp[0]['state']),
           'the generator changed after restore(): a stale update was applied')
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
