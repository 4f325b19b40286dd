#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied: its modules are imported in-process (easydict /
torchvision / pytorch_pretrained_bert are absent here and are replaced by
empty in-process stand-ins, as SURVEY.md section 8c describes), fed closed-form
inputs from oracle/fill.py, and only numeric outputs are written.

    python tools/make_golden.py            # all fixtures
    python tools/make_golden.py units      # just the reduced-size units
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/AttnGAN2/code'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import fill  # noqa: E402


def _install_stubs():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)
            super().__setattr__(k, v)
        __setitem__ = __setattr__
    m = types.ModuleType('easydict')
    m.EasyDict = EasyDict
    sys.modules['easydict'] = m
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm
    pb = types.ModuleType('pytorch_pretrained_bert')
    pb.BertModel = object
    pb.BertTokenizer = object
    sys.modules['pytorch_pretrained_bert'] = pb


def load_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    from miscc.config import cfg
    cfg.CUDA = False
    import GlobalAttention
    import model
    import model_bert
    from miscc import losses
    return cfg, GlobalAttention, model, model_bert, losses


def set_dims(cfg, d):
    cfg.GAN.GF_DIM = d['ngf']
    cfg.GAN.DF_DIM = d['ndf']
    cfg.TEXT.EMBEDDING_DIM = d['nef']
    cfg.GAN.CONDITION_DIM = d['ncf']
    cfg.GAN.Z_DIM = d['nz']
    cfg.GAN.W_DIM = d['nw']
    cfg.GAN.R_NUM = d.get('rnum', 2)
    cfg.TREE.BRANCH_NUM = d.get('branch', 3)
    cfg.TRAIN.SMOOTH.GAMMA1 = 4.0
    cfg.TRAIN.SMOOTH.GAMMA2 = 5.0
    cfg.TRAIN.SMOOTH.GAMMA3 = 10.0
    cfg.TRAIN.SMOOTH.LAMBDA = 5.0


TINY = dict(ngf=8, ndf=8, nef=16, ncf=10, nz=12, nw=16)
FULL = dict(ngf=32, ndf=64, nef=256, ncf=100, nz=100, nw=256)


def load_filled(net, salt=0):
    sd = net.state_dict()
    P = fill.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, salt=salt)
    net.load_state_dict(P)
    return P


def summarize(t, full_limit=70000, nsample=2048):
    t = t.detach().double().flatten()
    d = {'sum': np.float64(t.sum()), 'sumsq': np.float64((t * t).sum()),
         'numel': np.int64(t.numel())}
    if t.numel() <= full_limit:
        d['full'] = t.float().numpy()
    else:
        stride = max(1, t.numel() // nsample)
        d['stride'] = np.int64(stride)
        d['sample'] = t[::stride][:nsample].float().numpy()
    return d


def put(store, name, t):
    for k, v in summarize(t).items():
        store['%s/%s' % (name, k)] = v


def make_inputs(d, B, L, branch=3, lmax=None, tag=100):
    lmax = lmax or L
    caps, lens = fill.synthetic_captions(B, words_num=L + 2, lmax=lmax, tag=tag)
    x = dict(
        z=fill.unit((B, d['nz']), tag + 2),
        z2=fill.unit((2, B, d['nz']), tag + 3),
        sent=fill.unit((B, d['nef']), tag + 4),
        words=fill.unit((B, d['nef'], lmax), tag + 5),
        imgs=[fill.uniform((B, 3, 64 * 2 ** i, 64 * 2 ** i), tag + 10 + i) for i in range(branch)],
        captions=caps, cap_lens=lens,
        class_ids=np.arange(B),
    )
    x['mask'] = (caps == 0)[:, :lmax]
    return x


# ---------------------------------------------------------------- units
def gen_units(ref):
    cfg, GA, model, model_bert, losses = ref
    set_dims(cfg, TINY)
    d = TINY
    B, L = 3, 6
    S = {}
    x = make_inputs(d, B, L, tag=100)

    # --- word attention with the mask quirk (GlobalAttention.py:82-121)
    att = GA.GlobalAttentionGeneral(d['ngf'], d['nef'])
    load_filled(att)
    h = fill.unit((B, d['ngf'], 8, 8), 201).requires_grad_(True)
    att.applyMask(x['mask'])
    ctx, a = att(h, x['words'])
    dctx = fill.unit(tuple(ctx.shape), 202)
    gh, gw = torch.autograd.grad((ctx * dctx).sum(), [h, att.conv_context.weight])
    put(S, 'attn/ctx', ctx); put(S, 'attn/att', a); put(S, 'attn/gh', gh); put(S, 'attn/gw', gw)
    S['attn/att_is_zero'] = (a.detach() == 0).numpy()        # integer/bool part of the quirk
    S['attn/mask'] = x['mask'].numpy()

    # --- func_attention / words_loss / sent_loss (losses.py:20-132)
    feat = fill.unit((B, d['nef'], 17, 17), 211).requires_grad_(True)
    words = x['words'].clone().requires_grad_(True)
    wc, at = GA.func_attention(words, feat, 4.0)
    put(S, 'funcattn/wctx', wc); put(S, 'funcattn/att', at)
    labels = torch.arange(B)
    for cname, cids in (('', np.arange(B)), ('_sameclass', np.array([0, 1, 0]))):
        w0, w1, _ = losses.words_loss(feat, words, labels, x['cap_lens'], cids, B)
        gf, gq = torch.autograd.grad(w0 + w1, [feat, words])
        S['words_loss%s/w0' % cname] = np.float64(w0.item())
        S['words_loss%s/w1' % cname] = np.float64(w1.item())
        put(S, 'words_loss%s/gfeat' % cname, gf); put(S, 'words_loss%s/gwords' % cname, gq)
        code = fill.unit((B, d['nef']), 212).requires_grad_(True)
        sent = x['sent'].clone().requires_grad_(True)
        s0, s1 = losses.sent_loss(code, sent, labels, cids, B)
        gc, gs = torch.autograd.grad(s0 + s1, [code, sent])
        S['sent_loss%s/s0' % cname] = np.float64(s0.item())
        S['sent_loss%s/s1' % cname] = np.float64(s1.item())
        put(S, 'sent_loss%s/gcode' % cname, gc); put(S, 'sent_loss%s/gsent' % cname, gs)
    S['cap_lens'] = x['cap_lens'].numpy()
    S['captions'] = x['captions'].numpy()

    # --- KL (losses.py:210-214)
    mu, lv = fill.unit((B, d['ncf']), 221), 0.3 * fill.unit((B, d['ncf']), 222)
    S['kl'] = np.float64(losses.KL_loss(mu.clone(), lv.clone()).item())

    # --- generator variants (model.py:460-492, model_bert.py G_NET / G_NET_MIX)
    for vname, mod, cls, zkey in (('model', model, 'G_NET', 'z'), ('bert', model_bert, 'G_NET', 'z'),
                                  ('mix', model_bert, 'G_NET_MIX', 'z2')):
        net = getattr(mod, cls)()
        net.train()
        load_filled(net)
        torch.manual_seed(1234)
        eps = torch.FloatTensor(B, d['ncf']).normal_()
        torch.manual_seed(1234)
        imgs, atts, mu, logvar = net(x[zkey], x['sent'], x['words'], x['mask'])
        S['g_%s/eps' % vname] = eps.numpy()
        for i, im in enumerate(imgs):
            put(S, 'g_%s/img%d' % (vname, i), im)
        for i, a in enumerate(atts):
            put(S, 'g_%s/att%d' % (vname, i), a)
        put(S, 'g_%s/mu' % vname, mu); put(S, 'g_%s/logvar' % vname, logvar)
        loss = sum((im * fill.unit(tuple(im.shape), 230 + i)).sum() for i, im in enumerate(imgs)) \
            + losses.KL_loss(mu, logvar)
        names = [n for n, _ in net.named_parameters()]
        grads = torch.autograd.grad(loss, list(net.parameters()), allow_unused=True)
        for n, g in zip(names, grads):
            S['g_%s/gradnorm/%s' % (vname, n)] = np.float64(0.0 if g is None else g.double().norm().item())
        sd = net.state_dict()
        for n in sd:
            if n.endswith('running_mean') or n.endswith('running_var'):
                S['g_%s/buf/%s' % (vname, n)] = np.float64(sd[n].double().sum().item())

    # --- discriminators + discriminator_loss (model.py:611-674, losses.py:136-161)
    fake = [fill.uniform((B, 3, 64 * 2 ** i, 64 * 2 ** i), 300 + i) for i in range(3)]
    for i, cls in enumerate(('D_NET64', 'D_NET128', 'D_NET256')):
        net = getattr(model, cls)()
        net.train()
        load_filled(net, salt=i)
        feat_real = net(x['imgs'][i])
        put(S, 'd%d/feat_real' % i, feat_real)
        put(S, 'd%d/cond_logits' % i, net.COND_DNET(feat_real, x['sent']))
        put(S, 'd%d/uncond_logits' % i, net.UNCOND_DNET(feat_real))
        load_filled(net, salt=i)      # reset BN buffers
        errD = losses.discriminator_loss(net, x['imgs'][i], fake[i], x['sent'],
                                         torch.ones(B), torch.zeros(B))
        S['d%d/errD' % i] = np.float64(errD.item())
        names = [n for n, _ in net.named_parameters()]
        grads = torch.autograd.grad(errD, list(net.parameters()))
        for n, g in zip(names, grads):
            S['d%d/gradnorm/%s' % (i, n)] = np.float64(g.double().norm().item())
        sd = net.state_dict()
        for n in sd:
            if n.endswith('running_mean') or n.endswith('running_var'):
                S['d%d/buf/%s' % (i, n)] = np.float64(sd[n].double().sum().item())
        # gradient w.r.t. the image through D (the G-step path)
        fk = fake[i].clone().requires_grad_(True)
        f = net(fk)
        l = (net.COND_DNET(f, x['sent']) + net.UNCOND_DNET(f)).sum()
        put(S, 'd%d/gimg' % i, torch.autograd.grad(l, fk)[0])

    np.savez_compressed(os.path.join(OUT, 'units_tiny.npz'), **S)
    print('units_tiny.npz: %d arrays' % len(S))


# ---------------------------------------------------------------- full step
def run_reference_steps(ref, d, B, nsteps, variant='model', tag=500, branch=3, slim=False):
    """Drive the reference modules in the order of trainer.py:261-299.  `branch` = TREE.BRANCH_NUM
    (1: the 64 px stage with D_NET64 only, BASELINE config 1); `slim`: losses, gradient norms and
    1024-element slices of the fake images only (the B=20 fixture)."""
    import torch.optim as optim
    cfg, GA, model, model_bert, losses = ref
    set_dims(cfg, dict(d, branch=branch))
    L = 20
    x = make_inputs(d, B, L - 2, branch=branch, lmax=18, tag=tag)
    if variant == 'model':
        netG = model.G_NET()
    elif variant == 'bert':
        netG = model_bert.G_NET()
    else:
        netG = model_bert.G_NET_MIX()
    netsD = [model.D_NET64(), model.D_NET128(), model.D_NET256()][:branch]
    load_filled(netG)
    for i, n in enumerate(netsD):
        load_filled(n, salt=i)
    enc = fill.StandInImageEncoder(d['nef'])
    optG = optim.Adam(netG.parameters(), lr=2e-4, betas=(0.5, 0.999))
    optD = [optim.Adam(n.parameters(), lr=2e-4, betas=(0.5, 0.999)) for n in netsD]
    avg = [p.data.clone() for p in netG.parameters()]
    real_labels, fake_labels, match = torch.ones(B), torch.zeros(B), torch.arange(B)
    S = {'cap_lens': x['cap_lens'].numpy()}
    for step in range(nsteps):
        noise = fill.unit((2, B, d['nz']) if variant == 'mix' else (B, d['nz']), tag + 50 + step)
        torch.manual_seed(777 + step)
        eps = torch.FloatTensor(B, d['ncf']).normal_()
        S['step%d/eps' % step] = eps.numpy()
        torch.manual_seed(777 + step)
        fake, _, mu, logvar = netG(noise, x['sent'], x['words'], x['mask'])
        for i in range(branch):
            netsD[i].zero_grad()
            errD = losses.discriminator_loss(netsD[i], x['imgs'][i], fake[i], x['sent'],
                                             real_labels, fake_labels)
            errD.backward()
            S['step%d/errD%d' % (step, i)] = np.float64(errD.item())
            S['step%d/gnormD%d' % (step, i)] = np.float64(
                torch.sqrt(sum((p.grad.double() ** 2).sum() for p in netsD[i].parameters())).item())
            optD[i].step()
        netG.zero_grad()
        errG, logs = losses.generator_loss(netsD, enc, fake, real_labels, x['words'], x['sent'],
                                           match, x['cap_lens'], x['class_ids'])
        kl = losses.KL_loss(mu, logvar)
        errG = errG + kl
        errG.backward()
        S['step%d/errG_total' % step] = np.float64(errG.item())
        S['step%d/kl_loss' % step] = np.float64(kl.item())
        S['step%d/logs' % step] = np.array(logs)
        S['step%d/gnormG' % step] = np.float64(torch.sqrt(sum(
            (p.grad.double() ** 2).sum() for p in netG.parameters() if p.grad is not None)).item())
        optG.step()
        for p, a in zip(netG.parameters(), avg):
            a.mul_(0.999).add_(p.data, alpha=0.001)
        for i, f in enumerate(fake):
            if slim:
                for k, a in summarize(f, full_limit=0, nsample=1024).items():
                    S['step%d/fake%d/%s' % (step, i, k)] = a
            else:
                put(S, 'step%d/fake%d' % (step, i), f)
        print('  step', step, {k: float(v) for k, v in S.items()
                               if k.startswith('step%d/err' % step)}, flush=True)
    if slim:
        return S
    # post-run state of every parameter and buffer (sum, sumsq, strided sample)
    for n, v in netG.state_dict().items():
        for k, a in summarize(v.float(), full_limit=600, nsample=512).items():
            S['final/G/%s/%s' % (n, k)] = a
    for i, nD in enumerate(netsD):
        for n, v in nD.state_dict().items():
            for k, a in summarize(v.float(), full_limit=600, nsample=512).items():
                S['final/D%d/%s/%s' % (i, n, k)] = a
    S['final/avgG_sum'] = np.float64(sum(a.double().sum().item() for a in avg))
    S['final/avgG_sumsq'] = np.float64(sum((a.double() ** 2).sum().item() for a in avg))
    return S


def gen_step(ref, name, d, B, nsteps, variant, **kw):
    S = run_reference_steps(ref, d, B, nsteps, variant, **kw)
    np.savez_compressed(os.path.join(OUT, name), **S)
    print('%s: %d arrays' % (name, len(S)))


def gen_text(ref):
    """RNN_ENCODER.forward (model.py:127-159) in eval mode on closed-form parameters: two sizes
    (nhidden 128 / ninput 12 and the bird_style.yml size nhidden 256 / ninput 300)."""
    cfg, _, model, _, _ = ref
    S = {}
    for name, ntoken, ninput, nhidden, B, T, lens in (
            ('small', 40, 12, 128, 4, 6, [6, 4, 3, 1]),
            ('bird', 60, 300, 256, 5, 18, [18, 11, 11, 7, 2])):
        cfg.TEXT.WORDS_NUM = T
        net = model.RNN_ENCODER(ntoken, ninput=ninput, nhidden=nhidden)
        P = load_filled(net, salt=7)
        net.eval()
        cap = np.zeros((B, T), dtype=np.int64)
        for b in range(B):
            for t in range(lens[b]):
                cap[b, t] = 1 + (7 * b + 3 * t + b * t) % (ntoken - 1)
        with torch.no_grad():
            words, sent = net(torch.from_numpy(cap), torch.tensor(lens), net.init_hidden(B))
        S['%s/captions' % name] = cap
        S['%s/cap_lens' % name] = np.asarray(lens, dtype=np.int64)
        S['%s/dims' % name] = np.asarray([ntoken, ninput, nhidden], dtype=np.int64)
        S['%s/words_emb' % name] = words.contiguous().numpy()
        S['%s/sent_emb' % name] = sent.numpy()
    np.savez_compressed(os.path.join(OUT, 'text_encoder.npz'), **S)
    print('text_encoder.npz: %d arrays' % len(S))


def gen_text_train(ref):
    """RNN_ENCODER (model.py:75-159) in TRAINING mode (dropout probability 0 so that the run is deterministic):
    outputs and the gradients of every parameter for loss = sum(words * gw) + sum(sent * gs) -- what the DAMSM
    pre-training loop back-propagates through (pretrain_DAMSM.py:79-100)."""
    cfg, _, model, _, _ = ref
    S = {}
    for name, ntoken, ninput, nhidden, B, T, lens in (
            ('small', 40, 12, 128, 4, 6, [6, 4, 3, 1]),
            ('bird', 60, 300, 256, 5, 18, [18, 11, 11, 7, 2])):
        cfg.TEXT.WORDS_NUM = T
        net = model.RNN_ENCODER(ntoken, ninput=ninput, drop_prob=0.0, nhidden=nhidden)
        load_filled(net, salt=7)
        net.train()
        cap = np.zeros((B, T), dtype=np.int64)
        for b in range(B):
            for t in range(lens[b]):
                cap[b, t] = 1 + (7 * b + 3 * t + b * t) % (ntoken - 1)
        words, sent = net(torch.from_numpy(cap), torch.tensor(lens), net.init_hidden(B))
        gw, gs = fill.unit(tuple(words.shape), 801), fill.unit(tuple(sent.shape), 802)
        loss = (words * gw).sum() + (sent * gs).sum()
        names = [n for n, _ in net.named_parameters()]
        grads = torch.autograd.grad(loss, list(net.parameters()))
        S['%s/captions' % name] = cap
        S['%s/cap_lens' % name] = np.asarray(lens, dtype=np.int64)
        S['%s/dims' % name] = np.asarray([ntoken, ninput, nhidden], dtype=np.int64)
        S['%s/words_emb' % name] = words.detach().contiguous().numpy()
        S['%s/sent_emb' % name] = sent.detach().numpy()
        for n, g in zip(names, grads):
            put(S, '%s/grad/%s' % (name, n), g)
    np.savez_compressed(os.path.join(OUT, 'text_encoder_train.npz'), **S)
    print('text_encoder_train.npz: %d arrays' % len(S))


def main():
    os.makedirs(OUT, exist_ok=True)
    what = sys.argv[1:] or ['units', 'step_tiny', 'step_full']
    torch.set_num_threads(8)
    ref = load_reference()
    if 'text' in what:
        gen_text(ref)
    if 'text_train' in what:
        gen_text_train(ref)
    if 'units' in what:
        gen_units(ref)
    if 'step_tiny' in what:
        for v in ('model', 'bert', 'mix'):
            gen_step(ref, 'step_tiny_%s.npz' % v, TINY, 3, 2, v)
    if 'step_full' in what:
        gen_step(ref, 'step_full_model_b4.npz', FULL, 4, 2, 'model')
    if 'step_full_bert' in what:
        gen_step(ref, 'step_full_bert_b4.npz', FULL, 4, 2, 'bert')
    if 'step_full_mix' in what:
        gen_step(ref, 'step_full_mix_b4.npz', FULL, 4, 2, 'mix')
    if 'step_full_branch1' in what:     # BASELINE config 1: bird_style.yml, stage 1 only (64 px), B=4
        gen_step(ref, 'step_full_model_b4_branch1.npz', FULL, 4, 2, 'model', branch=1)
    if 'step_full_b20' in what:         # BASELINE config 2 at its own batch size (losses, grad norms, slices)
        gen_step(ref, 'step_full_model_b20.npz', FULL, 20, 2, 'model', slim=True)
    if 'step_full_bert_b20' in what:    # BASELINE config 3 at the benched batch size
        gen_step(ref, 'step_full_bert_b20.npz', FULL, 20, 2, 'bert', slim=True)
    if 'step_full_mix_b20' in what:     # BASELINE config 5 at the benched batch size
        gen_step(ref, 'step_full_mix_b20.npz', FULL, 20, 2, 'mix', slim=True)


if __name__ == '__main__':
    main()
