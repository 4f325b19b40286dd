"""Shared helpers for the parity tests: rebuild the closed-form inputs the
golden generator used, compare tensors with the summaries stored in the
fixtures."""
import os

import numpy as np
import torch

from oracle import fill

TINY = dict(ngf=8, ndf=8, nef=16, ncf=10, nz=12, nw=16)
FULL = dict(ngf=32, ndf=64, nef=256, ncf=100, nz=100, nw=256)
SMOOTH = dict(GAMMA1=4.0, GAMMA2=5.0, GAMMA3=10.0, LAMBDA=5.0)


def load_golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=True)


def make_inputs(d, B, L, branch=3, lmax=None, tag=100):
    """Must stay identical to tools/make_golden.py:make_inputs."""
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


def g_shapes(d, branch=3, variant='model', rnum=2):
    """state_dict {name: shape} of G_NET (model.py:440-458 / model_bert.py)
    written out from the architecture; checked against the reference's own
    key listing by tests/test_oracle_golden.py (the fixtures hold one scalar
    per reference key)."""
    ngf, nef, ncf, nz, nw = d['ngf'], d['nef'], d['ncf'], d['nz'], d['nw']
    S = {}

    def bn(p, c):
        S[p + '.weight'] = (c,); S[p + '.bias'] = (c,)
        S[p + '.running_mean'] = (c,); S[p + '.running_var'] = (c,)
        S[p + '.num_batches_tracked'] = ()
    S['ca_net.fc.weight'] = (ncf * 4, nef); S['ca_net.fc.bias'] = (ncf * 4,)
    nmap = 6 if variant == 'model' else 8
    for i in range(nmap):
        S['mapping_net.fc.%d.weight' % i] = (nw, nz if i == 0 else nw)
    g16 = ngf * 16
    in_dim = nz + ncf if variant == 'model' else ncf
    S['h_net1.fc.0.weight'] = (g16 * 4 * 4 * 2, in_dim)
    bn('h_net1.fc.1', g16 * 4 * 4 * 2)
    c = g16
    for i in (1, 2, 3, 4):
        S['h_net1.upsample%d.1.weight' % i] = (c, c, 3, 3)      # out*2 == c
        bn('h_net1.upsample%d.2' % i, c)
        c //= 2
    S['img_net1.img.0.weight'] = (3, ngf, 3, 3)
    ad = 'adain' if variant == 'model' else 'adain2'
    for s in range(2, branch + 1):
        p = 'h_net%d' % s
        S[p + '.att.conv_context.weight'] = (ngf, nef, 1, 1)
        S['%s.%s.style.weight' % (p, ad)] = (ngf * 2, nw)
        S['%s.%s.style.bias' % (p, ad)] = (ngf * 2,)
        for r in range(rnum):
            q = '%s.residual.%d.block' % (p, r)
            S[q + '.0.weight'] = (ngf * 4, ngf * 2, 3, 3)
            bn(q + '.1', ngf * 4)
            S[q + '.3.weight'] = (ngf * 2, ngf * 2, 3, 3)
            bn(q + '.4', ngf * 2)
        S[p + '.upsample.1.weight'] = (ngf * 2, ngf * 2, 3, 3)
        bn(p + '.upsample.2', ngf * 2)
        S['img_net%d.img.0.weight' % s] = (3, ngf, 3, 3)
    return S


def d_shapes(d, which):
    """state_dict {name: shape} of D_NET64/128/256 (model.py:611-674)."""
    ndf, nef = d['ndf'], d['nef']
    S = {}

    def bn(p, c):
        S[p + '.weight'] = (c,); S[p + '.bias'] = (c,)
        S[p + '.running_mean'] = (c,); S[p + '.running_var'] = (c,)
        S[p + '.num_batches_tracked'] = ()
    S['img_code_s16.0.weight'] = (ndf, 3, 4, 4)
    S['img_code_s16.2.weight'] = (ndf * 2, ndf, 4, 4); bn('img_code_s16.3', ndf * 2)
    S['img_code_s16.5.weight'] = (ndf * 4, ndf * 2, 4, 4); bn('img_code_s16.6', ndf * 4)
    S['img_code_s16.8.weight'] = (ndf * 8, ndf * 4, 4, 4); bn('img_code_s16.9', ndf * 8)
    if which >= 1:
        S['img_code_s32.0.weight'] = (ndf * 16, ndf * 8, 4, 4); bn('img_code_s32.1', ndf * 16)
    if which == 1:
        S['img_code_s32_1.0.weight'] = (ndf * 8, ndf * 16, 3, 3); bn('img_code_s32_1.1', ndf * 8)
    if which == 2:
        S['img_code_s64.0.weight'] = (ndf * 32, ndf * 16, 4, 4); bn('img_code_s64.1', ndf * 32)
        S['img_code_s64_1.0.weight'] = (ndf * 16, ndf * 32, 3, 3); bn('img_code_s64_1.1', ndf * 16)
        S['img_code_s64_2.0.weight'] = (ndf * 8, ndf * 16, 3, 3); bn('img_code_s64_2.1', ndf * 8)
    S['UNCOND_DNET.outlogits.0.weight'] = (1, ndf * 8, 4, 4)
    S['UNCOND_DNET.outlogits.0.bias'] = (1,)
    S['COND_DNET.jointConv.0.weight'] = (ndf * 8, ndf * 8 + nef, 3, 3)
    bn('COND_DNET.jointConv.1', ndf * 8)
    S['COND_DNET.outlogits.0.weight'] = (1, ndf * 8, 4, 4)
    S['COND_DNET.outlogits.0.bias'] = (1,)
    return S


def check(G, name, t, rtol=1e-4, atol=1e-5, l2tol=None):
    """Compare tensor `t` with the summary stored under `name` in fixture G."""
    t = t.detach().double().flatten().cpu()
    n = int(G[name + '/numel'])
    assert t.numel() == n, (name, t.numel(), n)
    if name + '/full' in G.files:
        ref = torch.from_numpy(G[name + '/full']).double()
        got = t
    else:
        stride = int(G[name + '/stride'])
        ref = torch.from_numpy(G[name + '/sample']).double()
        got = t[::stride][:ref.numel()]
    if l2tol is not None:
        # relative L2 over the stored sample (used where elementwise bounds are
        # not meaningful: low-precision arithmetic, or values downstream of an
        # Adam update, see test_oracle_golden._check_steps)
        rel = float((got - ref).norm() / ref.norm().clamp(min=1e-30))
        assert rel <= l2tol, '%s: rel L2 err %.3e > %.1e' % (name, rel, l2tol)
        return
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert bool((err <= tol).all()), '%s: max err %.3e (ref scale %.3e)' % (
        name, float(err.max()), float(ref.abs().max()))
    # global checksums (looser: they accumulate rounding over all elements)
    s_ref, q_ref = float(G[name + '/sum']), float(G[name + '/sumsq'])
    assert abs(float(t.sum()) - s_ref) <= 10 * atol * n ** 0.5 + 10 * rtol * (q_ref * n) ** 0.5 / n ** 0.5 + 1e-6 * abs(s_ref), name
    assert abs(float((t * t).sum()) - q_ref) <= 10 * rtol * q_ref + atol, name


def rel_l2(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def check_param(G, name, t, lr_steps, frac=None, med=0.1, q90=0.5):
    """Final-parameter check after Adam steps.  Adam's early steps are
    sign-like (p -= lr * g / (|g| + eps)): each element moves ~lr per step
    whatever the gradient's size, and rounding-level differences in small
    gradients (amplified through the updated D nets) change individual updates
    by O(lr).  So the check is statistical, in units of the distance Adam moved
    the parameters (lr_steps = lr * number of steps): median error <= med,
    90th percentile <= q90, and no element further than Adam can move it."""
    t = t.detach().double().flatten().cpu()
    assert t.numel() == int(G[name + '/numel']), name
    if name + '/full' in G.files:
        ref = torch.from_numpy(G[name + '/full']).double(); got = t
    else:
        ref = torch.from_numpy(G[name + '/sample']).double()
        got = t[::int(G[name + '/stride'])][:ref.numel()]
    err = (got - ref).abs()
    m, q = float(err.median()), float(torch.quantile(err, 0.9))
    assert m <= med * lr_steps + 1e-7, '%s: median err %.3e' % (name, m)
    assert q <= q90 * lr_steps + 1e-7, '%s: q90 err %.3e' % (name, q)
    assert float(err.max()) <= 2.2 * lr_steps + 1e-6, '%s: max err %.3e' % (name, float(err.max()))
