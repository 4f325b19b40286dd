"""CPU oracle for the SBA-GAN adversarial training hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch fp32 CPU restatement of
the reference algorithm, written from the math of the reference (file:line
citations are relative to /root/reference/AttnGAN2/code).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product path (sba-gan_amd/) never does.

Parity status: PINNED.  tools/make_golden.py imports the reference's own
modules in the build container, runs them on closed-form inputs and stores the
outputs under tests/golden/; tests/test_oracle_golden.py checks every function
below against those fixtures (<= 1e-5 abs/rel, integer/mask work bit-exact).

Everything is functional: parameters live in a flat dict keyed by the
reference's state_dict names (e.g. 'h_net1.upsample1.1.weight'); BatchNorm
running statistics are updated in place in that dict, exactly as the reference
modules mutate their buffers.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm default, model.py:43
BN_MOMENTUM = 0.1
IN_EPS = 1e-5          # nn.InstanceNorm2d default, model.py:329


# --------------------------------------------------------------------------
# elementary blocks
# --------------------------------------------------------------------------
def glu(x):
    """model.py:19-23 : first half of the channels times sigmoid(second half)."""
    nc = x.size(1)
    assert nc % 2 == 0, 'channels dont divide 2!'
    h = nc // 2
    return x[:, :h] * torch.sigmoid(x[:, h:])


def batch_norm_train(x, P, prefix, train=True):
    """BatchNorm{1,2}d in train mode (model.py:43,62,65,355,543,553).

    Normalises with the biased batch variance, updates running_mean /
    running_var (unbiased) with momentum 0.1 and bumps num_batches_tracked.
    """
    w, b = P[prefix + '.weight'], P[prefix + '.bias']
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if train:
        n = x.numel() // x.size(1)
        mean = x.mean(dims)
        var = ((x - mean.view(shape)) ** 2).mean(dims)
        with torch.no_grad():
            rm, rv = P[prefix + '.running_mean'], P[prefix + '.running_var']
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
            if prefix + '.num_batches_tracked' in P:
                P[prefix + '.num_batches_tracked'] += 1
    else:
        mean, var = P[prefix + '.running_mean'], P[prefix + '.running_var']
    xhat = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS)
    return xhat * w.view(shape) + b.view(shape)


def up_block(x, P, prefix, train=True):
    """model.py:39-45 : nearest x2 -> conv3x3 (no bias) -> BN -> GLU."""
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    x = F.conv2d(x, P[prefix + '.1.weight'], None, 1, 1)
    x = batch_norm_train(x, P, prefix + '.2', train)
    return glu(x)


def res_block(x, P, prefix, train=True):
    """model.py:57-71 : conv-BN-GLU-conv-BN, plus the input."""
    y = F.conv2d(x, P[prefix + '.block.0.weight'], None, 1, 1)
    y = glu(batch_norm_train(y, P, prefix + '.block.1', train))
    y = F.conv2d(y, P[prefix + '.block.3.weight'], None, 1, 1)
    y = batch_norm_train(y, P, prefix + '.block.4', train)
    return y + x


def ca_net(sent_emb, P, eps, prefix='ca_net'):
    """model.py:271-299.  `eps` is the N(0,1) draw the reference makes inside
    reparametrize (:289-293); it is injected so the oracle is deterministic."""
    x = glu(F.linear(sent_emb, P[prefix + '.fc.weight'], P[prefix + '.fc.bias']))
    c = x.size(1) // 2
    mu, logvar = x[:, :c], x[:, c:]
    std = torch.exp(0.5 * logvar)
    return eps * std + mu, mu, logvar


def mapping_net(z, P, prefix='mapping_net'):
    """model.py:301-321 (6 layers) / model_bert.py:334-356 (8 layers):
    chained bias-free Linears with no activation in between."""
    i = 0
    while '%s.fc.%d.weight' % (prefix, i) in P:
        z = F.linear(z, P['%s.fc.%d.weight' % (prefix, i)])
        i += 1
    return z


def adain_norm(h, w_code, P, prefix):
    """model.py:324-339 : InstanceNorm2d (no affine, biased var, eps 1e-5) then
    (gamma + 1) * xhat + beta with (gamma, beta) = Linear(w)."""
    style = F.linear(w_code, P[prefix + '.style.weight'], P[prefix + '.style.bias'])
    gamma, beta = style.chunk(2, 1)
    mean = h.mean((2, 3), keepdim=True)
    var = ((h - mean) ** 2).mean((2, 3), keepdim=True)
    xhat = (h - mean) / torch.sqrt(var + IN_EPS)
    return (gamma[:, :, None, None] + 1.0) * xhat + beta[:, :, None, None]


def word_attention(h, words, w_ctx, mask):
    """GlobalAttention.py:82-121 (GlobalAttentionGeneral.forward).

    h: B x idf x ih x iw, words: B x cdf x L, w_ctx: idf x cdf x 1 x 1,
    mask: B x L bool or None.  Reproduces the reference's mask quirk
    (:105-108): the score matrix is viewed as (B*queryL) x L and masked with
    mask.repeat(queryL, 1), so row r = b*queryL + q takes mask[r % B].
    """
    B, idf, ih, iw = h.shape
    L = words.size(2)
    Q = ih * iw
    src = torch.einsum('ic,bcl->bil', w_ctx.view(idf, -1), words)      # B x idf x L
    tgt = h.reshape(B, idf, Q).transpose(1, 2)                          # B x Q x idf
    s = torch.bmm(tgt, src).reshape(B * Q, L)
    if mask is not None:
        rows = torch.arange(B * Q) % B
        s = s.masked_fill(mask[rows], -float('inf'))
    a = torch.softmax(s, dim=1).view(B, Q, L).transpose(1, 2)           # B x L x Q
    ctx = torch.bmm(src, a)                                             # B x idf x Q
    return ctx.reshape(B, idf, ih, iw), a.reshape(B, L, ih, iw)


def word_attention_mask_rows(B, Q, mask):
    """Integer part of the quirk: which caption's mask each (b, q) row uses."""
    r = np.arange(B * Q).reshape(B, Q)
    return r % B


def init_stage_g(z, c, P, prefix='h_net1', variant='model', train=True):
    """model.py:342-383 (input cat(c, z)); model_bert.py:377-425 (input c)."""
    x = torch.cat((c, z), 1) if variant == 'model' else c
    x = F.linear(x, P[prefix + '.fc.0.weight'])
    x = glu(batch_norm_train(x, P, prefix + '.fc.1', train))
    ngf16 = P[prefix + '.upsample1.1.weight'].size(1)
    x = x.view(-1, ngf16, 4, 4)
    for i in (1, 2, 3, 4):
        x = up_block(x, P, '%s.upsample%d' % (prefix, i), train)
    return x


def next_stage_g(h, w_code, words, mask, P, prefix, variant='model', train=True):
    """model.py:408-423 : attention, AdaIN, concat, R_NUM ResBlocks, upBlock."""
    ctx, att = word_attention(h, words, P[prefix + '.att.conv_context.weight'], mask)
    ad = 'adain' if variant == 'model' else 'adain2'
    h = adain_norm(h, w_code, P, '%s.%s' % (prefix, ad))
    x = torch.cat((h, ctx), 1)
    i = 0
    while '%s.residual.%d.block.0.weight' % (prefix, i) in P:
        x = res_block(x, P, '%s.residual.%d' % (prefix, i), train)
        i += 1
    return up_block(x, P, prefix + '.upsample', train), att


def get_image_g(h, P, prefix):
    """model.py:426-437 : conv3x3 ngf->3, tanh."""
    return torch.tanh(F.conv2d(h, P[prefix + '.img.0.weight'], None, 1, 1))


def g_net(P, z, sent_emb, words, mask, eps, branch_num=3, variant='model', train=True):
    """G_NET.forward model.py:460-492; variant 'bert' = model_bert.py G_NET,
    variant 'mix' = G_NET_MIX (model_bert.py:505-539, z is 2 x B x nz)."""
    c, mu, logvar = ca_net(sent_emb, P, eps)
    v = 'model' if variant == 'model' else 'bert'
    if variant == 'mix':
        w2, w3 = mapping_net(z[0], P), mapping_net(z[1], P)
        z1 = z
    else:
        w2 = w3 = mapping_net(z, P)
        z1 = z
    imgs, atts = [], []
    h = init_stage_g(z1, c, P, 'h_net1', v, train)
    imgs.append(get_image_g(h, P, 'img_net1'))
    if branch_num > 1:
        h, a = next_stage_g(h, w2, words, mask, P, 'h_net2', v, train)
        imgs.append(get_image_g(h, P, 'img_net2'))
        atts.append(a)
    if branch_num > 2:
        h, a = next_stage_g(h, w3, words, mask, P, 'h_net3', v, train)
        imgs.append(get_image_g(h, P, 'img_net3'))
        atts.append(a)
    return imgs, atts, mu, logvar


# --------------------------------------------------------------------------
# discriminators
# --------------------------------------------------------------------------
def _down(x, P, prefix, iw, ib, train):
    x = F.conv2d(x, P['%s.%d.weight' % (prefix, iw)], None, 2, 1)
    x = batch_norm_train(x, P, '%s.%d' % (prefix, ib), train)
    return F.leaky_relu(x, 0.2)


def _block3x3_leak(x, P, prefix, train):
    x = F.conv2d(x, P[prefix + '.0.weight'], None, 1, 1)
    x = batch_norm_train(x, P, prefix + '.1', train)
    return F.leaky_relu(x, 0.2)


def encode_image_by_16times(x, P, prefix='img_code_s16', train=True):
    """model.py:560-578."""
    x = F.leaky_relu(F.conv2d(x, P[prefix + '.0.weight'], None, 2, 1), 0.2)
    x = _down(x, P, prefix, 2, 3, train)
    x = _down(x, P, prefix, 5, 6, train)
    x = _down(x, P, prefix, 8, 9, train)
    return x


def d_net(P, x, train=True):
    """D_NET64/128/256.forward model.py:611-674; the variant is inferred from
    which keys the parameter dict holds."""
    x = encode_image_by_16times(x, P, 'img_code_s16', train)
    if 'img_code_s32.0.weight' in P:
        x = _down(x, P, 'img_code_s32', 0, 1, train)
    if 'img_code_s64.0.weight' in P:
        x = _down(x, P, 'img_code_s64', 0, 1, train)
        x = _block3x3_leak(x, P, 'img_code_s64_1', train)
        x = _block3x3_leak(x, P, 'img_code_s64_2', train)
    elif 'img_code_s32_1.0.weight' in P:
        x = _block3x3_leak(x, P, 'img_code_s32_1', train)
    return x


def d_get_logits(P, prefix, h, c=None, train=True):
    """D_GET_LOGITS.forward model.py:594-607."""
    if c is not None and (prefix + '.jointConv.0.weight') in P:
        cc = c.view(c.size(0), -1, 1, 1).repeat(1, 1, 4, 4)
        h = _block3x3_leak(torch.cat((h, cc), 1), P, prefix + '.jointConv', train)
    o = F.conv2d(h, P[prefix + '.outlogits.0.weight'], P[prefix + '.outlogits.0.bias'], 4)
    return torch.sigmoid(o).view(-1)


def bce(p, t):
    """nn.BCELoss (mean reduction).  torch's definition: the two logs are
    clamped at -100 in the forward, and the backward is
    (p - t) / max(p * (1 - p), 1e-12) / N  -- finite even at p in {0, 1}, which
    saturated discriminators do reach; F.binary_cross_entropy is used so the
    oracle has exactly that behaviour."""
    return F.binary_cross_entropy(p, t)


def discriminator_loss(PD, real, fake, cond, real_labels, fake_labels, train=True):
    """miscc/losses.py:136-161."""
    rf = d_net(PD, real, train)
    ff = d_net(PD, fake.detach(), train)
    cr = bce(d_get_logits(PD, 'COND_DNET', rf, cond, train), real_labels)
    cf = bce(d_get_logits(PD, 'COND_DNET', ff, cond, train), fake_labels)
    B = rf.size(0)
    cw = bce(d_get_logits(PD, 'COND_DNET', rf[:B - 1], cond[1:B], train), fake_labels[1:B])
    if 'UNCOND_DNET.outlogits.0.weight' in PD:
        r = bce(d_get_logits(PD, 'UNCOND_DNET', rf, None, train), real_labels)
        f = bce(d_get_logits(PD, 'UNCOND_DNET', ff, None, train), fake_labels)
        return (r + cr) / 2. + (f + cf + cw) / 3.
    return cr + (cf + cw) / 2.


# --------------------------------------------------------------------------
# DAMSM
# --------------------------------------------------------------------------
def func_attention(query, context, gamma1):
    """GlobalAttention.py:31-69.  query B x ndf x T, context B x ndf x ih x iw."""
    B, T = query.size(0), query.size(2)
    ih, iw = context.size(2), context.size(3)
    S = ih * iw
    ctx = context.reshape(B, -1, S)
    attn = torch.bmm(ctx.transpose(1, 2), query)              # B x S x T
    attn = torch.softmax(attn.reshape(B * S, T), dim=1).view(B, S, T)
    attn = attn.transpose(1, 2).reshape(B * T, S) * gamma1
    attn = torch.softmax(attn, dim=1).view(B, T, S)
    wctx = torch.bmm(ctx, attn.transpose(1, 2))               # B x ndf x T
    return wctx, attn.view(B, T, ih, iw)


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """miscc/losses.py:11-17."""
    w12 = torch.sum(x1 * x2, dim)
    w1 = torch.norm(x1, 2, dim)
    w2 = torch.norm(x2, 2, dim)
    return w12 / (w1 * w2).clamp(min=eps)


def class_mask(class_ids, B):
    """miscc/losses.py:24-32,73-76,116-119 : mask[i, j] = same class, j != i.
    Integer/bool work: bit-exact."""
    if class_ids is None:
        return None
    class_ids = np.asarray(class_ids)
    m = (class_ids[None, :] == class_ids[:, None])
    m[np.arange(B), np.arange(B)] = False
    return torch.from_numpy(m)


def words_similarity(img_features, words_emb, cap_lens, gamma1, gamma2):
    """The B x B matrix of miscc/losses.py:72-115 before gamma3 and masking:
    sim[j, i] = log sum_t exp(gamma2 * cos(word_t of caption i, its attended
    context in image j))."""
    B = img_features.size(0)
    cols = []
    cap_lens = [int(v) for v in cap_lens]
    for i in range(B):
        T = cap_lens[i]
        word = words_emb[i, :, :T].unsqueeze(0).repeat(B, 1, 1)
        wctx, _ = func_attention(word, img_features, gamma1)
        w = word.transpose(1, 2).reshape(B * T, -1)
        c = wctx.transpose(1, 2).reshape(B * T, -1)
        row = cosine_similarity(w, c).view(B, T)
        cols.append(torch.log(torch.exp(row * gamma2).sum(1, keepdim=True)))
    return torch.cat(cols, 1)


def words_loss(img_features, words_emb, labels, cap_lens, class_ids, B,
               gamma1, gamma2, gamma3):
    """miscc/losses.py:62-132."""
    sim = words_similarity(img_features, words_emb, cap_lens, gamma1, gamma2) * gamma3
    m = class_mask(class_ids, B)
    if m is not None:
        sim = sim.masked_fill(m, -float('inf'))
    return F.cross_entropy(sim, labels), F.cross_entropy(sim.t(), labels)


def sent_loss(cnn_code, rnn_code, labels, class_ids, B, gamma3, eps=1e-8):
    """miscc/losses.py:20-59."""
    n0 = cnn_code.norm(2, dim=1, keepdim=True)
    n1 = rnn_code.norm(2, dim=1, keepdim=True)
    s = cnn_code @ rnn_code.t() / (n0 @ n1.t()).clamp(min=eps) * gamma3
    m = class_mask(class_ids, B)
    if m is not None:
        s = s.masked_fill(m, -float('inf'))
    return F.cross_entropy(s, labels), F.cross_entropy(s.t(), labels)


def kl_loss(mu, logvar):
    """miscc/losses.py:210-214."""
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


def generator_loss(PDs, image_encoder, fake_imgs, real_labels, words_embs, sent_emb,
                   match_labels, cap_lens, class_ids, smooth, train=True):
    """miscc/losses.py:164-206.  `smooth` = dict(GAMMA1, GAMMA2, GAMMA3, LAMBDA).
    Returns (errG_total, dict of the logged components)."""
    B = real_labels.size(0)
    total = 0
    logs = {}
    for i, PD in enumerate(PDs):
        feat = d_net(PD, fake_imgs[i], train)
        g = bce(d_get_logits(PD, 'COND_DNET', feat, sent_emb, train), real_labels)
        if 'UNCOND_DNET.outlogits.0.weight' in PD:
            g = bce(d_get_logits(PD, 'UNCOND_DNET', feat, None, train), real_labels) + g
        total = total + g
        logs['g_loss%d' % i] = g
        if i == len(PDs) - 1:
            region, code = image_encoder(fake_imgs[i])
            w0, w1 = words_loss(region, words_embs, match_labels, cap_lens, class_ids, B,
                                smooth['GAMMA1'], smooth['GAMMA2'], smooth['GAMMA3'])
            s0, s1 = sent_loss(code, sent_emb, match_labels, class_ids, B, smooth['GAMMA3'])
            logs['w_loss'] = (w0 + w1) * smooth['LAMBDA']
            logs['s_loss'] = (s0 + s1) * smooth['LAMBDA']
            total = total + logs['w_loss'] + logs['s_loss']
    return total, logs


# --------------------------------------------------------------------------
# step glue (trainer.py:238-299)
# --------------------------------------------------------------------------
def build_mask(captions, num_words):
    """trainer.py:253-256 : mask = (captions == 0)[:, :Lmax].  Bit-exact."""
    mask = (captions == 0)
    if mask.size(1) > num_words:
        mask = mask[:, :num_words]
    return mask


def sort_by_caption_length(cap_lens):
    """datasets.py:32-33 : descending sort, returns (sorted lens, permutation)."""
    return torch.sort(cap_lens, 0, True)


def adam_update(p, g, m, v, step, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    """torch.optim.Adam as configured at trainer.py:136-143 (no weight decay,
    no amsgrad); `step` is the 1-based step count.  In place."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_update(avg, p):
    """trainer.py:298-299."""
    avg.mul_(0.999).add_(p, alpha=0.001)


TRAINABLE_SUFFIXES = ('.weight', '.bias')


def trainable_keys(P):
    return [k for k in P if k.endswith(TRAINABLE_SUFFIXES)]


class OracleState(object):
    """Parameters + Adam moments + EMA shadow for G and the D nets."""

    def __init__(self, PG, PDs):
        self.PG, self.PDs = PG, PDs
        self.step = 0
        self.mG = {k: torch.zeros_like(PG[k]) for k in trainable_keys(PG)}
        self.vG = {k: torch.zeros_like(PG[k]) for k in trainable_keys(PG)}
        self.mD = [{k: torch.zeros_like(P[k]) for k in trainable_keys(P)} for P in PDs]
        self.vD = [{k: torch.zeros_like(P[k]) for k in trainable_keys(P)} for P in PDs]
        self.avgG = {k: PG[k].clone() for k in trainable_keys(PG)}


def _with_grad(P):
    Q = dict(P)
    for k in trainable_keys(P):
        Q[k] = P[k].detach().requires_grad_(True)
    return Q


def train_step(st, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, eps,
               image_encoder, smooth, lr_g=2e-4, lr_d=2e-4, variant='model'):
    """One full G+D step in the reference's order (trainer.py:261-299):
    G forward once; for each D: loss on (real, fake.detach()), backward, Adam;
    then G loss against the UPDATED D nets (+KL), backward, Adam, EMA.
    Returns a dict of scalars and per-network gradient L2 norms."""
    B = sent_emb.size(0)
    nD = len(st.PDs)
    real_labels, fake_labels = torch.ones(B), torch.zeros(B)
    match_labels = torch.arange(B)
    st.step += 1
    out = {}

    PGg = _with_grad(st.PG)
    fake, _, mu, logvar = g_net(PGg, noise, sent_emb, words_embs, mask, eps,
                                branch_num=nD, variant=variant)
    for i in range(nD):
        PDg = _with_grad(st.PDs[i])
        errD = discriminator_loss(PDg, imgs[i], fake[i], sent_emb, real_labels, fake_labels)
        keys = trainable_keys(st.PDs[i])
        grads = torch.autograd.grad(errD, [PDg[k] for k in keys])
        out['errD%d' % i] = float(errD)
        out['gnormD%d' % i] = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
        with torch.no_grad():
            for k, g in zip(keys, grads):
                adam_update(st.PDs[i][k], g, st.mD[i][k], st.vD[i][k], st.step, lr_d)

    # D nets are not differentiated in the G step (their grads are discarded
    # by the next zero_grad at trainer.py:270); BN buffers still update.
    errG, logs = generator_loss(st.PDs, image_encoder, fake, real_labels, words_embs,
                                sent_emb, match_labels, cap_lens, class_ids, smooth)
    kl = kl_loss(mu, logvar)
    errG = errG + kl
    keys = trainable_keys(st.PG)
    grads = torch.autograd.grad(errG, [PGg[k] for k in keys], allow_unused=True)
    grads = [torch.zeros_like(st.PG[k]) if g is None else g for k, g in zip(keys, grads)]
    out['errG_total'] = float(errG)
    out['kl_loss'] = float(kl)
    for k, v in logs.items():
        out[k] = float(v)
    out['gnormG'] = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    with torch.no_grad():
        for k, g in zip(keys, grads):
            adam_update(st.PG[k], g, st.mG[k], st.vG[k], st.step, lr_g)
            ema_update(st.avgG[k], st.PG[k])
    out['fake'] = [f.detach() for f in fake]
    return out
