"""Losses of the adversarial step with the reference's names and positional signatures
(AttnGAN2/code/miscc/losses.py:11-214), computed by fused HIP kernels.

Differences from the reference, none of which change a result:
  * no host synchronisation: cap_lens is read on the device (the reference calls
    .tolist(), losses.py:71), and the per-term log strings (losses.py:184,205, five
    .item() syncs) are replaced by a dict of device scalars;
  * words_loss evaluates all B x B (caption, image) pairs in one launch instead of a Python
    loop over captions, and does not return the per-caption attention maps (they are only
    used by the PNG visualiser); the third return value is an empty list;
  * the same-class mask is built on the host with numpy exactly like the reference
    (losses.py:24-32,73-76) -- it is integer work on `class_ids`, a host array.
"""
import numpy as np
import os

import torch

from miscc.config import cfg
from sbagan import ops


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """losses.py:11-17 (API parity; the training path uses the fused kernels)."""
    w12 = torch.sum(x1 * x2, dim)
    w1 = torch.norm(x1, 2, dim)
    w2 = torch.norm(x2, 2, dim)
    return (w12 / (w1 * w2).clamp(min=eps)).squeeze()


class LossLogs(dict):
    """generator_loss's second return value.  The reference builds a string with five .item() host syncs inside the
    step (losses.py:184,205: 'g_loss0: 1.23 ... w_loss: 4.56 s_loss: 7.89 '); here the values stay device scalars in
    a dict, and the reference's string is produced on demand: str(logs), '%s' % logs, logs + '\n' all give the
    reference's text (and synchronise only then)."""

    def __str__(self):
        out = ''
        i = 0
        while 'g_loss%d' % i in self:
            out += 'g_loss%d: %.2f ' % (i, float(self['g_loss%d' % i]))
            i += 1
        if 'w_loss' in self and 's_loss' in self:
            out += 'w_loss: %.2f s_loss: %.2f ' % (float(self['w_loss']), float(self['s_loss']))
        return out

    def __add__(self, other):
        return str(self) + other

    def __radd__(self, other):
        return other + str(self)


_MASK_CACHE = {}
BATCH_REAL_FAKE = True
FUSED_HEADS = os.environ.get('SBA_FUSED_HEADS', '1') != '0'   # ops.DHeadsFn: one autograd node per discriminator term
DIRECT_DAMSM = os.environ.get('SBA_DIRECT_DAMSM', '1') != '0'  # damsm_image_terms: loss heads + gradients without autograd


def class_mask(class_ids, batch_size, device):
    """masks[i][j] = (class_ids[j] == class_ids[i]) and j != i  (losses.py:24-32).  Bit-exact
    integer work on the host; cached per (ids, device) so a training loop with fixed ids
    uploads it once."""
    if class_ids is None:
        return None
    ids = np.asarray(class_ids).reshape(-1)[:batch_size]
    key = (ids.tobytes(), str(ids.dtype), batch_size, str(device))
    m = _MASK_CACHE.get(key)
    if m is None:
        masks = []
        for i in range(batch_size):
            mask = (ids == ids[i]).astype(np.uint8)
            mask[i] = 0
            masks.append(mask.reshape((1, -1)))
        m = torch.from_numpy(np.concatenate(masks, 0)).to(device)
        if len(_MASK_CACHE) > 64:
            _MASK_CACHE.clear()
        _MASK_CACHE[key] = m
    return m


def sent_loss(cnn_code, rnn_code, labels, class_ids, batch_size, eps=1e-8):
    """losses.py:20-59.  labels must be arange(batch_size) (the only value the reference
    ever passes, trainer.py:150) or None."""
    if labels is None:
        return None, None
    mask = class_mask(class_ids, batch_size, cnn_code.device)
    return ops.SentLossFn.apply(cnn_code, rnn_code, mask, float(cfg.TRAIN.SMOOTH.GAMMA3), float(eps))


def words_loss(img_features, words_emb, labels, cap_lens, class_ids, batch_size):
    """losses.py:62-132."""
    if labels is None:
        return None, None, []
    mask = class_mask(class_ids, batch_size, img_features.device)
    s = cfg.TRAIN.SMOOTH
    l0, l1 = ops.WordsLossFn.apply(img_features, words_emb, cap_lens, mask,
                                   (float(s.GAMMA1), float(s.GAMMA2), float(s.GAMMA3)))
    return l0, l1, []


def discriminator_loss(netD, real_imgs, fake_imgs, conditions, real_labels, fake_labels, real_features=None):
    """losses.py:136-161: two separate trunk passes (real, fake.detach()), five heads,
    errD = (real + cond_real)/2 + (fake + cond_fake + cond_wrong)/3.

    real_features (optional, beyond the reference's signature): netD(real_imgs) evaluated AHEAD of this call (the
    data-parallel step runs it beside the generator's gradient exchange: it does not depend on the generator); the
    fake half then runs as its own pass, in the reference's order (real first), with the same per-pass BatchNorm
    batches and running-statistic updates."""
    feats = None
    if real_features is not None:
        n = real_features.size(0)
        fake_features = netD(fake_imgs.detach())
        if FUSED_HEADS and fake_features.shape == real_features.shape:
            feats = torch.cat((real_features, fake_features), 0)
            if netD.UNCOND_DNET is not None:
                heads = ((0, n, 0, 1., .5, 1), (n, n, 0, 0., 1. / 3, 3), (0, n - 1, 1, 0., 1. / 3, 4),
                         (0, n, None, 1., .5, 0), (n, n, None, 0., 1. / 3, 2))
            else:
                heads = ((0, n, 0, 1., 1., 0), (n, n, 0, 0., .5, 1), (0, n - 1, 1, 0., .5, 2))
            return ops.d_heads(netD, feats, conditions, heads)
    elif BATCH_REAL_FAKE and real_imgs.shape == fake_imgs.shape:
        # one trunk pass over [real | fake] with per-half BatchNorm batches: same results and module
        # state as the reference's two calls, half the launches and weight reads
        n = real_imgs.size(0)
        feats = netD(torch.cat((real_imgs, fake_imgs.detach()), 0), groups=2)
        real_features, fake_features = feats[:n], feats[n:]
    else:
        real_features = netD(real_imgs)
        fake_features = netD(fake_imgs.detach())
    batch_size = real_features.size(0)
    if FUSED_HEADS and feats is not None:
        # the five heads + the BCE sum as one autograd node over the [real | fake] feature map (ops.DHeadsFn):
        # evaluated in the reference's call order, summed in the order of the terms of errD
        n = batch_size
        if netD.UNCOND_DNET is not None:
            heads = ((0, n, 0, 1., .5, 1), (n, n, 0, 0., 1. / 3, 3), (0, n - 1, 1, 0., 1. / 3, 4),
                     (0, n, None, 1., .5, 0), (n, n, None, 0., 1. / 3, 2))
        else:
            heads = ((0, n, 0, 1., 1., 0), (n, n, 0, 0., .5, 1), (0, n - 1, 1, 0., .5, 2))
        return ops.d_heads(netD, feats, conditions, heads)
    cond_real = netD.COND_DNET(real_features, conditions)
    cond_fake = netD.COND_DNET(fake_features, conditions)
    cond_wrong = netD.COND_DNET(real_features[:(batch_size - 1)], conditions[1:batch_size])
    if netD.UNCOND_DNET is not None:
        real = netD.UNCOND_DNET(real_features)
        fake = netD.UNCOND_DNET(fake_features)
        return ops.BCEMultiFn.apply((1., 1., 0., 0., 0.), (.5, .5, 1. / 3, 1. / 3, 1. / 3),
                                    real, cond_real, fake, cond_fake, cond_wrong)
    return ops.BCEMultiFn.apply((1., 0., 0.), (1., .5, .5), cond_real, cond_fake, cond_wrong)


def discriminator_real_term(netD, real_features, conditions):
    """The terms of errD that read the REAL images only (losses.py:141-149: real, cond_real and the wrong-pair term, with
    their weights 1/2, 1/2, 1/3), as one node over netD(real_imgs): they depend on neither the generator nor the fake
    images, so a trainer can run their backward pass ahead of the fake half (GANStep.real_bwd_early).  Beyond the
    reference: the conditional head's BatchNorm then sees its batches in the order real, wrong, fake instead of real,
    fake, wrong (running statistics only; nothing reads them in training)."""
    n = real_features.size(0)
    if netD.UNCOND_DNET is not None:
        heads = ((0, n, 0, 1., .5, 1), (0, n - 1, 1, 0., 1. / 3, 2), (0, n, None, 1., .5, 0))
    else:
        heads = ((0, n, 0, 1., 1., 0), (0, n - 1, 1, 0., .5, 1))
    return ops.d_heads(netD, real_features, conditions, heads)


def discriminator_fake_term(netD, fake_imgs, conditions):
    """the terms of errD that read the fake images (losses.py:145-158: fake, cond_fake), as their own trunk pass"""
    fake_features = netD(fake_imgs.detach())
    n = fake_features.size(0)
    if netD.UNCOND_DNET is not None:
        heads = ((0, n, 0, 0., 1. / 3, 1), (0, n, None, 0., 1. / 3, 0))
    else:
        heads = ((0, n, 0, 0., .5, 0),)
    return ops.d_heads(netD, fake_features, conditions, heads)


def damsm_image_terms(image_encoder, fake_img, words_embs, sent_emb, match_labels, cap_lens, class_ids):
    """The DAMSM ranking terms of generator_loss (losses.py:187-204) for one batch of fake images, together with
    their gradient with respect to the images: returns (w_loss, s_loss, d(w_loss + s_loss)/d fake_img).

    These terms depend on the generator's output only -- not on the discriminators -- so a trainer can evaluate
    them (image encoder forward, words / sentence loss, backward through the frozen encoder) beside the
    discriminator updates and hand the image gradient to the generator's backward pass later
    (generator_loss(..., damsm=...)); by linearity the parameter gradients are those of the reference's single
    backward pass of errG_total."""
    batch_size = fake_img.size(0)
    leaf = fake_img.detach().requires_grad_(True)
    region_features, cnn_code = image_encoder(leaf)
    if DIRECT_DAMSM and match_labels is not None and region_features.is_cuda:
        # the loss heads and their gradients as ten back-to-back launches (ops.damsm_terms_direct), then ONLY the encoder's
        # backward pass through autograd -- bit-identical to the Function path below
        s = cfg.TRAIN.SMOOTH
        mask = class_mask(class_ids, batch_size, region_features.device)
        w_loss, s_loss, dfeat, dcnn = ops.damsm_terms_direct(
            region_features, cnn_code, words_embs, sent_emb, cap_lens, mask,
            (float(s.GAMMA1), float(s.GAMMA2), float(s.GAMMA3)), float(s.LAMBDA))
        (grad,) = torch.autograd.grad([region_features, cnn_code],
                                      leaf, [dfeat.to(region_features.dtype), dcnn.to(cnn_code.dtype)])
        return w_loss, s_loss, grad
    w_loss0, w_loss1, _ = words_loss(region_features, words_embs, match_labels, cap_lens, class_ids, batch_size)
    w_loss = (w_loss0 + w_loss1) * cfg.TRAIN.SMOOTH.LAMBDA
    s_loss0, s_loss1 = sent_loss(cnn_code, sent_emb, match_labels, class_ids, batch_size)
    s_loss = (s_loss0 + s_loss1) * cfg.TRAIN.SMOOTH.LAMBDA
    (grad,) = torch.autograd.grad(w_loss + s_loss, leaf)
    return w_loss.detach(), s_loss.detach(), grad


def _g_term(netD, features, sent_emb):
    """the adversarial term of generator_loss for one discriminator's features of the fake images (losses.py:168-186)"""
    if FUSED_HEADS:
        n = features.size(0)
        if netD.UNCOND_DNET is not None:
            return ops.d_heads(netD, features, sent_emb, ((0, n, 0, 1., 1., 1), (0, n, None, 1., 1., 0)))
        return ops.d_heads(netD, features, sent_emb, ((0, n, 0, 1., 1., 0),))
    cond_logits = netD.COND_DNET(features, sent_emb)
    if netD.UNCOND_DNET is not None:
        logits = netD.UNCOND_DNET(features)
        return ops.BCEMultiFn.apply((1., 1.), (1., 1.), logits, cond_logits)
    return ops.BCEMultiFn.apply((1.,), (1.,), cond_logits)


def generator_d_term(netD, fake_img, sent_emb):
    """One discriminator's term of generator_loss (losses.py:168-186) on its own, together with its gradient with
    respect to the fake images: returns (g_loss, d g_loss / d fake_img).

    The term needs discriminator i AFTER its update and fake image i, nothing else: a trainer can evaluate it -- forward
    through the discriminator, backward to the image -- on the stream of that discriminator's update, right behind its
    optimizer step, while the other discriminators are still updating, and hand the image gradients to the generator's
    single backward pass later (generator_loss(..., d_terms=...), backward_with_image_grads); by linearity the parameter
    gradients are those of the reference's backward pass of errG_total.  The discriminator's parameters must not
    require gradients (the reference computes and discards them, trainer.py:270,287)."""
    leaf = fake_img.detach().requires_grad_(True)
    g_loss = _g_term(netD, netD(leaf), sent_emb)
    (grad,) = torch.autograd.grad(g_loss, leaf)
    return g_loss.detach(), grad


def generator_loss(netsD, image_encoder, fake_imgs, real_labels, words_embs, sent_emb, match_labels,
                   cap_lens, class_ids, streams=None, damsm=None, d_terms=None):
    """losses.py:164-206.  Returns (errG_total, logs) where logs is a dict of device scalars
    {'g_loss0', ..., 'w_loss', 's_loss'} (format with .item() outside the step).

    streams (optional, len(netsD) + 1 HIP streams): the per-discriminator terms and the
    encoder + DAMSM term are independent branches (forward and backward), so each may run on its
    own stream; autograd replays every branch's backward on the stream of its forward.

    damsm (optional, the (w_loss, s_loss, image gradient) of damsm_image_terms): the ranking terms were
    evaluated ahead of time; errG_total then carries their VALUES (same summation order as the reference) and the
    caller back-propagates with `backward_with_image_grad`.

    d_terms (optional, the g_loss values of generator_d_term, one per discriminator): likewise for the adversarial
    terms; the caller back-propagates with `backward_with_image_grads`."""
    import contextlib
    numDs = len(netsD)
    batch_size = real_labels.size(0)
    logs = LossLogs()
    main = torch.cuda.current_stream() if streams else None

    def branch(k):
        if not streams:
            return contextlib.nullcontext()
        streams[k].wait_stream(main)
        return torch.cuda.stream(streams[k])

    terms = []
    for i in range(numDs):
        if d_terms is not None:         # evaluated ahead of time (generator_d_term): the VALUES, same summation order
            g_loss = d_terms[i]
        else:
            with branch(i):
                g_loss = _g_term(netsD[i], netsD[i](fake_imgs[i]), sent_emb)
        terms.append(g_loss)
        logs['g_loss%d' % i] = g_loss.detach()
    if damsm is not None:
        w_loss, s_loss = damsm[0], damsm[1]
    else:
        # ranking loss on the last scale (losses.py:187-204).  An encoder that forks its own streams
        # (sbagan.inception_hip) stays on the calling stream: it already overlaps the D branches, and
        # nested forks inside a captured branch crash hipStreamEndCapture on ROCm 7.2.
        own = streams and not getattr(image_encoder, 'parallel', False)
        with (branch(numDs) if own else contextlib.nullcontext()):
            region_features, cnn_code = image_encoder(fake_imgs[numDs - 1])
            w_loss0, w_loss1, _ = words_loss(region_features, words_embs, match_labels, cap_lens, class_ids,
                                             batch_size)
            w_loss = (w_loss0 + w_loss1) * cfg.TRAIN.SMOOTH.LAMBDA
            s_loss0, s_loss1 = sent_loss(cnn_code, sent_emb, match_labels, class_ids, batch_size)
            s_loss = (s_loss0 + s_loss1) * cfg.TRAIN.SMOOTH.LAMBDA
    if streams:
        for st in streams[:numDs + 1]:
            main.wait_stream(st)
    errG_total = 0
    for i in range(numDs):
        errG_total = errG_total + terms[i]          # same summation order as the reference
    errG_total = errG_total + w_loss + s_loss
    logs['w_loss'] = w_loss.detach()
    logs['s_loss'] = s_loss.detach()
    return errG_total, logs


def backward_with_image_grads(errG_total, fake_imgs, image_grads):
    """One backward pass for errG_total whose adversarial AND ranking terms were evaluated ahead of time
    (generator_loss(..., d_terms=..., damsm=...)): their gradients enter at the generator's images; errG_total itself
    carries the graph of the KL term only."""
    torch.autograd.backward([errG_total] + list(fake_imgs), [None] + list(image_grads))


def backward_with_image_grad(errG_total, fake_img, image_grad):
    """One backward pass for errG_total whose DAMSM terms were evaluated ahead of time: their gradient enters at
    the generator's last image (generator_loss(..., damsm=...))."""
    torch.autograd.backward([errG_total, fake_img], [None, image_grad])


def KL_loss(mu, logvar):
    """losses.py:210-214."""
    return ops.KLFn.apply(mu, logvar)
