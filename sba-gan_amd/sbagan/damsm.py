"""One update of the DAMSM pre-training loop (AttnGAN2/code/pretrain_DAMSM.py:49-130): the text encoder and the two
embedding layers of the image encoder are trained with the words / sentence matching losses.

    words_features, sent_code = cnn_model(imgs[-1])          frozen Inception trunk (HIP implicit-GEMM kernels,
                                                             no tape) + emb_features / emb_cnn_code (trainable)
    words_emb, sent_emb = rnn_model(captions, cap_lens, h)   bi-LSTM with dropout, HIP recurrence + BPTT kernels
    loss = w_loss0 + w_loss1 + s_loss0 + s_loss1             fused all-pairs DAMSM kernels, gradients to both sides
    loss.backward(); clip_grad_norm(rnn params, RNN_GRAD_CLIP); Adam(lr, betas (0.5, 0.999)).step()

The two embedding layers are plain GEMMs (1x1 conv over 289 regions, Linear on the pooled code) and run through
the BLAS library; everything else on the path is a hand-written kernel."""
import torch
import torch.nn as nn

from miscc.config import cfg
from miscc.losses import sent_loss, words_loss

from . import ops
from .trainer import FlatParams, FusedAdam


class _Trainable(nn.Module):
    def __init__(self, text_encoder, image_encoder):
        super(_Trainable, self).__init__()
        self.text_encoder = text_encoder
        self.emb_features = image_encoder.emb_features
        self.emb_cnn_code = image_encoder.emb_cnn_code


class DAMSMStep(object):
    def __init__(self, text_encoder, image_encoder, batch_size, lr=None, grad_clip=None):
        from .inception_hip import InceptionHIP
        self.text_encoder, self.image_encoder = text_encoder, image_encoder
        self.batch_size = batch_size
        self.device = next(text_encoder.parameters()).device
        self.trunk = InceptionHIP(image_encoder)
        self.trainable = _Trainable(text_encoder, image_encoder)
        for p in self.trainable.parameters():
            p.requires_grad_(True)
        self.flat = FlatParams(self.trainable)
        self.grad_clip = cfg.TRAIN.RNN_GRAD_CLIP if grad_clip is None else grad_clip
        self.labels = torch.arange(batch_size, dtype=torch.int64, device=self.device)
        self.set_lr(cfg.TRAIN.ENCODER_LR if lr is None else lr)
        self._rnn_params = list(text_encoder.parameters())

    def set_lr(self, lr):
        """pretrain_DAMSM.py:264: a NEW Adam every epoch (moments restart) with the decayed learning rate."""
        self.flat.m.zero_()
        self.flat.v.zero_()
        self.opt = FusedAdam(self.flat, lr, betas=(0.5, 0.999))

    def image_forward(self, img):
        """CNN_ENCODER.forward (model.py:207-267) with gradients only into emb_features / emb_cnn_code."""
        # (training mode = the reference's cnn_model.train(): BatchNorm with batch statistics, running statistics moved)
        f768, pooled = self.trunk.trunk_features(img, train=self.image_encoder.training)    # [B,17,17,768], [B,2048]
        B = f768.shape[0]
        w = self.image_encoder.emb_features.weight.view(-1, 768)           # [nef, 768]
        feat = torch.matmul(f768.view(B, 289, 768), w.t())                  # [B, 289, nef]
        words_features = feat.transpose(1, 2).reshape(B, -1, 17, 17)
        sent_code = torch.nn.functional.linear(pooled, self.image_encoder.emb_cnn_code.weight,
                                               self.image_encoder.emb_cnn_code.bias)
        return words_features, sent_code

    def losses(self, img, captions, cap_lens, class_ids):
        B = self.batch_size
        words_features, sent_code = self.image_forward(img)
        hidden = self.text_encoder.init_hidden(B)
        words_emb, sent_emb = self.text_encoder(captions, cap_lens, hidden)
        w_loss0, w_loss1, attn_maps = words_loss(words_features, words_emb, self.labels, cap_lens, class_ids, B)
        s_loss0, s_loss1 = sent_loss(sent_code, sent_emb, self.labels, class_ids, B)
        return w_loss0, w_loss1, s_loss0, s_loss1

    def step(self, img, captions, cap_lens, class_ids):
        """Returns the four loss terms as device scalars."""
        ops.det_reset()
        self.flat.zero_grad()
        w0, w1, s0, s1 = self.losses(img, captions, cap_lens, class_ids)
        loss = w0 + w1 + s0 + s1
        loss.backward()
        # torch.nn.utils.clip_grad_norm(rnn_model.parameters(), cfg.TRAIN.RNN_GRAD_CLIP)   (pretrain_DAMSM.py:99-100)
        total = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in self._rnn_params if p.grad is not None))
        coef = torch.clamp(self.grad_clip / (total + 1e-6), max=1.0)
        for p in self._rnn_params:
            if p.grad is not None:
                p.grad.mul_(coef)
        self.opt.step()
        return {'w_loss0': w0.detach(), 'w_loss1': w1.detach(), 's_loss0': s0.detach(), 's_loss1': s1.detach(),
                'rnn_grad_norm': total.detach()}

    @torch.no_grad()
    def evaluate(self, img, captions, cap_lens, class_ids):
        """pretrain_DAMSM.py:133-163 (one batch)."""
        was = self.text_encoder.training
        self.text_encoder.eval()
        w0, w1, s0, s1 = self.losses(img, captions, cap_lens, class_ids)
        self.text_encoder.train(was)
        return (s0 + s1).detach(), (w0 + w1).detach()
