"""condGANTrainer with the reference's surface (AttnGAN2/code/trainer.py:28-518): build_models,
define_optimizers, prepare_labels, save_model, train, sampling, gen_example -- driving the HIP modules.

What differs, none of it in results:
  * the optimizers are the fused Adam(+EMA) launches over flat parameter buffers (sbagan.trainer.GANStep):
    `define_optimizers` returns them in the reference's (optimizerG, [optimizerD...]) shape;
  * one training step is GANStep.step (same order as trainer.py:245-299), the per-100-iteration log lines are
    formatted from device scalars outside the step (the reference's five .item() syncs per step are gone);
  * `save_img_results` writes a plain image grid per scale: the attention-overlay visualiser
    (miscc/utils.py:53-282, PIL text rendering) is host-side drawing and out of scope;
  * checkpoints are the reference's files: Model/netG_epoch_N.pth (EMA weights, trainer.py:159-164) and
    Model/netD{i}.pth, loadable by either implementation.  With cfg.TRAIN.NET_E == '' the reference prints an
    error and fails; pass `allow_random_encoders=True` to train against randomly initialised frozen encoders
    (synthetic-data runs and tests).
"""
import os
import time

import numpy as np
import torch
from PIL import Image

from datasets import prepare_data
from miscc.config import cfg
from miscc.utils import copy_G_params, load_params, mkdir_p, weights_init
from model import CNN_ENCODER, D_NET64, D_NET128, D_NET256, G_NET, RNN_ENCODER
from sbagan.trainer import GANStep, build_mask, prepare_labels


def _to_uint8(img):
    """[-1, 1] CHW float -> HWC uint8 (trainer.py:419-422)."""
    im = (img.detach().float().cpu().numpy() + 1.0) * 127.5
    return np.transpose(im.clip(0, 255).astype(np.uint8), (1, 2, 0))


class condGANTrainer(object):
    def __init__(self, output_dir, data_loader, n_words, ixtoword, allow_random_encoders=False):
        if cfg.TRAIN.FLAG:
            self.model_dir = os.path.join(output_dir, 'Model')
            self.image_dir = os.path.join(output_dir, 'Image')
            mkdir_p(self.model_dir)
            mkdir_p(self.image_dir)
        torch.cuda.set_device(cfg.GPU_ID)
        self.device = torch.device('cuda', cfg.GPU_ID)
        self.batch_size = cfg.TRAIN.BATCH_SIZE
        self.max_epoch = cfg.TRAIN.MAX_EPOCH
        self.snapshot_interval = cfg.TRAIN.SNAPSHOT_INTERVAL
        self.n_words = n_words
        self.ixtoword = ixtoword
        self.data_loader = data_loader
        self.num_batches = len(self.data_loader)
        self.allow_random_encoders = allow_random_encoders

    # ------------------------------------------------------------------ models (trainer.py:47-130)
    def _generator(self):
        if cfg.GAN.B_DCGAN:
            raise NotImplementedError('G_DCGAN is dead code in the reference (SURVEY.md 2): not built')
        return G_NET()

    def build_models(self):
        dev = self.device
        image_encoder = CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM)
        text_encoder = RNN_ENCODER(self.n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
        if cfg.TRAIN.NET_E == '':
            print('Error: no pretrained text-image encoders')
            if not self.allow_random_encoders:
                return
        else:
            img_encoder_path = cfg.TRAIN.NET_E.replace('text_encoder', 'image_encoder')
            image_encoder.load_state_dict(torch.load(img_encoder_path, map_location='cpu'))
            print('Load image encoder from:', img_encoder_path)
            text_encoder.load_state_dict(torch.load(cfg.TRAIN.NET_E, map_location='cpu'))
            print('Load text encoder from:', cfg.TRAIN.NET_E)
        for p in image_encoder.parameters():
            p.requires_grad = False
        image_encoder.eval()
        for p in text_encoder.parameters():
            p.requires_grad = False
        text_encoder.eval()
        netG = self._generator()
        netsD = [D() for D in (D_NET64, D_NET128, D_NET256)[:cfg.TREE.BRANCH_NUM]]
        netG.apply(weights_init)
        for d in netsD:
            d.apply(weights_init)
        print('# of netsD', len(netsD))
        epoch = 0
        if cfg.TRAIN.NET_G != '':
            netG.load_state_dict(torch.load(cfg.TRAIN.NET_G, map_location='cpu'))
            print('Load G from: ', cfg.TRAIN.NET_G)
            istart = cfg.TRAIN.NET_G.rfind('_') + 1
            iend = cfg.TRAIN.NET_G.rfind('.')
            epoch = int(cfg.TRAIN.NET_G[istart:iend]) + 1
            if cfg.TRAIN.B_NET_D:
                Gname = cfg.TRAIN.NET_G
                for i in range(len(netsD)):
                    Dname = '%s/netD%d.pth' % (Gname[:Gname.rfind('/')], i)
                    print('Load D from: ', Dname)
                    netsD[i].load_state_dict(torch.load(Dname, map_location='cpu'))
        text_encoder = text_encoder.to(dev)
        image_encoder = image_encoder.to(dev)
        netG.to(dev)
        for d in netsD:
            d.to(dev)
        return [text_encoder, image_encoder, netG, netsD, epoch]

    def define_optimizers(self, netG, netsD, image_encoder=None):
        """trainer.py:132-145.  Returns (optimizerG, optimizersD): the fused Adam objects of the GANStep that owns
        the flat parameter / gradient / moment buffers (kept as self.gan)."""
        enc = image_encoder
        if enc is not None and enc.__class__.__name__ == 'CNN_ENCODER':
            from sbagan.inception_hip import InceptionHIP
            enc = InceptionHIP(enc)
        self.gan = GANStep(netG, netsD, enc, self.batch_size)
        return self.gan.optG, self.gan.optD

    def prepare_labels(self):
        return prepare_labels(self.batch_size, self.device)

    def save_model(self, netG, avg_param_G, netsD, epoch):
        """trainer.py:159-170: netG_epoch_N.pth holds the EMA weights."""
        backup_para = copy_G_params(netG)
        load_params(netG, avg_param_G)
        torch.save(netG.state_dict(), '%s/netG_epoch_%d.pth' % (self.model_dir, epoch))
        load_params(netG, backup_para)
        for i, netD in enumerate(netsD):
            torch.save(netD.state_dict(), '%s/netD%d.pth' % (self.model_dir, i))
        print('Save G/Ds models.')

    def set_requires_grad_value(self, models_list, brequires):
        for m in models_list:
            for p in m.parameters():
                p.requires_grad = brequires

    def save_img_results(self, netG, noise, sent_emb, words_embs, mask, image_encoder, captions, cap_lens,
                         gen_iterations, name='current'):
        """trainer.py:177-216 without the attention overlays: one grid (up to 8 samples) per scale."""
        was = netG.training
        netG.eval()
        with torch.no_grad():
            fake_imgs, _, _, _ = netG(noise, sent_emb, words_embs, mask)
        netG.train(was)
        for i, f in enumerate(fake_imgs):
            tiles = [_to_uint8(f[j]) for j in range(min(8, f.size(0)))]
            Image.fromarray(np.concatenate(tiles, 1)).save('%s/G_%s_%d_%d.png' % (self.image_dir, name, gen_iterations, i))

    # ------------------------------------------------------------------ training loop (trainer.py:218-346)
    def _encode(self, text_encoder, captions, cap_lens):
        hidden = text_encoder.init_hidden(captions.size(0))
        with torch.no_grad():
            words_embs, sent_emb = text_encoder(captions, cap_lens, hidden, max_len=int(cap_lens.max()))
        return words_embs.detach(), sent_emb.detach()

    def train(self, max_steps=None):
        built = self.build_models()
        if built is None:
            return
        text_encoder, image_encoder, netG, netsD, start_epoch = built
        self.define_optimizers(netG, netsD, image_encoder)
        gan = self.gan
        netG.set_return_attention(False)        # unused in the step (trainer.py:262)
        batch_size, nz = self.batch_size, cfg.GAN.Z_DIM
        noise = torch.empty((batch_size, nz), device=self.device)
        fixed_noise = torch.randn((batch_size, nz), device=self.device)
        gen_iterations = 0
        out = None
        for epoch in range(start_epoch, self.max_epoch):
            start_t = time.time()
            step = 0
            for data in self.data_loader:
                imgs, captions, cap_lens, class_ids, keys = prepare_data(data)
                words_embs, sent_emb = self._encode(text_encoder, captions, cap_lens)
                mask = build_mask(captions, words_embs.size(2))
                noise.normal_(0, 1)
                out = gan.step(imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise)
                step += 1
                gen_iterations += 1
                if gen_iterations % 100 == 0:
                    v = {k: float(t) for k, t in out.items()}
                    nD = len(netsD)
                    print(' '.join('errD%d: %.2f' % (i, v['errD%d' % i]) for i in range(nD)) + '\n' +
                          ' '.join('g_loss%d: %.2f' % (i, v['g_loss%d' % i]) for i in range(nD)) +
                          ' w_loss: %.2f s_loss: %.2f kl_loss: %.2f' % (v['w_loss'], v['s_loss'], v['kl_loss']))
                if gen_iterations % 1000 == 0:
                    gan.finish()        # (data-parallel: a pending generator update is applied before it is read)
                    backup_para = copy_G_params(netG)
                    load_params(netG, gan.flatG.ema_params())
                    self.save_img_results(netG, fixed_noise, sent_emb, words_embs, mask, image_encoder, captions,
                                          cap_lens, epoch, name='average')
                    load_params(netG, backup_para)
                if max_steps is not None and gen_iterations >= max_steps:
                    break
            end_t = time.time()
            if out is not None:
                errD_total = sum(float(out['errD%d' % i]) for i in range(len(netsD)))
                print('[%d/%d][%d]\n                  Loss_D: %.2f Loss_G: %.2f Time: %.2fs'
                      % (epoch, self.max_epoch, self.num_batches, errD_total, float(out['errG_total']),
                         end_t - start_t))
            if epoch % cfg.TRAIN.SNAPSHOT_INTERVAL == 0:
                gan.finish()
                self.save_model(netG, gan.flatG.ema_params(), netsD, epoch)
            if max_steps is not None and gen_iterations >= max_steps:
                break
        gan.finish()
        self.save_model(netG, gan.flatG.ema_params(), netsD, self.max_epoch)

    # ------------------------------------------------------------------ inference (trainer.py:348-518)
    # Written from the OUTPUT-FILE contract of the reference's three inference entry points (what a user of its eval
    # scripts finds on disk), not from their statements:
    #   save_singleimages  <save_dir>/single_samples/<split_dir>/<filename>_<sentenceID>.jpg
    #   sampling           <NET_G without .pth>/<split>/single/<key>_s-1.png          (last stage only; 'test' -> 'valid')
    #   gen_example        <NET_G without .pth>/<key>/0_s_<original caption index>_g<stage>.png
    @staticmethod
    def _write_image(tensor_chw, path, made=None):
        """one CHW image in [-1, 1] to `path`, creating its directory the first time it is seen"""
        parent = os.path.dirname(path)
        if made is None or parent not in made:
            if not os.path.isdir(parent):
                mkdir_p(parent)
            if made is not None:
                made.add(parent)
        Image.fromarray(_to_uint8(tensor_chw)).save(path)

    def save_singleimages(self, images, filenames, save_dir, split_dir, sentenceID=0):
        root = os.path.join(save_dir, 'single_samples', split_dir)
        made = set()
        for img, name in zip(images, filenames):
            self._write_image(img, '%s_%d.jpg' % (os.path.join(root, name), sentenceID), made)

    def _load_inference_models(self):
        dev = self.device
        netG = self._generator()
        netG.apply(weights_init)
        text_encoder = RNN_ENCODER(self.n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
        if cfg.TRAIN.NET_E != '':
            text_encoder.load_state_dict(torch.load(cfg.TRAIN.NET_E, map_location='cpu'))
            print('Load text encoder from:', cfg.TRAIN.NET_E)
        elif not self.allow_random_encoders:
            raise RuntimeError('cfg.TRAIN.NET_E is empty: no text encoder to load')
        netG.load_state_dict(torch.load(cfg.TRAIN.NET_G, map_location='cpu'))
        print('Load G from: ', cfg.TRAIN.NET_G)
        return netG.to(dev).eval(), text_encoder.to(dev).eval()

    @staticmethod
    def _output_root():
        """the directory named after the generator checkpoint (its path without the .pth suffix), or None"""
        ckpt = cfg.TRAIN.NET_G
        if not ckpt:
            print('cfg.TRAIN.NET_G is empty: inference needs a generator checkpoint')
            return None
        cut = ckpt.rfind('.pth')
        return ckpt[:cut] if cut >= 0 else ckpt

    def _generate(self, netG, text_encoder, captions, cap_lens, noise):
        """fake images of every stage for one caption batch (noise is refilled in place)"""
        words_embs, sent_emb = self._encode(text_encoder, captions, cap_lens)
        noise.normal_(0, 1)
        with torch.no_grad():
            fake_imgs, _, _, _ = netG(noise, sent_emb, words_embs, build_mask(captions, words_embs.size(2)))
        return fake_imgs

    def sampling(self, split_dir):
        """trainer.py:363-433: one image (the last stage's) per caption of the split."""
        root = self._output_root()
        if root is None:
            return None
        out_dir = os.path.join(root, 'valid' if split_dir == 'test' else split_dir)
        mkdir_p(out_dir)
        netG, text_encoder = self._load_inference_models()
        noise = torch.empty((self.batch_size, cfg.GAN.Z_DIM), device=self.device)
        made = set()
        for nbatch, data in enumerate(self.data_loader):
            if nbatch % 100 == 0:
                print('step: ', nbatch)
            _, captions, cap_lens, _, keys = prepare_data(data)
            last = self._generate(netG, text_encoder, captions, cap_lens, noise)[-1]
            for img, key in zip(last, keys):
                self._write_image(img, os.path.join(out_dir, 'single', key) + '_s-1.png', made)
        return out_dir

    def gen_example(self, data_dic):
        """trainer.py:435-518: data_dic[key] = [captions (n x Lmax int64, sorted by length), cap_lens, sorted_indices];
        every stage's image per caption, named by the caption's ORIGINAL position (attention overlays: out of scope)."""
        root = self._output_root()
        if root is None:
            return None
        netG, text_encoder = self._load_inference_models()
        for key, (captions, cap_lens, order) in data_dic.items():
            out_dir = os.path.join(root, key)
            print(out_dir)
            mkdir_p(out_dir)
            captions = torch.from_numpy(np.ascontiguousarray(captions)).to(self.device)
            cap_lens = torch.from_numpy(np.ascontiguousarray(cap_lens)).to(self.device)
            noise = torch.empty((captions.shape[0], cfg.GAN.Z_DIM), device=self.device)
            stages = self._generate(netG, text_encoder, captions, cap_lens, noise)
            for stage, batch in enumerate(stages):
                for img, src in zip(batch, order):
                    self._write_image(img, os.path.join(out_dir, '0_s_%d_g%d.png' % (int(src), stage)))
        return root
