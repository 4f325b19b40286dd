"""DAMSM pre-training with the reference's entry point (AttnGAN2/code/pretrain_DAMSM.py):

    python pretrain_DAMSM.py --cfg cfg/DAMSM/bird.yml --gpu 0 [--data_dir ...] [--manualSeed N]

train / evaluate / build_models keep the reference's signatures; the per-batch work is sbagan.damsm.DAMSMStep
(HIP kernels).  The attention-map PNGs of the reference's logging (build_super_images) are out of scope."""
import os
import sys
import time

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from datasets import TextDataset, prepare_data  # noqa: E402
from miscc import cli, transforms  # noqa: E402
from miscc.config import cfg  # noqa: E402
from miscc.utils import mkdir_p  # noqa: E402
from model import CNN_ENCODER, RNN_ENCODER  # noqa: E402

UPDATE_INTERVAL = 50


def parse_args(argv=None):
    return cli.options('Train a DAMSM network', 'cfg/DAMSM/bird.yml', argv)


def train(dataloader, cnn_model, rnn_model, batch_size, labels, optimizer, epoch, ixtoword, image_dir,
          max_steps=None):
    """pretrain_DAMSM.py:49-130.  `optimizer` is the sbagan.damsm.DAMSMStep that owns the flat parameter buffers
    and the fused Adam (build it once per run, call .set_lr(lr) per epoch like the reference re-creates Adam)."""
    cnn_model.train()
    rnn_model.train()
    s0 = s1 = w0 = w1 = 0.0
    count = (epoch + 1) * len(dataloader)
    start_time = time.time()
    for step, data in enumerate(dataloader, 0):
        imgs, captions, cap_lens, class_ids, keys = prepare_data(data)
        out = optimizer.step(imgs[-1], captions, cap_lens, class_ids)
        w0 += float(out['w_loss0']); w1 += float(out['w_loss1'])
        s0 += float(out['s_loss0']); s1 += float(out['s_loss1'])
        if step % UPDATE_INTERVAL == 0:
            count = epoch * len(dataloader) + step
            elapsed = time.time() - start_time
            print('| epoch {:3d} | {:5d}/{:5d} batches | ms/batch {:5.2f} | s_loss {:5.2f} {:5.2f} | '
                  'w_loss {:5.2f} {:5.2f}'.format(epoch, step, len(dataloader), elapsed * 1000. / UPDATE_INTERVAL,
                                                  s0 / UPDATE_INTERVAL, s1 / UPDATE_INTERVAL, w0 / UPDATE_INTERVAL,
                                                  w1 / UPDATE_INTERVAL))
            s0 = s1 = w0 = w1 = 0.0
            start_time = time.time()
        if max_steps is not None and step + 1 >= max_steps:
            break
    return count


def evaluate(dataloader, cnn_model, rnn_model, batch_size, damsm=None):
    """pretrain_DAMSM.py:133-163: mean sentence / words loss over (at most) 50 validation batches."""
    cnn_model.eval()
    rnn_model.eval()
    s_total = w_total = 0.0
    step = 0
    for step, data in enumerate(dataloader, 0):
        real_imgs, captions, cap_lens, class_ids, keys = prepare_data(data)
        s, w = damsm.evaluate(real_imgs[-1], captions, cap_lens, class_ids)
        s_total += float(s)
        w_total += float(w)
        if step == 50:
            break
    n = max(step, 1)
    return s_total / n, w_total / n


def build_models(n_words, batch_size):
    """pretrain_DAMSM.py:166-193."""
    text_encoder = RNN_ENCODER(n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
    image_encoder = CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM)
    labels = torch.arange(batch_size, dtype=torch.int64)
    start_epoch = 0
    if cfg.TRAIN.NET_E != '':
        text_encoder.load_state_dict(torch.load(cfg.TRAIN.NET_E, map_location='cpu'))
        print('Load ', cfg.TRAIN.NET_E)
        name = cfg.TRAIN.NET_E.replace('text_encoder', 'image_encoder')     # (the pair is saved side by side)
        image_encoder.load_state_dict(torch.load(name, map_location='cpu'))
        print('Load ', name)
        start_epoch = cli.epoch_of(cfg.TRAIN.NET_E) + 1                     # text_encoder<epoch>.pth
        print('start_epoch', start_epoch)
    dev = torch.device('cuda', cfg.GPU_ID)
    return text_encoder.to(dev), image_encoder.to(dev), labels.to(dev), start_epoch


def main(argv=None, max_steps=None):
    args = parse_args(argv)
    cli.configure(args)
    output_dir = cli.output_dir()
    model_dir, image_dir = os.path.join(output_dir, 'Model'), os.path.join(output_dir, 'Image')
    mkdir_p(model_dir)
    mkdir_p(image_dir)
    torch.cuda.set_device(cfg.GPU_ID)
    imsize = cli.image_size()
    batch_size = cfg.TRAIN.BATCH_SIZE
    image_transform = transforms.Compose([transforms.Scale(int(imsize * 76 / 64)), transforms.RandomCrop(imsize),
                                          transforms.RandomHorizontalFlip()])
    dataset = TextDataset(cfg.DATA_DIR, 'train', base_size=cfg.TREE.BASE_SIZE, transform=image_transform)
    print(dataset.n_words, dataset.embeddings_num)
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, drop_last=True, shuffle=True,
                                             num_workers=int(cfg.WORKERS))
    dataset_val = TextDataset(cfg.DATA_DIR, 'test', base_size=cfg.TREE.BASE_SIZE, transform=image_transform)
    dataloader_val = torch.utils.data.DataLoader(dataset_val, batch_size=batch_size, drop_last=True, shuffle=True,
                                                 num_workers=int(cfg.WORKERS))
    text_encoder, image_encoder, labels, start_epoch = build_models(dataset.n_words, batch_size)
    from sbagan.damsm import DAMSMStep
    damsm = DAMSMStep(text_encoder, image_encoder, batch_size, lr=cfg.TRAIN.ENCODER_LR)
    try:
        lr = cfg.TRAIN.ENCODER_LR
        for epoch in range(start_epoch, cfg.TRAIN.MAX_EPOCH):
            damsm.set_lr(lr)
            train(dataloader, image_encoder, text_encoder, batch_size, labels, damsm, epoch, dataset.ixtoword,
                  image_dir, max_steps=max_steps)
            print('-' * 89)
            if len(dataloader_val) > 0:
                s_loss, w_loss = evaluate(dataloader_val, image_encoder, text_encoder, batch_size, damsm)
                print('| end epoch {:3d} | valid loss {:5.2f} {:5.2f} | lr {:.5f}|'.format(epoch, s_loss, w_loss, lr))
            print('-' * 89)
            if lr > cfg.TRAIN.ENCODER_LR / 10.:
                lr *= 0.98
            if epoch % cfg.TRAIN.SNAPSHOT_INTERVAL == 0 or epoch == cfg.TRAIN.MAX_EPOCH:
                torch.save(image_encoder.state_dict(), '%s/image_encoder%d.pth' % (model_dir, epoch))
                torch.save(text_encoder.state_dict(), '%s/text_encoder%d.pth' % (model_dir, epoch))
                print('Save G/Ds models.')
            if max_steps is not None:
                break
    except KeyboardInterrupt:
        print('-' * 89)
        print('Exiting from training early')
    return model_dir


if __name__ == '__main__':
    main()
