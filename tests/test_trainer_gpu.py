"""condGANTrainer end to end on a toy CUB-shaped data_dir (trainer.py:28-518 surface): two training steps, the
reference's checkpoint files, resume, `sampling` and `gen_example` from the saved generator."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_host_cpu import _make_dataset  # noqa: E402


def test_train_checkpoint_resume_and_sampling(tmp_path):
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 2
    cfg.TEXT.CAPTIONS_PER_IMAGE, cfg.TEXT.WORDS_NUM, cfg.TEXT.EMBEDDING_DIM = 2, 8, 256
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.MAX_EPOCH, cfg.TRAIN.SNAPSHOT_INTERVAL = 2, 1, 1
    cfg.TRAIN.NET_E, cfg.TRAIN.NET_G, cfg.TRAIN.FLAG, cfg.CUDA, cfg.GPU_ID = '', '', True, True, 0
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3, s.LAMBDA = 4.0, 5.0, 10.0, 5.0
    import datasets
    import main
    from miscc import transforms
    from sbagan import ops
    from trainer import condGANTrainer
    ops.set_compute_dtype(torch.bfloat16)
    root = str(tmp_path / 'toy')
    _make_dataset(root)
    tf = transforms.Compose([transforms.Resize(int(128 * 76 / 64)), transforms.RandomCrop(128),
                             transforms.RandomHorizontalFlip()])
    ds = datasets.TextDataset(root, 'train', base_size=64, transform=tf)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, drop_last=True, shuffle=True)
    out_dir = str(tmp_path / 'out')
    torch.manual_seed(3)
    algo = condGANTrainer(out_dir, loader, ds.n_words, ds.ixtoword, allow_random_encoders=True)
    algo.train(max_steps=2)
    g_ckpt = os.path.join(out_dir, 'Model', 'netG_epoch_%d.pth' % cfg.TRAIN.MAX_EPOCH)
    assert os.path.isfile(g_ckpt) and os.path.isfile(os.path.join(out_dir, 'Model', 'netD0.pth'))
    assert os.path.isfile(os.path.join(out_dir, 'Model', 'netD1.pth'))
    # the generator checkpoint holds the EMA weights (trainer.py:159-164), under the reference's key names
    sd = torch.load(g_ckpt, map_location='cpu')
    import model
    ref = model.G_NET()
    assert set(sd.keys()) == set(ref.state_dict().keys())
    ema = algo.gan.flatG.ema_params()
    names = [n for n, _ in algo.gan.netG.named_parameters()]
    for n, a in zip(names, ema):
        assert torch.equal(sd[n], a.detach().cpu()), n
    live = dict(algo.gan.netG.named_parameters())
    assert any(not torch.equal(sd[n], live[n].detach().cpu()) for n in names)      # EMA != live weights
    # resume: build_models loads G (epoch parsed from the file name) and the discriminators beside it
    cfg.TRAIN.NET_G = g_ckpt
    algo2 = condGANTrainer(out_dir, loader, ds.n_words, ds.ixtoword, allow_random_encoders=True)
    text_encoder, image_encoder, netG, netsD, epoch = algo2.build_models()
    assert epoch == cfg.TRAIN.MAX_EPOCH + 1 and len(netsD) == 2
    d0 = torch.load(os.path.join(out_dir, 'Model', 'netD0.pth'), map_location='cpu')
    for n, p in netsD[0].state_dict().items():
        assert torch.equal(p.cpu(), d0[n]), n
    for n, p in netG.state_dict().items():
        assert torch.equal(p.cpu(), sd[n]), n
    # sampling over the test split and customised captions, from the saved generator (eval mode)
    cfg.TRAIN.FLAG = False
    ds_t = datasets.TextDataset(root, 'test', base_size=64, transform=tf)
    loader_t = torch.utils.data.DataLoader(ds_t, batch_size=2, drop_last=True, shuffle=False)
    algo3 = condGANTrainer(out_dir, loader_t, ds_t.n_words, ds_t.ixtoword, allow_random_encoders=True)
    save_dir = algo3.sampling('test')
    pngs = sorted(glob.glob(os.path.join(save_dir, 'single', 'cls', '*_s-1.png')))
    assert len(pngs) == 2
    from PIL import Image
    im = np.asarray(Image.open(pngs[0]))
    assert im.shape == (128, 128, 3) and im.std() > 0
    with open(os.path.join(root, 'example_filenames.txt'), 'w') as f:
        f.write('example_captions\n')
    with open(os.path.join(root, 'example_captions.txt'), 'w') as f:
        f.write('the small red bird\nblue wing\na long white yellow belly tail\n')
    base = algo3.gen_example(main.build_example_dic(ds_t.wordtoix, root))
    assert len(glob.glob(os.path.join(base, 'example_captions', '0_s_*_g1.png'))) == 3
    reset_cfg()


def test_pretrain_damsm_entry_point(tmp_path, monkeypatch):
    """pretrain_DAMSM.py main(): two updates on the toy data_dir, validation pass, encoder checkpoints under the
    reference's file names, loadable by build_models."""
    import yaml
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    root = str(tmp_path / 'toy')
    _make_dataset(root, n_train=4, n_test=4)
    yml = tmp_path / 'damsm_toy.yml'
    yml.write_text(yaml.safe_dump({
        'CONFIG_NAME': 'DAMSM', 'DATASET_NAME': 'toy', 'DATA_DIR': root, 'GPU_ID': 0, 'WORKERS': 0,
        'TREE': {'BRANCH_NUM': 1, 'BASE_SIZE': 128},
        'TRAIN': {'FLAG': True, 'NET_E': '', 'BATCH_SIZE': 4, 'MAX_EPOCH': 1, 'SNAPSHOT_INTERVAL': 1,
                  'ENCODER_LR': 0.002, 'RNN_GRAD_CLIP': 0.25,
                  'SMOOTH': {'GAMMA1': 4.0, 'GAMMA2': 5.0, 'GAMMA3': 10.0}},
        'TEXT': {'EMBEDDING_DIM': 256, 'CAPTIONS_PER_IMAGE': 2, 'WORDS_NUM': 8}}))
    monkeypatch.chdir(tmp_path / 'toy')          # the reference writes to ../output/<name>
    import pretrain_DAMSM
    from sbagan import ops
    ops.set_compute_dtype(torch.bfloat16)
    model_dir = pretrain_DAMSM.main(['--cfg', str(yml), '--gpu', '0', '--manualSeed', '7'], max_steps=2)
    te, ie = os.path.join(model_dir, 'text_encoder0.pth'), os.path.join(model_dir, 'image_encoder0.pth')
    assert os.path.isfile(te) and os.path.isfile(ie)
    cfg.TRAIN.NET_E = te
    text_encoder, image_encoder, labels, start_epoch = pretrain_DAMSM.build_models(
        torch.load(te, map_location='cpu')['encoder.weight'].shape[0], 4)
    assert start_epoch == 1 and labels.tolist() == [0, 1, 2, 3]
    reset_cfg()
