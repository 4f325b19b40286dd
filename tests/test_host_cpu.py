"""CPU-only tests of the host side: the C-ABI library loads and exports exactly what
include/sbagan_hip.h declares, the config loader mirrors the reference's semantics, the
module tree has the reference's state_dict surface, integer caption work is bit-exact, the
product path refuses to run without a GPU, and the data-parallel exchange is correct on two
gloo ranks."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    txt = open(os.path.join(ROOT, 'include', 'sbagan_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    decls = {}
    for m in re.finditer(r'\bint\s+(sba_\w+)\s*\(([^;]*?)\)\s*;', txt, flags=re.S):
        args = [a for a in m.group(2).split(',') if a.strip() and a.strip() != 'void']
        decls[m.group(1)] = len(args)
    return decls


def test_library_exports_every_declared_symbol():
    from sbagan import _lib
    decls = _header_decls()
    assert len(decls) >= 40
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, nargs in decls.items():
        assert hasattr(_lib.lib, name), name
        assert len(_lib.SIGNATURES[name]) == nargs, (name, nargs, len(_lib.SIGNATURES[name]))
    assert 'gfx950' in _lib.version()


def test_no_cpu_fallback():
    """the product path fails loudly without a device; nothing under sba-gan_amd imports the oracle"""
    from sbagan import ops
    with pytest.raises(RuntimeError):
        ops.LinearFn.apply(torch.zeros(2, 3), torch.zeros(4, 3), None)
    pkg = os.path.join(ROOT, 'sba-gan_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, os.path.join(dp, f)


def test_no_memset_nodes_in_the_library():
    """accumulators are cleared by kernels: hipMemsetAsync captured into a hipGraph is not reliably ordered
    before the following kernel node on ROCm 7.2 (tests/test_step_gpu.py::test_graph_replay_matches_eager_generator)"""
    csrc = os.path.join(ROOT, 'sba-gan_amd', 'csrc')
    for f in os.listdir(csrc):
        if f == 'replay.hip':       # re-issues a captured graph's memset nodes as EAGER (stream-ordered) memsets
            continue
        if f.endswith(('.hip', '.h')):
            code = '\n'.join(l.split('//')[0] for l in open(os.path.join(csrc, f)).read().splitlines())
            assert 'hipMemsetAsync' not in code and 'hipMemset(' not in code, f


def test_config_semantics(tmp_path):
    from miscc.config import cfg, cfg_from_file, reset_cfg
    reset_cfg()
    assert cfg.GAN.GF_DIM == 128 and cfg.TRAIN.SMOOTH.GAMMA1 == 5.0 and cfg.TEXT.WORDS_NUM == 20
    cfg_from_file(os.path.join(ROOT, 'sba-gan_amd', 'cfg', 'bird_style.yml'))
    assert (cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TRAIN.BATCH_SIZE, cfg.TREE.BRANCH_NUM) == (32, 64, 20, 3)
    assert cfg.TRAIN.SMOOTH.LAMBDA == 5.0 and cfg.TRAIN.SMOOTH.GAMMA1 == 4.0 and cfg['GAN']['R_NUM'] == 2
    bad = tmp_path / 'bad.yml'
    bad.write_text('NOT_A_KEY: 1\n')
    with pytest.raises(KeyError):
        cfg_from_file(str(bad))
    bad.write_text('GAN:\n    GF_DIM: "thirty-two"\n')
    with pytest.raises(ValueError):
        cfg_from_file(str(bad))
    for name in ('bird_attn2', 'bird_attnDCGAN2', 'coco_attn2', 'eval_bird', 'eval_bird_attnDCGAN2', 'eval_coco',
                 'DAMSM/bird', 'DAMSM/coco'):
        reset_cfg()
        cfg_from_file(os.path.join(ROOT, 'sba-gan_amd', 'cfg', name + '.yml'))
    reset_cfg()


@pytest.mark.skipif(not os.path.isdir('/root/reference/AttnGAN2/code/cfg'), reason='reference not mounted')
def test_reference_yml_files_load_unchanged():
    from miscc.config import cfg, cfg_from_file, reset_cfg
    base = '/root/reference/AttnGAN2/code/cfg'
    n = 0
    for dp, _, fs in os.walk(base):
        for f in fs:
            if f.endswith('.yml'):
                reset_cfg()
                cfg_from_file(os.path.join(dp, f))
                n += 1
    assert n == 9
    reset_cfg()


def test_shipped_cfg_equals_reference(golden_dir):
    """Every shipped cfg/*.yml parses to exactly the values of the reference file of the same name
    (tests/golden/cfg_values.json = yaml.safe_load of the reference's nine files, written by the same session
    that ran tools/make_cfg.py; compared against /root/reference directly where it is mounted)."""
    import json
    import yaml
    with open(os.path.join(golden_dir, 'cfg_values.json')) as f:
        ref = json.load(f)
    assert len(ref) == 9
    base = os.path.join(ROOT, 'sba-gan_amd', 'cfg')
    shipped = sorted(os.path.relpath(os.path.join(dp, f), base) for dp, _, fs in os.walk(base) for f in fs
                     if f.endswith('.yml'))
    assert shipped == sorted(ref)
    for rel in shipped:
        with open(os.path.join(base, rel)) as f:
            mine = yaml.safe_load(f)
        assert mine == ref[rel], rel
        live = os.path.join('/root/reference/AttnGAN2/code/cfg', rel)
        if os.path.exists(live):
            with open(live) as f:
                assert mine == yaml.safe_load(f), rel


def test_state_dict_surface_matches_reference():
    from helpers import FULL, d_shapes, g_shapes
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    import model
    import model_bert
    for net, exp in ((model.G_NET(), g_shapes(FULL, 3, 'model')), (model_bert.G_NET(), g_shapes(FULL, 3, 'bert')),
                     (model_bert.G_NET_MIX(), g_shapes(FULL, 3, 'bert')), (model.D_NET64(), d_shapes(FULL, 0)),
                     (model.D_NET128(), d_shapes(FULL, 1)), (model.D_NET256(), d_shapes(FULL, 2))):
        sd = net.state_dict()
        assert set(sd) == set(exp)
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(exp[k]), k
    assert len(model.G_NET().state_dict()) == 107 and len(model.D_NET256().state_dict()) == 53
    # weights_init + checkpoint round trip keep the packed (channels_last) storage
    from miscc.utils import copy_G_params, weights_init
    g = model.G_NET()
    g.apply(weights_init)
    w = g.h_net2.upsample[1].weight
    assert w.is_contiguous(memory_format=torch.channels_last)
    flat = w.detach().permute(0, 2, 3, 1).reshape(w.size(0), -1)
    assert torch.allclose(flat @ flat.t(), torch.eye(w.size(0)), atol=1e-4)      # orthogonal rows
    g2 = model.G_NET()
    g2.load_state_dict(g.state_dict())
    assert g2.h_net2.upsample[1].weight.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(g2.h_net2.upsample[1].weight, w)
    assert len(copy_G_params(g)) == 62           # SURVEY: 62 G parameter tensors
    rnn = model.RNN_ENCODER(50, nhidden=256)
    words, sent = rnn(torch.randint(1, 50, (3, 7)), torch.tensor([7, 5, 2]), rnn.init_hidden(3))
    assert words.shape == (3, 256, 7) and sent.shape == (3, 256)


def test_caption_integer_work_bit_exact():
    from oracle import fill
    from oracle import sbagan_oracle as O
    from miscc.losses import class_mask
    from sbagan.trainer import build_mask, sort_by_caption_length
    caps, lens = fill.synthetic_captions(7, 20, 18, tag=3)
    assert torch.equal(build_mask(caps, 18), O.build_mask(caps, 18))
    assert build_mask(caps, 25).shape == (7, 20)
    perm = torch.tensor([3, 1, 6, 0, 2, 5, 4])
    a, b = sort_by_caption_length(lens[perm]), O.sort_by_caption_length(lens[perm])
    assert torch.equal(a[0], b[0])
    ids = np.array([4, 1, 4, 2, 1, 1, 9])
    m = class_mask(ids, 7, 'cpu')
    assert m.dtype == torch.uint8 and np.array_equal(m.numpy().astype(bool), O.class_mask(ids, 7).numpy())
    assert class_mask(None, 7, 'cpu') is None


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sbagan.trainer import FlatParams, GradExchange
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.Conv2d(4, 6, 3))
    net[1].weight.data = net[1].weight.data.contiguous(memory_format=torch.channels_last)
    flat = FlatParams(net)
    ex = GradExchange('cpu')
    assert ex.enabled and ex.world == world
    # each rank's "local gradient": deterministic function of (rank, parameter index)
    for i, p in enumerate(net.parameters()):
        p.grad += (rank + 1) * (i + 1) * torch.ones_like(p)
    h = ex.start(flat.grad)
    ex.wait(h)
    ok = True
    for i, p in enumerate(net.parameters()):
        expect = sum((r + 1) * (i + 1) for r in range(world))
        ok &= bool(torch.allclose(p.grad, torch.full_like(p, float(expect))))
    # mean gradient = what the fused Adam consumes with grad_scale = 1/world
    mean = flat.grad / ex.world
    ok &= bool(torch.allclose(mean[:13 * 7], torch.full((13 * 7,), (world + 1) / 2.0)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 500)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _make_dataset(root, n_train=4, n_test=2, per_image=2):
    """A CUB-shaped toy data_dir: images/<key>.jpg, text/<key>.txt, {train,test}/filenames.pickle, class_info."""
    import pickle
    from PIL import Image
    rng = np.random.RandomState(0)
    words = ['bird', 'red', 'small', 'wing', 'blue', 'beak', 'long', 'white', 'yellow', 'belly', 'tail', 'black']
    os.makedirs(os.path.join(root, 'images', 'cls'), exist_ok=True)
    os.makedirs(os.path.join(root, 'text', 'cls'), exist_ok=True)
    names = {'train': ['cls/a%d' % i for i in range(n_train)], 'test': ['cls/t%d' % i for i in range(n_test)]}
    for split, keys in names.items():
        os.makedirs(os.path.join(root, split), exist_ok=True)
        with open(os.path.join(root, split, 'filenames.pickle'), 'wb') as f:
            pickle.dump(keys, f)
        with open(os.path.join(root, split, 'class_info.pickle'), 'wb') as f:
            pickle.dump(list(range(1, len(keys) + 1)), f)
        for k in keys:
            Image.fromarray(rng.randint(0, 255, (90, 110, 3)).astype(np.uint8)).save(os.path.join(root, 'images', k + '.jpg'))
            with open(os.path.join(root, 'text', k + '.txt'), 'w') as f:
                for c in range(per_image):
                    n = 3 + rng.randint(0, 6)
                    f.write('The ' + ' '.join(words[rng.randint(0, len(words))] for _ in range(n)) + ', it is!\n')
    return names


def test_data_path_matches_reference_contract(tmp_path):
    """datasets.py:28-56,93-318: dictionary building, caption padding, per-scale images in [-1, 1], and
    prepare_data's descending length sort (bit-exact integer work)."""
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    cfg.TREE.BRANCH_NUM, cfg.TEXT.CAPTIONS_PER_IMAGE, cfg.TEXT.WORDS_NUM, cfg.CUDA = 2, 2, 6, False
    import datasets
    from miscc import transforms
    root = str(tmp_path / 'toy')
    names = _make_dataset(root)
    assert datasets.tokenize('The Bird, it is! café') == ['the', 'bird', 'it', 'is', 'caf']
    tf = transforms.Compose([transforms.Resize(int(128 * 76 / 64)), transforms.RandomCrop(128),
                             transforms.RandomHorizontalFlip()])
    ds = datasets.TextDataset(root, 'train', base_size=64, transform=tf)
    assert os.path.isfile(os.path.join(root, 'captions.pickle'))       # built on first use, like the reference
    assert ds.filenames == names['train'] and len(ds) == 4 and ds.imsize == [64, 128]
    assert ds.ixtoword[0] == '<end>' and ds.wordtoix['<end>'] == 0 and ds.n_words == len(ds.ixtoword)
    assert len(ds.captions) == 4 * 2 and all(0 not in c for c in ds.captions)
    imgs, caps, cap_len, cls_id, key = ds[1]
    assert [tuple(i.shape) for i in imgs] == [(3, 64, 64), (3, 128, 128)]
    assert float(imgs[1].min()) >= -1.0 and float(imgs[1].max()) <= 1.0
    assert caps.shape == (6, 1) and caps.dtype == np.int64 and 1 <= cap_len <= 6
    assert (caps[:cap_len, 0] > 0).all() and (caps[cap_len:, 0] == 0).all()
    ds2 = datasets.TextDataset(root, 'test', base_size=64, transform=tf)      # second use loads the pickle
    assert ds2.filenames == names['test'] and ds2.wordtoix == ds.wordtoix
    loader = torch.utils.data.DataLoader(ds, batch_size=4, drop_last=True, shuffle=False)
    batch = next(iter(loader))
    lens_in = batch[2].clone()
    real, captions, lens, class_ids, keys = datasets.prepare_data(batch)
    assert torch.equal(lens, torch.sort(lens_in, 0, True)[0]) and captions.shape == (4, 6)
    assert len(keys) == 4 and real[1].shape == (4, 3, 128, 128) and class_ids.shape == (4,)
    for b in range(4):      # the rows travelled together
        assert int((captions[b] != 0).sum()) == int(lens[b])
    # example sentences -> data_dic (main.py:34-83)
    import main
    with open(os.path.join(root, 'example_filenames.txt'), 'w') as f:
        f.write('example_captions\n')
    with open(os.path.join(root, 'example_captions.txt'), 'w') as f:
        f.write('the small red bird\nblue wing\n\na long white yellow belly tail\n')
    dic = main.build_example_dic(ds.wordtoix, root)
    arr, cl, order = dic['example_captions']
    assert list(cl) == sorted(cl, reverse=True) and arr.shape == (3, max(cl)) and list(order) == [2, 0, 1]
    reset_cfg()


def test_generator_loss_logs_format_like_the_reference():
    """losses.py:184,205 build 'g_loss0: 1.23 g_loss1: ... w_loss: 4.56 s_loss: 7.89 ' with .item() syncs inside
    the step; miscc.losses.LossLogs keeps device scalars and renders the same text on demand."""
    from miscc.losses import LossLogs
    logs = LossLogs()
    logs['g_loss0'], logs['g_loss1'] = torch.tensor(1.234), torch.tensor(0.5)
    logs['w_loss'], logs['s_loss'] = torch.tensor(12.345), torch.tensor(0.004)
    want = 'g_loss0: 1.23 g_loss1: 0.50 w_loss: 12.35 s_loss: 0.00 '
    assert str(logs) == want and '%s' % logs == want
    assert 'errD0: 0.70 ' + '\n' + logs == 'errD0: 0.70 \n' + want       # trainer.py:313 concatenates D_logs + G_logs
    assert isinstance(logs, dict) and float(logs['w_loss']) == pytest.approx(12.345)
