"""What the two entry points (main.py, pretrain_DAMSM.py) share, written from their command-line contract:

    --cfg FILE  --gpu ID  --data_dir DIR  --manualSeed N

the yml file is merged into miscc.config.cfg, --gpu / --data_dir override it, the seed is 100 outside training (the
reference's evaluation runs are seeded that way), the given one or a random one in training, and every run gets an
output directory ../output/<DATASET>_<CONFIG>_<timestamp>."""
import argparse
import datetime
import pprint
import random
import re

import numpy as np
import torch

from .config import cfg, cfg_from_file

EVAL_SEED = 100


def options(what, default_cfg, argv=None):
    ap = argparse.ArgumentParser(description=what)
    ap.add_argument('--cfg', dest='cfg_file', type=str, default=default_cfg, help='optional config file')
    ap.add_argument('--gpu', dest='gpu_id', type=int, default=0)
    ap.add_argument('--data_dir', dest='data_dir', type=str, default='')
    ap.add_argument('--manualSeed', type=int, help='manual seed')
    return ap.parse_args(argv)


def configure(args):
    """merge the yml file and the command-line overrides into cfg, seed every generator; returns the seed"""
    if args.cfg_file:
        cfg_from_file(args.cfg_file)
    if args.gpu_id < 0:
        raise RuntimeError('--gpu -1 (CPU): the HIP modules have no CPU path; the CPU restatement of the step is the '
                           'test oracle (oracle/), not a product path')
    cfg.GPU_ID = args.gpu_id
    if args.data_dir:
        cfg.DATA_DIR = args.data_dir
    print('Using config:')
    pprint.pprint(cfg)
    seed = args.manualSeed
    if not cfg.TRAIN.FLAG:
        seed = EVAL_SEED
    elif seed is None:
        seed = random.randint(1, 10000)
    args.manualSeed = seed
    for seeder in (random.seed, np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all):
        seeder(seed)
    return seed


def output_dir():
    stamp = datetime.datetime.now().strftime('%Y_%m_%d_%H_%M_%S')
    return '../output/%s_%s_%s' % (cfg.DATASET_NAME, cfg.CONFIG_NAME, stamp)


def image_size():
    """side of the largest generated image: BASE_SIZE doubled per extra stage"""
    return cfg.TREE.BASE_SIZE << (cfg.TREE.BRANCH_NUM - 1)


def epoch_of(checkpoint_path):
    """the epoch number at the end of a checkpoint's file name (text_encoder200.pth, netG_epoch_600.pth), or None"""
    m = re.search(r'(\d+)\.[^./\\]+$', checkpoint_path)
    return int(m.group(1)) if m else None
