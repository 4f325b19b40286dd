"""Global configuration, drop-in for the reference's `miscc.config`
(AttnGAN2/code/miscc/config.py:9-109): a mutable attribute/item-access `cfg`
holding every default key, and `cfg_from_file(path)` that overlays a yml file with
the same strict checks (unknown key -> KeyError, type mismatch -> ValueError).
Model classes read `cfg` at construction time, exactly like the reference.

Differences: no dependency on `easydict`; the yml is read with yaml.safe_load
(the reference's bare yaml.load(f), config.py:107, fails on PyYAML >= 6).
"""
import numpy as np


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted on assignment."""

    def __init__(self, d=None, **kw):
        super(AttrDict, self).__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super(AttrDict, self).__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def _defaults():
    c = AttrDict()
    c.DATASET_NAME = 'birds'
    c.CONFIG_NAME = ''
    c.DATA_DIR = ''
    c.GPU_ID = 0
    c.CUDA = True
    c.WORKERS = 6
    c.RNN_TYPE = 'LSTM'
    c.B_VALIDATION = False
    c.TREE = AttrDict(BRANCH_NUM=3, BASE_SIZE=64)
    c.TRAIN = AttrDict(
        BATCH_SIZE=64, MAX_EPOCH=600, SNAPSHOT_INTERVAL=2000,
        DISCRIMINATOR_LR=2e-4, GENERATOR_LR=2e-4, ENCODER_LR=2e-4, RNN_GRAD_CLIP=0.25,
        FLAG=True, NET_E='', NET_G='', B_NET_D=True,
        SMOOTH=AttrDict(GAMMA1=5.0, GAMMA3=10.0, GAMMA2=5.0, LAMBDA=1.0),
        MIXING=False)
    c.GAN = AttrDict(DF_DIM=64, GF_DIM=128, Z_DIM=100, W_DIM=256, CONDITION_DIM=100, R_NUM=2,
                     B_ATTENTION=True, B_DCGAN=False)
    c.TEXT = AttrDict(CAPTIONS_PER_IMAGE=10, EMBEDDING_DIM=256, WORDS_NUM=20)
    return c


cfg = _defaults()
__C = cfg


def reset_cfg():
    """Restore every key to its default (handy for tests; not in the reference)."""
    d = _defaults()
    for k in list(cfg.keys()):
        del cfg[k]
    for k, v in d.items():
        cfg[k] = v


def _merge_a_into_b(a, b):
    """Overlay dict a onto config b; a may only name keys b already has, with the same type."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old_type = type(b[k])
        if isinstance(v, dict) and isinstance(b[k], AttrDict):
            try:
                _merge_a_into_b(v, b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
            continue
        if old_type is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(b[k]), type(v), k))
        b[k] = v


def cfg_from_file(filename):
    """Load a yml config file and merge it into the defaults."""
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = yaml.safe_load(f) or {}
    _merge_a_into_b(yaml_cfg, cfg)
