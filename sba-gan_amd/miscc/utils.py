"""Training-step helpers with the reference's names (AttnGAN2/code/miscc/utils.py:286-316).
The PIL / skimage attention visualisers of that file (:53-282) are host-side PNG drawing
off the step path and are out of scope."""
import errno
import os
from copy import deepcopy

import torch
import torch.nn as nn


def weights_init(m):
    """utils.py:286-296: orthogonal init for Conv / Linear, BN gamma ~ N(1, 0.02), beta = 0.
    The orthogonal matrix is drawn in a dense temporary because conv weights here are stored
    channels_last (nn.init.orthogonal_ needs a viewable 2-D layout)."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 and hasattr(m, 'weight'):
        tmp = torch.empty(m.weight.shape, dtype=torch.float32)
        nn.init.orthogonal_(tmp, 1.0)
        m.weight.data.copy_(tmp)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif classname.find('Linear') != -1:
        tmp = torch.empty(m.weight.shape, dtype=torch.float32)
        nn.init.orthogonal_(tmp, 1.0)
        m.weight.data.copy_(tmp)
        if m.bias is not None:
            m.bias.data.fill_(0.0)


def load_params(model, new_param):
    """utils.py:299-301."""
    for p, new_p in zip(model.parameters(), new_param):
        p.data.copy_(new_p)
    from sbagan import ops
    ops.weights_changed()


def copy_G_params(model):
    """utils.py:304-306."""
    return deepcopy(list(p.data for p in model.parameters()))


def mkdir_p(path):
    """utils.py:309-316."""
    try:
        os.makedirs(path)
    except OSError as exc:
        if exc.errno == errno.EEXIST and os.path.isdir(path):
            pass
        else:
            raise
