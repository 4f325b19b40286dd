"""Drop-in for the reference's model.py (AttnGAN2/code/model.py): the same class names,
constructor / forward signatures and state_dict keys; forwards run hand-written HIP
kernels for gfx950 (sba-gan_amd/csrc) instead of torch.nn / cuDNN.

    from miscc.config import cfg, cfg_from_file   # load the yml BEFORE building modules
    from model import G_NET, D_NET64, D_NET128, D_NET256, RNN_ENCODER, CNN_ENCODER
"""
from sbagan.nets import (ADAIN_NORM, CA_NET, D_GET_LOGITS, D_NET64, D_NET128, D_NET256,  # noqa: F401
                         G_NET, GET_IMAGE_G, GLU, INIT_STAGE_G, MAPPING_NET, NEXT_STAGE_G,
                         RNN_ENCODER, Block3x3_leakRelu, ResBlock, conv1x1, conv3x3, downBlock,
                         encode_image_by_16times, upBlock)
from sbagan.encoders import CNN_ENCODER  # noqa: F401
