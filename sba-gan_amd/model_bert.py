"""Drop-in for the reference's model_bert.py (AttnGAN2/code/model_bert.py): the "+BERT /
+style" variant -- 8-layer mapping net (:334-356), initial stage fed by c_code only
(:377-425), AdaIN attribute `adain2` (:447), and G_NET_MIX (:485-539).  The D nets are
the same classes as model.py (trainer_bert.py:88 imports them from there)."""
from sbagan.nets import (ADAIN_NORM, CA_NET, D_GET_LOGITS, D_NET64, D_NET128, D_NET256,  # noqa: F401
                         G_NET_MIX, GET_IMAGE_G, GLU, RNN_ENCODER, Block3x3_leakRelu, ResBlock,
                         conv1x1, conv3x3, downBlock, encode_image_by_16times, upBlock)
from sbagan.nets import G_NET_BERT as G_NET  # noqa: F401
from sbagan.encoders import BertEncoder, CNN_ENCODER  # noqa: F401
