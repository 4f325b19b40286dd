"""Drop-in for the reference's GlobalAttention.py (AttnGAN2/code/GlobalAttention.py):
same public names, HIP kernels underneath (sbagan/csrc/attention.hip, damsm.hip)."""
from sbagan.nets import GlobalAttentionGeneral, conv1x1, func_attention  # noqa: F401
