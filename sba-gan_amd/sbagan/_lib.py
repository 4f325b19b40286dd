"""ctypes binding of libsbagan_hip.so (the C ABI declared in include/sbagan_hip.h).

The library is loaded eagerly and every symbol the header declares is bound with
explicit argtypes; a missing library or symbol raises at import time -- there is
no CPU or PyTorch fallback behind these calls.
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: PyTorch-ROCm brings its own libamdhip64; if this library is loaded before
#                     it, it binds /opt/rocm's copy and its kernels are registered with a HIP runtime that does
#                     not own torch's streams (every launch then fails)
from ctypes import (POINTER, Structure, byref, c_char_p, c_float, c_int, c_int8, c_int32, c_int64,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SBA_LIB_PATH') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libsbagan_hip.so')

SBA_F32, SBA_BF16, SBA_BF16_YH = 0, 1, 2
IGEMM_TILES = 18          # SBA_IGEMM_TILES of include/sbagan_hip.h
ACT_NONE, ACT_GLU, ACT_LRELU, ACT_RELU = 0, 1, 2, 3
MAX_TAPS = 32


class ConvGeom(Structure):
    _fields_ = [('N', c_int32), ('IH', c_int32), ('IW', c_int32), ('Cin', c_int32),
                ('OH', c_int32), ('OW', c_int32), ('Cout', c_int32),
                ('OHs', c_int32), ('OWs', c_int32),
                ('sy', c_int32), ('sx', c_int32),
                ('osy', c_int32), ('osx', c_int32), ('ooy', c_int32), ('oox', c_int32),
                ('ups', c_int32), ('ntaps', c_int32),
                ('ty', c_int8 * MAX_TAPS), ('tx', c_int8 * MAX_TAPS),
                ('x_cstride', c_int32), ('x_coff', c_int32), ('y_cstride', c_int32), ('y_coff', c_int32),
                ('relu', c_int32), ('tile', c_int32), ('ksplit', c_int32), ('first_write', c_int32),
                ('w_layout', c_int32)]


class ConvGroupItem(Structure):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('y', c_void_p), ('addend', c_void_p), ('bias', c_void_p),
                ('relu_mask', c_void_p), ('g', POINTER(ConvGeom))]


GROUP_MAX = 8

if not os.path.exists(LIB_PATH):
    raise ImportError('libsbagan_hip.so not built: run `python __graft_entry__.py` or '
                      '`make -C sba-gan_amd/csrc` (expected %s)' % LIB_PATH)
lib = ctypes.CDLL(LIB_PATH)

P, I, F, L = c_void_p, c_int, c_float, c_int64
G = POINTER(ConvGeom)

# name -> argtypes; mirrors include/sbagan_hip.h one to one
SIGNATURES = {
    'sba_conv_igemm': [I, P, P, P, P, P, G, P, L, P],
    'sba_conv_igemm_bias': [I, P, P, P, P, P, P, P, G, P, L, P],
    'sba_conv_igemm_plan': [I, G, L, POINTER(c_int)],
    'sba_conv_igemm_group': [I, I, POINTER(ConvGroupItem), I, P],
    'sba_conv_igemm_group_splitk': [I, I, POINTER(ConvGroupItem), I, I, P, L, P],
    'sba_conv_wgrad': [I, P, P, P, G, I, P],
    'sba_pack_weight': [I, P, P, I, I, I, I, I, P],
    'sba_pack_frag_multi': [P, I, I, P],
    'sba_pack_weights_multi': [I, P, I, I, P],
    'sba_pool2x2_sum': [I, P, P, I, I, I, I, P],
    'sba_bn_stats': [I, P, P, L, I, I, P],
    'sba_bn_act_fwd': [I, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, F, F, I, P],
    'sba_bn_act_bwd_reduce': [I, P, P, P, P, L, I, I, I, I, I, P],
    'sba_bn_act_fwd_fused': [I, P, P, P, P, P, P, P, P, L, I, I, I, I, I, F, F, P],
    'sba_bn_act_bwd_fused': [I, P, P, P, P, P, P, L, I, I, I, I, I, P],
    'sba_bn_act_bwd_apply': [I, P, P, P, P, P, P, P, L, I, I, I, I, I, P],
    'sba_bn1d_glu_fwd': [I, P, P, P, P, P, P, P, P, P, I, I, F, F, P],
    'sba_bn1d_glu_bwd': [I, P, P, P, P, P, P, P, P, P, I, I, P],
    'sba_linear_fwd': [P, P, P, P, I, I, I, P],
    'sba_linear_bwd': [P, P, P, P, P, P, I, I, I, P],
    'sba_ca_fwd': [P, P, P, P, P, I, I, P],
    'sba_ca_bwd': [P, P, P, P, P, P, I, I, P],
    'sba_ctx_proj_fwd': [P, P, P, I, I, I, I, P],
    'sba_ctx_proj_bwd': [P, P, P, P, P, I, I, I, I, P],
    'sba_ctx_proj_fwd_fp8': [P, P, P, I, I, I, I, P],
    'sba_instnorm_stats': [I, P, P, P, I, I, I, F, P],
    'sba_adain_fwd': [I, P, P, P, P, P, I, I, I, I, I, P],
    'sba_adain_bwd_reduce': [I, P, P, P, P, P, I, I, I, I, I, P],
    'sba_adain_bwd_apply': [I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    'sba_word_attn_fwd': [I, P, P, P, P, P, I, I, I, I, I, I, I, P],
    'sba_word_attn_fwd_fp8': [P, P, P, P, P, I, I, I, I, I, I, I, P],
    'sba_word_attn_bwd': [I, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    'sba_img_head_fwd': [I, P, P, P, I, I, I, I, P],
    'sba_img_head_bwd': [I, P, P, P, P, P, P, I, I, I, I, I, P],
    'sba_d_stem_fwd': [I, P, P, P, I, I, I, P],
    'sba_d_stem_bwd': [I, P, P, P, P, P, P, I, I, I, P],
    'sba_logits_fwd': [I, P, P, P, P, I, I, P],
    'sba_logits_bwd': [I, P, P, P, P, P, P, P, I, I, I, P],
    'sba_cond_cat_fwd': [I, P, P, P, I, I, I, P],
    'sba_cond_cat_bwd': [I, P, P, P, I, I, I, I, P],
    'sba_resize_bilinear': [P, P, I, I, I, I, P],
    'sba_enc_stem_fwd': [I, P, P, P, P, I, I, I, P],
    'sba_enc_stem_resize_fwd': [I, P, P, P, P, I, I, I, I, P],
    'sba_enc_stem_bwd': [I, P, P, P, P, I, I, I, P],
    'sba_maxpool3x3s2_fwd': [I, P, P, I, I, I, I, I, I, I, I, P],
    'sba_maxpool3x3s2_bwd': [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    'sba_maxpool3x3s2_fwd_arg': [I, P, P, P, I, I, I, I, I, I, I, I, P],
    'sba_maxpool3x3s2_bwd_arg': [I, P, P, P, I, I, I, I, I, I, I, I, I, P, P],
    'sba_avgpool3x3': [I, P, P, I, I, I, I, I, I, I, I, I, P],
    'sba_relu_bwd': [I, P, P, P, L, I, I, I, I, I, P],
    'sba_global_avgpool': [I, P, P, I, I, I, I, P],
    'sba_layout_nhwc_nchw': [I, P, P, I, I, I, I, P],
    'sba_bce_multi': [P, P, P, P, I, P, P, P],
    'sba_kl_loss': [P, P, P, P, P, I, P],
    'sba_damsm_words_fwd': [P, P, P, P, P, P, P, I, I, I, I, F, F, P],
    'sba_damsm_words_bwd': [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, P],
    'sba_damsm_prep': [P, P, P, P, L, I, I, I, I, P],
    'sba_damsm_words_fwd_mfma': [P, P, P, P, P, P, P, I, I, I, I, F, F, P],
    'sba_damsm_words_bwd_mfma': [P, P, P, P, P, P, P, P, P, L, P, I, P, I, I, I, I, F, F, P],
    'sba_ce_pair_direct': [P, P, F, F, P, P, I, P],
    'sba_damsm_sent_direct': [P, P, P, F, F, F, P, P, I, I, P],
    'sba_damsm_sent_fwd': [P, P, P, I, I, F, F, P],
    'sba_damsm_sent_bwd': [P, P, P, P, P, I, I, F, F, P],
    'sba_ce_pair': [P, P, F, P, P, P, I, P],
    'sba_combine2': [P, P, P, P, P, I, P],
    'sba_adam_prepare': [P, F, F, F, P],
    'sba_adam_step': [P, P, P, P, P, P, P, L, F, F, F, F, P],
    'sba_cast': [I, P, I, P, L, P],
    'sba_lstm_bidir_fwd': [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    'sba_lstm_recur_train': [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    'sba_lstm_recur_bwd': [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    'sba_bert_embed_ln': [I, P, P, P, P, P, P, P, I, I, I, I, F, P],
    'sba_bert_add_ln': [I, P, P, P, P, P, I, I, F, P],
    'sba_bert_attention': [I, P, P, I, I, I, I, P],
    'sba_bert_gelu': [I, P, L, P],
    'sba_bert_tanh_transpose': [I, P, P, I, I, I, P],
    'sba_set_deterministic': [I, P, L],
    'sba_set_reduce_scratch': [P, L],
    'sba_det_reset': [],
    'sba_get_deterministic': [],      # (returns the flag, not a status)
    'sba_bn_stat_slots': [],          # (returns the compiled replica count)
    'sba_replay_create': [P, I, I, POINTER(c_void_p)],
    'sba_replay_launch': [P, P],
    'sba_replay_info': [P, POINTER(c_int)],
    'sba_replay_prioritize': [P, P, I, I, I, c_float, I],
    'sba_replay_marker': [I, P],
    'sba_replay_set_callback': [P, P, P],
    'sba_replay_destroy': [P],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError if the .so lacks a declared symbol
    _fn.argtypes = _args
    _fn.restype = c_int
lib.sba_version.restype = c_char_p
lib.sba_version.argtypes = []
lib.sba_det_high_water.restype = c_int64
lib.sba_det_high_water.argtypes = []
lib.sba_damsm_prep_bytes.restype = c_int64
lib.sba_damsm_prep_bytes.argtypes = [I, I, I, I]
lib.sba_damsm_bwd_bytes.restype = c_int64
lib.sba_damsm_bwd_bytes.argtypes = [I, I, I, I]

_ERR = {-1: 'SBA_E_ARG (unsupported shape/alignment/enum)', -2: 'SBA_E_LAUNCH (HIP launch failed)',
        -3: 'SBA_E_UNSUPPORTED (graph node kind the replayer cannot re-issue)'}


AUDIT_HOOK = None        # sbagan.stream_audit: called with (name, args) before every launch while an audit records


def call(name, *args):
    """Invoke a C-ABI entry point; non-zero status becomes RuntimeError (the
    reference surfaces errors as Python exceptions)."""
    if AUDIT_HOOK is not None:
        AUDIT_HOOK(name, args)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError('%s failed: %s' % (name, _ERR.get(rc, rc)))


def version():
    return lib.sba_version().decode()
