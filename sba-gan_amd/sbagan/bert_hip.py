"""BertEncoder.forward (model_bert.py:177-189) of the FROZEN text encoder on the HIP kernels: embedding + LayerNorm,
12 x [QKV GEMM -> per-head softmax attention without mask -> output GEMM -> add+LayerNorm -> FFN GEMM -> GELU -> FFN
GEMM -> add+LayerNorm], pooler, and the encoder's two heads (1x1 conv + tanh on the tokens, Linear + tanh on the pooled
[CLS] state).  The GEMMs are 1x1 convolutions on sba_conv_igemm_bias (M = B*L rows); csrc/bert.hip holds the rest.

`BertHIP(enc)` wraps a sbagan.encoders.BertEncoder whose `.model` is a HuggingFace BertModel (the substitute for the
reference's pytorch_pretrained_bert, SURVEY.md 8c: third-party arithmetic, parity unpinned) and is a drop-in
callable: captions [B][L] int64 -> (words_embs B x nef x L, sent_emb B x nef), f32, no gradients."""
import ctypes

import torch

from . import _lib, ops
from ._lib import ConvGeom, call


def _gemm_geom(M, K, N):
    g = ConvGeom()
    g.N, g.IH, g.IW, g.Cin = 1, M, 1, K
    g.OH, g.OW, g.Cout = M, 1, N
    g.OHs, g.OWs = M, 1
    g.sy = g.sx = g.osy = g.osx = 1
    g.ntaps = 1
    return g


class BertHIP(object):
    def __init__(self, enc, dtype=None):
        self.enc = enc
        self.dtype = dtype or ops.compute_dtype()
        m = enc.model
        cfgb = m.config
        if cfgb.hidden_act != 'gelu' or cfgb.hidden_size % 64 or cfgb.hidden_size // cfgb.num_attention_heads != 64:
            raise RuntimeError('BertHIP supports BERT-base shaped trunks (GELU, heads of 64 channels)')
        self.C, self.heads, self.eps = cfgb.hidden_size, cfgb.num_attention_heads, float(cfgb.layer_norm_eps)
        dev = next(m.parameters()).device
        self.device = dev
        dt = self.dtype
        f = lambda t: t.detach().float().contiguous()
        w = lambda t: t.detach().to(dt).contiguous()
        e = m.embeddings
        self.we, self.pe, self.te = f(e.word_embeddings.weight), f(e.position_embeddings.weight), f(e.token_type_embeddings.weight[0])
        self.eg, self.eb = f(e.LayerNorm.weight), f(e.LayerNorm.bias)
        self.layers = []
        for l in m.encoder.layer:
            a = l.attention
            self.layers.append(dict(
                wqkv=w(torch.cat((a.self.query.weight, a.self.key.weight, a.self.value.weight), 0)),
                bqkv=f(torch.cat((a.self.query.bias, a.self.key.bias, a.self.value.bias), 0)),
                wo=w(a.output.dense.weight), bo=f(a.output.dense.bias),
                g1=f(a.output.LayerNorm.weight), b1=f(a.output.LayerNorm.bias),
                wi=w(l.intermediate.dense.weight), bi=f(l.intermediate.dense.bias),
                wo2=w(l.output.dense.weight), bo2=f(l.output.dense.bias),
                g2=f(l.output.LayerNorm.weight), b2=f(l.output.LayerNorm.bias)))
        self.wp, self.bp = f(m.pooler.dense.weight), f(m.pooler.dense.bias)
        self.wfc, self.bfc = f(enc.fc.weight), f(enc.fc.bias)
        self.nef = enc.conv_text.weight.shape[0]
        self.wct, self.bct = w(enc.conv_text.weight.view(self.nef, self.C)), f(enc.conv_text.bias)
        self._geoms = {}

    def _dt(self):
        return _lib.SBA_BF16 if self.dtype == torch.bfloat16 else _lib.SBA_F32

    def _linear(self, x, wgt, bias, M, K, N):
        g = self._geoms.get((M, K, N))
        if g is None:
            g = self._geoms[(M, K, N)] = _gemm_geom(M, K, N)
        y = torch.empty((M, N), dtype=self.dtype, device=self.device)
        ws = ops.workspace(self.device)
        ops.tune_geom(g, self._dt())
        call('sba_conv_igemm_bias', self._dt(), x.data_ptr(), wgt.data_ptr(), y.data_ptr(), None, None, bias.data_ptr(),
             None, ctypes.byref(g), ws.data_ptr(), ops.WORKSPACE_BYTES, ops._stream())
        return y

    @torch.no_grad()
    def __call__(self, captions):
        ops._need_gpu(captions)
        B, L = captions.shape
        C, M, dt, st = self.C, B * L, self._dt(), ops._stream()
        cap = captions.to(torch.int64).contiguous()
        x = torch.empty((M, C), dtype=self.dtype, device=self.device)
        call('sba_bert_embed_ln', dt, cap.data_ptr(), self.we.data_ptr(), self.pe.data_ptr(), self.te.data_ptr(),
             self.eg.data_ptr(), self.eb.data_ptr(), x.data_ptr(), B, L, C, self.we.shape[0], self.eps, st)
        for p in self.layers:
            qkv = self._linear(x, p['wqkv'], p['bqkv'], M, C, 3 * C)
            ctx = torch.empty((M, C), dtype=self.dtype, device=self.device)
            call('sba_bert_attention', dt, qkv.data_ptr(), ctx.data_ptr(), B, L, C, self.heads, st)
            a = self._linear(ctx, p['wo'], p['bo'], M, C, C)
            x1 = torch.empty_like(x)
            call('sba_bert_add_ln', dt, a.data_ptr(), x.data_ptr(), p['g1'].data_ptr(), p['b1'].data_ptr(), x1.data_ptr(),
                 M, C, self.eps, st)
            h = self._linear(x1, p['wi'], p['bi'], M, C, 4 * C)
            call('sba_bert_gelu', dt, h.data_ptr(), h.numel(), st)
            o = self._linear(h, p['wo2'], p['bo2'], M, 4 * C, C)
            x = torch.empty_like(x1)
            call('sba_bert_add_ln', dt, o.data_ptr(), x1.data_ptr(), p['g2'].data_ptr(), p['b2'].data_ptr(), x.data_ptr(),
                 M, C, self.eps, st)
        # heads: tanh(conv1x1(tokens)) and tanh(fc(tanh(pooler(CLS))))   (model_bert.py:182-187)
        wt = self._linear(x, self.wct, self.bct, M, C, self.nef)
        words = torch.empty((B, self.nef, L), dtype=torch.float32, device=self.device)
        call('sba_bert_tanh_transpose', dt, wt.data_ptr(), words.data_ptr(), B, L, self.nef, st)
        cls = x.view(B, L, C)[:, 0].float().contiguous()
        pooled = torch.tanh(ops.LinearFn.apply(cls, self.wp, self.bp)) if B <= 32 else \
            torch.tanh(torch.nn.functional.linear(cls, self.wp, self.bp))
        sent = torch.tanh(ops.LinearFn.apply(pooled, self.wfc, self.bfc)) if B <= 32 else \
            torch.tanh(torch.nn.functional.linear(pooled, self.wfc, self.bfc))
        return words, sent
