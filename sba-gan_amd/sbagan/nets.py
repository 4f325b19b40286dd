"""Module classes of the hot path with the reference's constructor / forward /
state_dict surface (AttnGAN2/code/model.py, model_bert.py, GlobalAttention.py),
implemented over the HIP operators in sbagan.ops.

The nn.Module tree only HOLDS parameters (same attribute names and
nn.Sequential indices as the reference, so `load_state_dict` of reference
checkpoints works); every forward goes through a fused autograd Function.
Conv weights are OIHW tensors stored channels_last ([O][KH][KW][I] in memory),
which is the packed layout the implicit-GEMM kernels read.
"""
import os

import torch
import torch.nn as nn

from miscc.config import cfg

from . import ops
from .ops import ACT_GLU, ACT_LRELU, ACT_NONE, CL


# ----------------------------------------------------------------------------
# parameter holders
# ----------------------------------------------------------------------------
class _Slot(nn.Module):
    """Occupies an nn.Sequential index that holds a parameter-less module in the
    reference (nn.Upsample, GLU, LeakyReLU, Tanh, Sigmoid)."""

    def forward(self, x):
        return x


def _conv(cin, cout, k, stride=1, pad=0, bias=False):
    m = nn.Conv2d(cin, cout, k, stride, pad, bias=bias)
    m.weight.data = m.weight.data.contiguous(memory_format=CL)
    return m


def conv1x1(in_planes, out_planes, bias=False):
    return _conv(in_planes, out_planes, 1, 1, 0, bias)


def conv3x3(in_planes, out_planes):
    return _conv(in_planes, out_planes, 3, 1, 1, False)


class GLU(nn.Module):
    """model.py:15-23 (kept for API parity; the fused blocks apply it in-kernel)."""

    def forward(self, x):
        nc = x.size(1)
        assert nc % 2 == 0, 'channels dont divide 2!'
        nc = nc // 2
        return x[:, :nc] * torch.sigmoid(x[:, nc:])


class _Layer(object):
    """(conv, bn, packed weight) triple handed to the autograd Functions."""

    def __init__(self, conv, bn, kind='3x3'):
        self.conv, self.bn = conv, bn
        self.pw = ops.PackedWeight(conv.weight, kind)


class _ConvBNAct(nn.Sequential):
    """nn.Sequential-shaped holder whose forward is ONE fused conv+BN+activation."""
    kind = '3x3'
    act = ACT_NONE
    conv_idx = 0

    def _layer(self):
        l = self.__dict__.get('_l')
        if l is None:
            l = _Layer(self[self.conv_idx], self[self.conv_idx + 1], self.kind)
            self.__dict__['_l'] = l
        return l

    def forward(self, x, residual=None, groups=1):
        l = self._layer()
        return ops.ConvBNActFn.apply(x, l.conv.weight, l.bn.weight, l.bn.bias, l, self.kind, self.act, residual,
                                     groups)


class _UpBlock(_ConvBNAct):
    kind, act, conv_idx = '3x3up', ACT_GLU, 1


class _Block3x3LeakRelu(_ConvBNAct):
    kind, act, conv_idx = '3x3', ACT_LRELU, 0


class _DownBlock(_ConvBNAct):
    kind, act, conv_idx = '4x4s2', ACT_LRELU, 0


def upBlock(in_planes, out_planes):
    """model.py:39-45: Upsample(x2 nearest), conv3x3, BatchNorm2d, GLU."""
    return _UpBlock(_Slot(), conv3x3(in_planes, out_planes * 2), nn.BatchNorm2d(out_planes * 2), _Slot())


def Block3x3_leakRelu(in_planes, out_planes):
    """model.py:540-546."""
    return _Block3x3LeakRelu(conv3x3(in_planes, out_planes), nn.BatchNorm2d(out_planes), _Slot())


def downBlock(in_planes, out_planes):
    """model.py:550-556."""
    return _DownBlock(_conv(in_planes, out_planes, 4, 2, 1), nn.BatchNorm2d(out_planes), _Slot())


class ResBlock(nn.Module):
    """model.py:57-71."""

    def __init__(self, channel_num):
        super(ResBlock, self).__init__()
        self.block = nn.Sequential(
            conv3x3(channel_num, channel_num * 2), nn.BatchNorm2d(channel_num * 2), _Slot(),
            conv3x3(channel_num, channel_num), nn.BatchNorm2d(channel_num))

    def _layers(self):
        if '_l1' not in self.__dict__:
            self.__dict__['_l1'] = _Layer(self.block[0], self.block[1])
            self.__dict__['_l2'] = _Layer(self.block[3], self.block[4])
        return self.__dict__['_l1'], self.__dict__['_l2']

    @property
    def l1(self):
        return self._layers()[0]

    @property
    def l2(self):
        return self._layers()[1]

    def forward(self, x):
        l1, l2 = self._layers()
        return ops.ResBlockFn.apply(x, l1.conv.weight, l1.bn.weight, l1.bn.bias,
                                    l2.conv.weight, l2.bn.weight, l2.bn.bias, self)


def pack_group(net):
    """One ops.PackGroup over every fused conv layer of `net` (its packed weights are then
    refreshed by a single launch per optimizer step)."""
    layers = []
    for m in net.modules():
        if isinstance(m, _ConvBNAct):
            layers.append((m._layer().pw, m.kind))
        elif isinstance(m, ResBlock):
            l1, l2 = m._layers()
            layers += [(l1.pw, '3x3'), (l2.pw, '3x3')]
        elif isinstance(m, _EncodeBy16):
            layers += [(l.pw, '4x4s2') for l in m._layers()]
    return ops.PackGroup(layers) if layers else None


def _linear(x, lin):
    return ops.LinearFn.apply(x, lin.weight, lin.bias)


# ----------------------------------------------------------------------------
# attention (GlobalAttention.py)
# ----------------------------------------------------------------------------
def func_attention(query, context, gamma1):
    """GlobalAttention.py:31-69 for one (query, context) batch; provided for API parity
    (the training path uses the fused all-pairs kernel behind miscc.losses.words_loss).
    Returns (weightedContext B x ndf x T, attn B x T x ih x iw)."""
    B, T = query.size(0), query.size(2)
    ih, iw = context.size(2), context.size(3)
    S = ih * iw
    ctx = context.reshape(B, -1, S)
    attn = torch.bmm(ctx.transpose(1, 2), query)
    attn = torch.softmax(attn.reshape(B * S, T), dim=1).view(B, S, T)
    attn = torch.softmax(attn.transpose(1, 2).reshape(B * T, S) * gamma1, dim=1).view(B, T, S)
    return torch.bmm(ctx, attn.transpose(1, 2)), attn.view(B, T, ih, iw)


class GlobalAttentionGeneral(nn.Module):
    """GlobalAttention.py:72-121.  `reference_mask_order` keeps the reference's
    row-ordering quirk of the mask (:105-108); set it False for the per-sample mask."""

    def __init__(self, idf, cdf):
        super(GlobalAttentionGeneral, self).__init__()
        self.conv_context = conv1x1(cdf, idf)
        self.sm = nn.Softmax(dim=1)
        self.mask = None
        self.reference_mask_order = True

    def applyMask(self, mask):
        self.mask = mask  # batch x sourceL

    def forward(self, input, context):
        mode = 0 if self.reference_mask_order else 1
        out, att = ops.WordAttnFn.apply(input, context, self.conv_context.weight, self.mask, mode)
        return out, att


# ----------------------------------------------------------------------------
# generator
# ----------------------------------------------------------------------------
class CA_NET(nn.Module):
    """model.py:271-299.  The N(0,1) draw of reparametrize (:289-293) comes from
    torch's generator unless `eps` is injected (tests / golden parity)."""

    def __init__(self):
        super(CA_NET, self).__init__()
        self.t_dim = cfg.TEXT.EMBEDDING_DIM
        self.c_dim = cfg.GAN.CONDITION_DIM
        self.fc = nn.Linear(self.t_dim, self.c_dim * 4, bias=True)
        self.relu = GLU()
        self.eps = None

    def forward(self, text_embedding):
        h = _linear(text_embedding, self.fc)
        eps = self.eps
        if eps is None:
            eps = torch.randn((h.size(0), self.c_dim), dtype=torch.float32, device=h.device)
        return ops.CAFn.apply(h, eps)


class MAPPING_NET(nn.Module):
    """model.py:301-321 (6 layers) / model_bert.py:334-356 (8 layers)."""

    def __init__(self, n_layers=6):
        super(MAPPING_NET, self).__init__()
        self.z_dim = cfg.GAN.Z_DIM
        self.w_dim = cfg.GAN.W_DIM
        layers = [nn.Linear(self.z_dim, self.w_dim, bias=False)]
        layers += [nn.Linear(self.w_dim, self.w_dim, bias=False) for _ in range(n_layers - 1)]
        self.fc = nn.Sequential(*layers)

    def forward(self, z_code):
        x = z_code
        for lin in self.fc:
            x = _linear(x, lin)
        return x


class ADAIN_NORM(nn.Module):
    """model.py:324-339."""

    def __init__(self, ngf):
        super(ADAIN_NORM, self).__init__()
        self.norm = nn.InstanceNorm2d(ngf)
        self.style = nn.Linear(cfg.GAN.W_DIM, ngf * 2)

    def forward(self, h_code, w_code):
        return ops.AdainFn.apply(h_code, _linear(w_code, self.style))


class _FcBnGlu(nn.Sequential):
    """INIT_STAGE_G.fc: Linear(no bias), BatchNorm1d, GLU (model.py:353-356)."""

    @property
    def bn(self):
        return self[1]

    def forward(self, x):
        if not self[1].training and torch.is_grad_enabled() and \
                (x.requires_grad or self[0].weight.requires_grad or self[1].weight.requires_grad):
            # (checked here: grad mode is always off inside autograd.Function.forward)
            raise RuntimeError('INIT_STAGE_G.fc in eval mode is an inference path (running statistics, no backward): '
                               'call the generator under torch.no_grad()')
        return ops.FcBnGluFn.apply(x, self[0].weight, self[1].weight, self[1].bias, self)


class INIT_STAGE_G(nn.Module):
    """model.py:342-383; `cond_only` selects model_bert.py:377-425 (input = c_code)."""

    def __init__(self, ngf, ncf, cond_only=False):
        super(INIT_STAGE_G, self).__init__()
        self.gf_dim = ngf
        self.cond_only = cond_only
        self.in_dim = ncf if cond_only else cfg.GAN.Z_DIM + ncf
        self.fc = _FcBnGlu(nn.Linear(self.in_dim, ngf * 4 * 4 * 2, bias=False),
                           nn.BatchNorm1d(ngf * 4 * 4 * 2), _Slot())
        self.upsample1 = upBlock(ngf, ngf // 2)
        self.upsample2 = upBlock(ngf // 2, ngf // 4)
        self.upsample3 = upBlock(ngf // 4, ngf // 8)
        self.upsample4 = upBlock(ngf // 8, ngf // 16)

    def forward(self, *args):
        if self.cond_only:      # model_bert.py:402: forward(c_code, z_code, w_code)
            x = args[0]
        else:                   # model.py:363: forward(z_code, c_code)
            z_code, c_code = args[0], args[1]
            x = torch.cat((c_code, z_code), 1)
        out = self.fc(x)
        out = self.upsample1(out)
        out = self.upsample2(out)
        out = self.upsample3(out)
        return self.upsample4(out)


class NEXT_STAGE_G(nn.Module):
    """model.py:386-423 (`adain`) / model_bert.py:428-468 (`adain2`)."""

    def __init__(self, ngf, nef, ncf, adain_name='adain'):
        super(NEXT_STAGE_G, self).__init__()
        self.gf_dim, self.ef_dim, self.cf_dim = ngf, nef, ncf
        self.num_residual = cfg.GAN.R_NUM
        self._adain_name = adain_name
        self.att = GlobalAttentionGeneral(ngf, nef)
        setattr(self, adain_name, ADAIN_NORM(ngf))
        self.residual = nn.Sequential(*[ResBlock(ngf * 2) for _ in range(cfg.GAN.R_NUM)])
        self.upsample = upBlock(ngf * 2, ngf)
        self.return_attention = True

    def style_of(self, w_code):
        """ADAIN_NORM.style(w) (model.py:330,335): needs the style code only"""
        return _linear(w_code, getattr(self, self._adain_name).style)

    def keys_of(self, word_embs):
        """GlobalAttentionGeneral.conv_context(words) (GlobalAttention.py:97): needs the captions only"""
        return ops.CtxProjFn.apply(word_embs, self.att.conv_context.weight)

    def forward(self, h_code, c_code, w_code, word_embs, mask, style=None, keys=None):
        """style / keys: style_of(w_code) / keys_of(word_embs) when the caller has evaluated them already (beside the
        first stage, on the mapping network's stream: _GBase._styles)"""
        self.att.applyMask(mask)
        if style is None:
            style = self.style_of(w_code)
        mode = 0 if self.att.reference_mask_order else 1
        hc, att = ops.AttnAdainCatFn.apply(h_code, style, word_embs, self.att.conv_context.weight, mask,
                                           self.return_attention, mode, keys)
        out = hc
        for blk in self.residual:
            out = blk(out)
        out = self.upsample(out)
        return out, (att if self.return_attention else None)


class GET_IMAGE_G(nn.Module):
    """model.py:426-437."""

    def __init__(self, ngf):
        super(GET_IMAGE_G, self).__init__()
        self.gf_dim = ngf
        self.img = nn.Sequential(conv3x3(ngf, 3), _Slot())

    def forward(self, h_code):
        return ops.ImgHeadFn.apply(h_code, self.img[0].weight)


_MAP_STREAMS = {}        # device -> the side stream of the mapping network (module level: modules stay deep-copyable)


class _CrossStream(torch.autograd.Function):
    """Identity at the boundary between the mapping network's side stream and the main stream (experiment,
    SBA_FORK_MAPPING=2): tells the caching allocator about the cross-stream use in BOTH directions -- the forward value
    (allocated on the side stream, read on the main stream) and the gradient (allocated on the main stream, read by the
    mapping network's backward pass on the side stream)."""

    @staticmethod
    def forward(ctx, w, main, side):
        ctx.side = side
        w.record_stream(main)
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        g.record_stream(ctx.side)
        return g, None, None


class _GBase(nn.Module):
    def _build(self, n_map, cond_only, adain_name):
        ngf, nef, ncf = cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM, cfg.GAN.CONDITION_DIM
        self.ca_net = CA_NET()
        self.mapping_net = MAPPING_NET(n_map)
        self.branch_num = cfg.TREE.BRANCH_NUM
        if self.branch_num > 0:
            self.h_net1 = INIT_STAGE_G(ngf * 16, ncf, cond_only)
            self.img_net1 = GET_IMAGE_G(ngf)
        if self.branch_num > 1:
            self.h_net2 = NEXT_STAGE_G(ngf, nef, ncf, adain_name)
            self.img_net2 = GET_IMAGE_G(ngf)
        if self.branch_num > 2:
            self.h_net3 = NEXT_STAGE_G(ngf, nef, ncf, adain_name)
            self.img_net3 = GET_IMAGE_G(ngf)

    def set_return_attention(self, flag):
        """The B x L x H x W attention maps are unused in training (trainer.py:262); turning them
        off saves materialising them.  Default True = reference behaviour."""
        for m in self.modules():
            if isinstance(m, NEXT_STAGE_G):
                m.return_attention = flag

    # The style codes w = MAPPING_NET(z) are first read by stage 2 (AdaIN): the mapping network -- a chain of 6 / 8 dense
    # layers of ~13 us each, pure latency -- runs on a side stream beside CA_NET and the whole first stage, and autograd
    # replays its backward pass (another ~160 us chain) on that stream too, beside the first stage's backward pass
    # instead of at the very end of the generator's.  Same kernels, same operands, worth 0.2-0.4 ms of the 11 ms step.
    # Round 3 found a RACE in the unguarded fork (SBA_FORK_MAPPING=1: with two mapping calls, G_NET_MIX, 4 of 30 eager
    # runs produced one wrong weight gradient): the caching allocator re-using a block across the two streams.  The
    # fork is GUARDED (`_CrossStream`: record_stream in both directions at the stream boundary) and, since round 4, ON by
    # default: sbagan/stream_audit.py reports the unguarded fork's unjoined cross-stream frees and none for the guarded one
    # (tests/test_stream_audit_gpu.py::test_audit_flags_the_unguarded_mapping_fork), the whole eager step audits clean for
    # all three variants, and 40 eager passes per variant are bit-equal to the single-stream pass in the deterministic mode
    # (::test_mapping_fork_stress_bit_equal).  SBA_FORK_MAPPING=0 turns it off.
    fork_mapping = os.environ.get('SBA_FORK_MAPPING', '2') in ('1', '2')
    _fork_guard = os.environ.get('SBA_FORK_MAPPING', '2') == '2'
    on_image = None          # callable(i): called right after fake image i has been issued (the trainer forks the
    #                          update of discriminator i from that point instead of from the end of the forward pass)

    # Round 4, last third: the same side stream also evaluates what each later stage needs of w and of the captions BEFORE
    # it can start -- its AdaIN style projection (a 17 us dense layer of pure latency) and its attention's key projection
    # conv_context(words) (22 us) -- so that they leave the serial chain of the generator's forward pass, and (autograd
    # replays a node's backward on its forward's stream) their backward passes (ctx_proj_bwd_w 42 us, linear_bwd) leave the
    # serial chain of its backward pass.  Same kernels, same operands.  SBA_FORK_CTX=0: inside the stages, as before.
    fork_ctx = os.environ.get('SBA_FORK_CTX', '1') != '0'

    def _styles(self, *zs, word_embs=None):
        self._pre = None
        if not (self.fork_mapping and zs[0].is_cuda):
            return [self.mapping_net(z) for z in zs], None
        main = torch.cuda.current_stream()
        side = _MAP_STREAMS.get(zs[0].device)
        if side is None:
            side = _MAP_STREAMS[zs[0].device] = torch.cuda.Stream(device=zs[0].device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ws = [self.mapping_net(z) for z in zs]
            if self.fork_ctx and self._fork_guard and word_embs is not None and self.branch_num > 1:
                stages = [(self.h_net2, ws[0])] + ([(self.h_net3, ws[-1])] if self.branch_num > 2 else [])
                self._pre = [(st.style_of(w), st.keys_of(word_embs)) for st, w in stages]
        return ws, (main, side)

    image_stream = None      # callable(i) -> the stream image head i runs on, or None for the current one (the trainer names
    #                          the stream discriminator i is updated on for every image but the last: a head whose image
    #                          only that discriminator reads leaves the serial chain of the generator's forward pass, and
    #                          -- autograd replays a node's backward on its forward's stream -- of its backward pass)

    def _emit(self, fake_imgs, net, h):
        i = len(fake_imgs)
        st = self.image_stream(i) if (self.image_stream is not None and h.is_cuda) else None
        if st is None:
            img = net(h)
        else:
            main = torch.cuda.current_stream()
            st.wait_stream(main)
            with torch.cuda.stream(st):
                # h: produced on `main`, read on `st`; its gradient: produced on `st`, read on `main`
                img = net(_CrossStream.apply(h, st, main))
        fake_imgs.append(img)
        if self.on_image is not None:
            self.on_image(i)

    def _run(self, z1, ws, join, sent_emb, word_embs, mask):
        ops.reset_mask_cache()
        w2, w3 = ws[0], ws[-1]
        fake_imgs, att_maps = [], []
        c_code, mu, logvar = self.ca_net(sent_emb)
        if self.branch_num > 0:
            h = self.h_net1(c_code, z1, None) if self.h_net1.cond_only else self.h_net1(z1, c_code)
            self._emit(fake_imgs, self.img_net1, h)
        pre = [(None, None), (None, None)]
        if join is not None:
            join[0].wait_stream(join[1])
            if self._fork_guard:
                ws = [_CrossStream.apply(w, join[0], join[1]) for w in ws]
                w2, w3 = ws[0], ws[-1]
                if getattr(self, '_pre', None):
                    got = [(_CrossStream.apply(s, join[0], join[1]), _CrossStream.apply(k, join[0], join[1]))
                           for s, k in self._pre]
                    pre = got + pre[len(got):]
                    self._pre = None
        if self.branch_num > 1:
            h, att1 = self.h_net2(h, c_code, w2, word_embs, mask, style=pre[0][0], keys=pre[0][1])
            self._emit(fake_imgs, self.img_net2, h)
            if att1 is not None:
                att_maps.append(att1)
        if self.branch_num > 2:
            h, att2 = self.h_net3(h, c_code, w3, word_embs, mask, style=pre[1][0], keys=pre[1][1])
            self._emit(fake_imgs, self.img_net3, h)
            if att2 is not None:
                att_maps.append(att2)
        return fake_imgs, att_maps, mu, logvar


class G_NET(_GBase):
    """model.py:440-492: forward(z_code, sent_emb, word_embs, mask) ->
    (fake_imgs list, att_maps list, mu, logvar)."""

    def __init__(self):
        super(G_NET, self).__init__()
        self._build(6, False, 'adain')

    def forward(self, z_code, sent_emb, word_embs, mask):
        ws, join = self._styles(z_code, word_embs=word_embs)
        return self._run(z_code, ws, join, sent_emb, word_embs, mask)


class G_NET_BERT(_GBase):
    """model_bert.py G_NET: 8-layer mapping net, c-only initial stage, `adain2`."""

    def __init__(self):
        super(G_NET_BERT, self).__init__()
        self._build(8, True, 'adain2')

    def forward(self, z_code, sent_emb, word_embs, mask):
        ws, join = self._styles(z_code, word_embs=word_embs)
        return self._run(z_code, ws, join, sent_emb, word_embs, mask)


class G_NET_MIX(_GBase):
    """model_bert.py:485-539: z_code is 2 x B x nz; w(z[0]) styles stage 2, w(z[1]) stage 3."""

    def __init__(self):
        super(G_NET_MIX, self).__init__()
        self._build(8, True, 'adain2')

    def forward(self, z_code, sent_emb, word_embs, mask):
        ws, join = self._styles(z_code[0], z_code[1], word_embs=word_embs)
        return self._run(z_code, ws, join, sent_emb, word_embs, mask)


# ----------------------------------------------------------------------------
# discriminators
# ----------------------------------------------------------------------------
class _EncodeBy16(nn.Sequential):
    """encode_image_by_16times (model.py:560-578), same nn.Sequential indices."""

    def _layers(self):
        if '_ls' not in self.__dict__:
            self.__dict__['_ls'] = [_Layer(self[2], self[3], '4x4s2'), _Layer(self[5], self[6], '4x4s2'),
                                    _Layer(self[8], self[9], '4x4s2')]
        return self.__dict__['_ls']

    def forward(self, x, groups=1):
        h = ops.DStemFn.apply(x, self[0].weight)
        for l in self._layers():
            h = ops.ConvBNActFn.apply(h, l.conv.weight, l.bn.weight, l.bn.bias, l, '4x4s2', ACT_LRELU, None,
                                      groups)
        return h


def encode_image_by_16times(ndf):
    return _EncodeBy16(
        _conv(3, ndf, 4, 2, 1), _Slot(),
        _conv(ndf, ndf * 2, 4, 2, 1), nn.BatchNorm2d(ndf * 2), _Slot(),
        _conv(ndf * 2, ndf * 4, 4, 2, 1), nn.BatchNorm2d(ndf * 4), _Slot(),
        _conv(ndf * 4, ndf * 8, 4, 2, 1), nn.BatchNorm2d(ndf * 8), _Slot())


class D_GET_LOGITS(nn.Module):
    """model.py:581-607."""

    def __init__(self, ndf, nef, bcondition=False):
        super(D_GET_LOGITS, self).__init__()
        self.df_dim, self.ef_dim, self.bcondition = ndf, nef, bcondition
        if bcondition:
            self.jointConv = Block3x3_leakRelu(ndf * 8 + nef, ndf * 8)
        self.outlogits = nn.Sequential(_conv(ndf * 8, 1, 4, 4, 0, bias=True), _Slot())

    def forward(self, h_code, c_code=None):
        if self.bcondition and c_code is not None:
            h_c_code = self.jointConv(ops.CondCatFn.apply(h_code, c_code.reshape(-1, self.ef_dim)))
        else:
            h_c_code = h_code
        o = self.outlogits[0]
        return ops.LogitsFn.apply(h_c_code, o.weight, o.bias)


class _DBase(nn.Module):
    bucket_from = None      # D_NET128 / D_NET256: name of the module from which the gradient bucket "tail" starts
    _cuts = ()              # ... and the activations that enter it, one per forward call since clear_cuts()

    def clear_cuts(self, record=True):
        """start (or stop) recording the activations that enter the bucket tail"""
        self._cuts = [] if record else ()

    def _heads(self, b_jcu):
        ndf, nef = cfg.GAN.DF_DIM, cfg.TEXT.EMBEDDING_DIM
        self.UNCOND_DNET = D_GET_LOGITS(ndf, nef, bcondition=False) if b_jcu else None
        self.COND_DNET = D_GET_LOGITS(ndf, nef, bcondition=True)


class D_NET64(_DBase):
    """model.py:611-625."""

    def __init__(self, b_jcu=True):
        super(D_NET64, self).__init__()
        ndf = cfg.GAN.DF_DIM
        self.img_code_s16 = encode_image_by_16times(ndf)
        self._heads(b_jcu)

    def forward(self, x_var, groups=1):
        """groups > 1: x_var holds that many independent batches back to back (e.g. real | fake);
        BatchNorm treats each as its own batch, exactly like separate calls (see ops.ConvBNActFn)."""
        return self.img_code_s16(x_var, groups)


class D_NET128(_DBase):
    """model.py:629-648."""

    def __init__(self, b_jcu=True):
        super(D_NET128, self).__init__()
        ndf = cfg.GAN.DF_DIM
        self.img_code_s16 = encode_image_by_16times(ndf)
        self.img_code_s32 = downBlock(ndf * 8, ndf * 16)
        self.img_code_s32_1 = Block3x3_leakRelu(ndf * 16, ndf * 8)
        self._heads(b_jcu)

    # data-parallel gradient exchange in two buckets (sbagan.trainer.GANStep): the parameters from this module on --
    # 86 % of the network -- have their gradients complete when the backward pass reaches `_cut`
    bucket_from = 'img_code_s32'

    def forward(self, x_var, groups=1):
        x = self.img_code_s16(x_var, groups)
        if isinstance(self._cuts, list):
            self._cuts.append(x)
        x = self.img_code_s32(x, groups=groups)
        return self.img_code_s32_1(x, groups=groups)


class D_NET256(_DBase):
    """model.py:652-674."""

    def __init__(self, b_jcu=True):
        super(D_NET256, self).__init__()
        ndf = cfg.GAN.DF_DIM
        self.img_code_s16 = encode_image_by_16times(ndf)
        self.img_code_s32 = downBlock(ndf * 8, ndf * 16)
        self.img_code_s64 = downBlock(ndf * 16, ndf * 32)
        self.img_code_s64_1 = Block3x3_leakRelu(ndf * 32, ndf * 16)
        self.img_code_s64_2 = Block3x3_leakRelu(ndf * 16, ndf * 8)
        self._heads(b_jcu)

    bucket_from = 'img_code_s64'        # 60.6 M of the 71.9 M parameters (242 of 287 MB of gradients)

    def forward(self, x_var, groups=1):
        x = self.img_code_s16(x_var, groups)
        x = self.img_code_s32(x, groups=groups)
        if isinstance(self._cuts, list):
            self._cuts.append(x)
        x = self.img_code_s64(x, groups=groups)
        x = self.img_code_s64_1(x, groups=groups)
        return self.img_code_s64_2(x, groups=groups)


# ----------------------------------------------------------------------------
# text encoder (module API kept; frozen + eval() in GAN training, trainer.py:64-73;
# not a hand-written-kernel target: SURVEY.md section 2 row 7)
# ----------------------------------------------------------------------------
class RNN_ENCODER(nn.Module):
    """model.py:75-159: Embedding -> dropout -> bi-LSTM over packed sequences ->
    (words_emb B x nhidden x Lmax, sent_emb B x nhidden)."""

    def __init__(self, ntoken, ninput=300, drop_prob=0.5, nhidden=128, nlayers=1, bidirectional=True):
        super(RNN_ENCODER, self).__init__()
        self.n_steps = cfg.TEXT.WORDS_NUM
        self.ntoken, self.ninput, self.drop_prob = ntoken, ninput, drop_prob
        self.nlayers, self.bidirectional = nlayers, bidirectional
        self.rnn_type = cfg.RNN_TYPE
        self.num_directions = 2 if bidirectional else 1
        self.nhidden = nhidden // self.num_directions
        self.encoder = nn.Embedding(ntoken, ninput)
        self.drop = nn.Dropout(drop_prob)
        rnn = {'LSTM': nn.LSTM, 'GRU': nn.GRU}.get(self.rnn_type)
        if rnn is None:
            raise NotImplementedError
        self.rnn = rnn(ninput, self.nhidden, nlayers, batch_first=True, dropout=drop_prob,
                       bidirectional=bidirectional)
        self.encoder.weight.data.uniform_(-0.1, 0.1)

    def init_hidden(self, bsz):
        w = next(self.parameters()).data
        z = lambda: w.new_zeros(self.nlayers * self.num_directions, bsz, self.nhidden)
        return (z(), z()) if self.rnn_type == 'LSTM' else z()

    use_hip = True      # the hand-written bi-LSTM (csrc/text.hip): frozen forward AND the training path

    def _hip_shape_ok(self, captions):
        return (self.use_hip and captions.is_cuda and self.rnn_type == 'LSTM' and self.nlayers == 1
                and self.bidirectional and self.nhidden in (64, 128) and self.ninput % 4 == 0)

    def _hip_ok(self, captions):
        """the inference kernel (embedding gather fused into the input projection, nothing saved)"""
        return (self._hip_shape_ok(captions) and not torch.is_grad_enabled()
                and (not self.training or self.drop_prob == 0))

    def _packed_lstm_weights(self):
        """[2][4H][*] stacks of the forward / reverse direction parameters, rebuilt when they change."""
        r = self.rnn
        ps = (r.weight_ih_l0, r.weight_ih_l0_reverse, r.weight_hh_l0, r.weight_hh_l0_reverse,
              r.bias_ih_l0, r.bias_ih_l0_reverse, r.bias_hh_l0, r.bias_hh_l0_reverse)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, '_lstm_key', None) != key:
            self._lstm_pack = tuple(torch.stack((ps[i].detach().float(), ps[i + 1].detach().float())).contiguous()
                                    for i in (0, 2, 4, 6))
            self._lstm_key = key
        return self._lstm_pack

    def forward(self, captions, cap_lens, hidden, mask=None, max_len=None, out=None):
        """`max_len` (optional, host int = the reference's max(cap_lens)): without it the sync-free path returns
        words_emb over the full padded width T; the extra columns are zeros and are masked / sliced away by
        every consumer (GlobalAttention mask, words_loss cap_lens), so results are unchanged.  `out` (optional,
        HIP path only): (words_emb, sent_emb) tensors to write into."""
        if self._hip_ok(captions):
            w_ih, w_hh, b_ih, b_hh = self._packed_lstm_weights()
            return ops.lstm_bidir_forward(captions, cap_lens, self.encoder.weight.detach().float(), w_ih, w_hh, b_ih,
                                          b_hh, hidden, max_len, out)
        if self._hip_shape_ok(captions):
            # training path (pretrain_DAMSM.py:79-81): embedding + dropout are ordinary autograd ops, the packed
            # bi-LSTM and its back-propagation through time are the HIP kernels behind ops.LstmBidirTrainFn
            r = self.rnn
            emb = self.drop(self.encoder(captions))
            w_ih = torch.stack((r.weight_ih_l0, r.weight_ih_l0_reverse))
            w_hh = torch.stack((r.weight_hh_l0, r.weight_hh_l0_reverse))
            b_ih = torch.stack((r.bias_ih_l0, r.bias_ih_l0_reverse))
            b_hh = torch.stack((r.bias_hh_l0, r.bias_hh_l0_reverse))
            L = captions.size(1) if max_len is None else int(max_len)
            h0 = c0 = None
            if hidden is not None:
                h0, c0 = hidden[0].detach().float().contiguous(), hidden[1].detach().float().contiguous()
            return ops.LstmBidirTrainFn.apply(emb, cap_lens, w_ih, w_hh, b_ih, b_hh, h0, c0, L)
        if captions.is_cuda and self.use_hip:
            raise RuntimeError('RNN_ENCODER on the GPU supports the one-layer bidirectional LSTM with nhidden 128 / 256 '
                               '(cfg.TEXT.EMBEDDING_DIM) and ninput % 4 == 0; set use_hip = False to run this '
                               'configuration through torch.nn.%s' % self.rnn_type)
        from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
        emb = self.drop(self.encoder(captions))
        lens = cap_lens.data.tolist()
        emb = pack_padded_sequence(emb, lens, batch_first=True)
        output, hidden = self.rnn(emb, hidden)
        output = pad_packed_sequence(output, batch_first=True)[0]
        words_emb = output.transpose(1, 2)
        h = hidden[0] if self.rnn_type == 'LSTM' else hidden
        sent_emb = h.transpose(0, 1).contiguous().view(-1, self.nhidden * self.num_directions)
        return words_emb, sent_emb
