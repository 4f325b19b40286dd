"""CNN_ENCODER (model.py:162-267) forward + backward-data on the HIP kernels.

The encoder is frozen in GAN training (trainer.py:57-63) but sits inside every generator
step (losses.py:190) and back-propagates to the fake image.  Here the Inception-v3 trunk runs on
the implicit-GEMM kernel (sba_conv_igemm_bias): BatchNorm(eval, eps 1e-3) is folded into the
packed weights + a bias, ReLU is the conv epilogue, the Inception concats are channel-slice
writes, and the backward is ReLU-mask -> data-gradient conv with in-place accumulation into the
block input's gradient.  Resize / stem / pools: csrc/encoder.hip.

`InceptionHIP(enc)` wraps a sbagan.encoders.CNN_ENCODER (whose state_dict layout matches the
reference's checkpoints) and is a drop-in callable: (B x 3 x S x S f32) ->
(B x nef x 17 x 17 f32, B x nef f32).
"""
import ctypes
import os

import torch

from . import _lib, ops
from ._lib import ConvGeom, ConvGroupItem, call

BN_EPS = 1e-3
SAVE_ARGMAX = os.environ.get('SBA_ENC_SAVE_ARGMAX', '1') != '0'    # max-pool forward keeps the window argmax for the backward
GROUP_MIN_TILES = int(os.environ.get('SBA_ENC_GROUP_MIN_TILES', '512'))
GROUP_TILE = int(os.environ.get('SBA_ENC_GROUP_TILE', '0'))             # tuning aid: force the grouped launches' tile id
FRAG_STEM = os.environ.get('SBA_ENC_FRAG_STEM', '1') != '0'         # register-weight halo kernel for the trunk's first 3 x 3 layers
GROUP_T7_MIN = int(os.environ.get('SBA_ENC_GROUP_T7_MIN', '0'))         # 128 x 128 tiles (tile id 7) when a level has at least this many (0 = never)


def _pad32(c):
    return (c + 31) // 32 * 32


class _Act(object):
    """NHWC activation: `t` is [N, H, W, Ct]; this view covers channels [coff, coff + C)."""

    def __init__(self, t, coff=0, C=None):
        self.t, self.coff = t, coff
        self.C = t.shape[3] - coff if C is None else C
        self.grad = None            # gradient w.r.t. the WHOLE tensor t (shared by all slices)
        self.grad_ready = False

    @property
    def shape(self):
        return self.t.shape


class _Conv(object):
    """One BasicConv2d (conv, BN eval, ReLU) or a plain conv: folded + packed operands."""

    def __init__(self, conv, bn, dtype, relu=True):
        w = conv.weight.detach().float()
        O, I, KH, KW = w.shape
        dev = w.device
        if bn is not None:
            s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            b = bn.bias.detach().float() - bn.running_mean.detach().float() * s
            w = w * s.view(-1, 1, 1, 1)
        else:
            b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(O, device=dev)
        self.O, self.I, self.KH, self.KW = O, I, KH, KW
        self.Op, self.Ip = _pad32(O), _pad32(I)
        self.bn, self.conv_mod, self._w_raw = bn, conv, None
        self.stride = conv.stride[0]
        self.ph, self.pw = conv.padding
        self.relu = relu
        taps = KH * KW
        wp = torch.zeros((self.Op, taps, self.Ip), dtype=torch.float32, device=dev)
        wp[:O, :, :I] = w.permute(0, 2, 3, 1).reshape(O, taps, I)
        self.bias = torch.zeros(self.Op, dtype=torch.float32, device=dev)
        self.bias[:O] = b
        self.w_fwd = wp.to(dtype).contiguous()
        wt = wp.permute(2, 1, 0).contiguous()                      # [Ip][tap][Op]
        if self.stride == 1:
            self.w_dgrad = [wt.to(dtype).contiguous()]
            self.dtaps = [[(self.ph - t // KW, self.pw - t % KW) for t in range(taps)]]
        else:       # stride 2 (3x3, pad 0): four parity classes of the input grid
            self.w_dgrad, self.dtaps = [], []
            for py in range(2):
                for px in range(2):
                    sel = [(kh, kw) for kh in range(KH) if (kh - py) % 2 == 0 for kw in range(KW) if (kw - px) % 2 == 0]
                    idx = torch.tensor([kh * KW + kw for kh, kw in sel], device=dev)
                    self.w_dgrad.append(wt[:, idx, :].to(dtype).contiguous())
                    self.dtaps.append([((py - kh) // 2, (px - kw) // 2) for kh, kw in sel])


def _raw_weights(L, dtype):
    """UNFOLDED packed weights [O][taps][Ip] of a BasicConv2d (training-mode BatchNorm: the conv output is normalised
    with batch statistics, so the running statistics cannot be folded in)"""
    if L._w_raw is None or L._w_raw.dtype != dtype:
        w = L.conv_mod.weight.detach().float()
        O, I, KH, KW = w.shape
        wp = torch.zeros((O, KH * KW, L.Ip), dtype=torch.float32, device=w.device)
        wp[:, :, :I] = w.permute(0, 2, 3, 1).reshape(O, KH * KW, I)
        L._w_raw = wp.to(dtype).contiguous()
    return L._w_raw


class _FusedHeads(object):
    """The 1x1 BasicConv2d layers at the head of several branches of one Inception block read the same
    input: ONE conv with their output channels concatenated forward, and ONE data-gradient with their
    output-gradients concatenated along K backward (the per-branch gradients add up inside the GEMM)."""

    def __init__(self, parts):
        assert all(c.KH == 1 and c.KW == 1 and c.stride == 1 and c.relu and c.Ip == parts[0].Ip for c in parts)
        self.parts = parts
        self.I, self.Ip = parts[0].I, parts[0].Ip
        self.O = self.Op = sum(c.Op for c in parts)
        self.KH = self.KW = 1
        self.stride, self.ph, self.pw, self.relu = 1, 0, 0, True
        self.bias = torch.cat([c.bias for c in parts]).contiguous()
        self.w_fwd = torch.cat([c.w_fwd for c in parts], 0).contiguous()               # [sum Op][1][Ip]
        self.w_dgrad = [torch.cat([c.w_dgrad[0] for c in parts], 2).contiguous()]      # [Ip][1][sum Op]
        self.dtaps = [[(0, 0)]]


def _geom(N, IH, IW, Cin, OH, OW, Cout, taps, sy=1, OHs=None, OWs=None, osy=1, ooy=0, oox=0, xcs=0, xco=0, ycs=0,
          yco=0, relu=0):
    g = ConvGeom()
    g.N, g.IH, g.IW, g.Cin, g.OH, g.OW, g.Cout = N, IH, IW, Cin, OH, OW, Cout
    g.OHs, g.OWs = OH if OHs is None else OHs, OW if OWs is None else OWs
    g.sy = g.sx = sy
    g.osy = g.osx = osy
    g.ooy, g.oox = ooy, oox
    g.ups = 0
    g.ntaps = len(taps)
    for t, (a, b) in enumerate(taps):
        g.ty[t], g.tx[t] = a, b
    g.x_cstride, g.x_coff, g.y_cstride, g.y_coff, g.relu = xcs, xco, ycs, yco, relu
    return g


class InceptionHIP(object):
    def __init__(self, enc, dtype=None):
        self.enc = enc
        self.dtype = dtype or ops.compute_dtype()
        self.nef = enc.nef
        self._convs = {}
        self._frag, self._frag_keep = {}, []
        self._fused = {}
        self._geoms = {}
        self._side, self._branch_ops, self._block = None, None, None
        self.parallel = os.environ.get('SBA_ENC_PARALLEL', '1') == '1'      # Inception branches on side streams
        # GROUPED launches (bf16): the branches of a block are issued level by level on ONE stream, and the implicit-GEMM
        # convs of one level -- independent of each other -- go out as one sba_conv_igemm_group grid instead of one
        # 120..273-workgroup launch each (hipGraph replay runs the side streams back to back anyway, ROCm 7.2)
        self.group = os.environ.get('SBA_ENC_GROUP', '1') == '1' and self.dtype == torch.bfloat16
        self._thunks = None          # forward: launch closures of the branch being recorded
        self._train = False          # trunk_features(train=True): BatchNorm with batch statistics (DAMSM pre-training)
        self._keep = []
        self._pending = None         # implicit-GEMM launches collected for the current level
        dev = next(enc.parameters()).device
        self.device = dev
        for name, m in enc.named_modules():
            if m.__class__.__name__ == 'BasicConv2d':
                self._convs[name] = _Conv(m.conv, m.bn, self.dtype)
        self._convs['emb_features'] = _Conv(enc.emb_features, None, self.dtype, relu=False)
        lin = enc.emb_cnn_code

        class _L(object):
            pass
        fake = _L()
        fake.weight = lin.weight.detach().view(lin.out_features, lin.in_features, 1, 1)
        fake.bias, fake.stride, fake.padding = lin.bias, (1, 1), (0, 0)
        self._convs['emb_cnn_code'] = _Conv(fake, None, self.dtype, relu=False)
        stem = enc.Conv2d_1a_3x3
        s = stem.bn.weight.detach().float() / torch.sqrt(stem.bn.running_var.detach().float() + stem.bn.eps)
        self.stem_w = (stem.conv.weight.detach().float() * s.view(-1, 1, 1, 1)).contiguous(
            memory_format=torch.channels_last)
        self.stem_b = (stem.bn.bias.detach().float() - stem.bn.running_mean.detach().float() * s).contiguous()

    def refold(self):
        """rebuild the folded (eval-mode) operands from the module's current BatchNorm buffers"""
        for name, m in self.enc.named_modules():
            if m.__class__.__name__ == 'BasicConv2d':
                self._convs[name] = _Conv(m.conv, m.bn, self.dtype)
        self._fused = {}
        self._frag, self._frag_keep = {}, []        # (keyed by the old operands' addresses)
        stem = self.enc.Conv2d_1a_3x3
        s = stem.bn.weight.detach().float() / torch.sqrt(stem.bn.running_var.detach().float() + stem.bn.eps)
        self.stem_w = (stem.conv.weight.detach().float() * s.view(-1, 1, 1, 1)).contiguous(
            memory_format=torch.channels_last)
        self.stem_b = (stem.bn.bias.detach().float() - stem.bn.running_mean.detach().float() * s).contiguous()

    # ------------------------------------------------------------------ primitive ops
    def _dt(self):
        return _lib.SBA_BF16 if self.dtype == torch.bfloat16 else _lib.SBA_F32

    def _new(self, N, H, W, C):
        if self._train:     # padding channels are not written by the BatchNorm pass: they must read as zeros
            return torch.zeros((N, H, W, C), dtype=self.dtype, device=self.device)
        return torch.empty((N, H, W, C), dtype=self.dtype, device=self.device)

    def _bn_relu_train(self, y, bn, rows, O, out_t, ocs, oco, stats=None):
        """BatchNorm2d(train: batch statistics, eps 1e-3, momentum 0.1, running statistics updated in place) + ReLU of
        a dense [rows][O] tensor y into channels [oco, oco + O) of out_t"""
        dt = self._dt()
        if stats is None:
            stats = torch.zeros((ops.BN_STAT_SLOTS, 2 * O), dtype=torch.float32, device=self.device)
            call('sba_bn_stats', dt, y.data_ptr(), stats.data_ptr(), rows, 1, O, ops._stream())
        aux = torch.empty((4, O), dtype=torch.float32, device=self.device)
        call('sba_bn_act_fwd', dt, y.data_ptr(), stats.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
             bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(), aux.data_ptr(),
             None, out_t.data_ptr(), rows, 1, O, _lib.ACT_RELU, ocs, oco, BN_EPS, 0.1, 1, ops._stream())

    def _conv_train(self, L, x, out, OH, OW):
        """BasicConv2d in TRAINING mode (pretrain_DAMSM.py:51 cnn_model.train(); weights frozen, model.py:174-175):
        conv with the unfolded weights (statistics of the f32 accumulators in its epilogue) -> BatchNorm + ReLU"""
        N, H, W, Ct = x.shape
        y = torch.empty((N, OH, OW, L.O), dtype=self.dtype, device=self.device)
        stats = torch.zeros((ops.BN_STAT_SLOTS, 2 * L.O), dtype=torch.float32, device=self.device)
        taps = [(t // L.KW - L.ph, t % L.KW - L.pw) for t in range(L.KH * L.KW)]
        g = _geom(N, H, W, L.Ip, OH, OW, L.O, taps, sy=L.stride, xcs=Ct, xco=x.coff)
        ws = ops.workspace(self.device)
        call('sba_conv_igemm', self._dt(), x.t.data_ptr(), _raw_weights(L, self.dtype).data_ptr(), y.data_ptr(), None,
             stats.data_ptr(), ctypes.byref(g), ws.data_ptr(), ops.WORKSPACE_BYTES, ops._stream())
        self._bn_relu_train(y, L.bn, N * OH * OW, L.O, out.t, out.shape[3], out.coff, stats)

    def _frag_for(self, w, g):
        """the fragment-major copy of the packed weights `w` ([R][9][K], frozen) when this launch goes to the register-weight
        halo-tile kernel (include/sbagan_hip.h: sba_conv_geom.w_layout): stride-1 3 x 3 convs on the big maps of the trunk's
        first layers (>= 64 pixels wide) and their data gradients; None otherwise"""
        if not (ops.FRAG_WEIGHTS and FRAG_STEM) or self.dtype != torch.bfloat16 or g.ntaps != 9 or g.sy != 1 or \
                g.osy != 1 or g.OW < 64 or g.OWs != g.OW or w.dim() != 3 or w.shape[0] % 32 or w.shape[2] % 32:
            return None
        key = (w.data_ptr(), g.OH, g.OW, g.IH, g.IW, g.N)
        ent = self._frag.get(key)
        if ent is None:
            plan = (ctypes.c_int * 3)()
            g.w_layout = 1
            try:
                rc = _lib.lib.sba_conv_igemm_plan(_lib.SBA_BF16, ctypes.byref(g), ops.WORKSPACE_BYTES, plan)
            finally:
                g.w_layout = 0
            ent = False
            if rc == 0 and plan[0] in (0, 4):
                R, taps, K = w.shape
                dst = self._frag.get(('w', w.data_ptr()))
                if dst is None:
                    dst = torch.zeros((R + 63) // 64 * 64 * taps * K, dtype=w.dtype, device=w.device)
                    ops._pack_frag([(w, dst, R, taps, K)], w.device)
                    self._frag[('w', w.data_ptr())] = dst
                    self._frag_keep.append(w)          # (the key is its address)
                ent = dst
            self._frag[key] = ent
        return ent if ent is not False else None

    def _igemm(self, x_ptr, w, y_ptr, addend_ptr, bias, g, mask_ptr=None):
        wf = self._frag_for(w, g)
        if wf is not None:          # register-weight halo kernel: a launch of its own (never part of a grouped level)
            g.w_layout = 1
            try:
                call('sba_conv_igemm_bias', self._dt(), x_ptr, wf.data_ptr(), y_ptr, addend_ptr, None,
                     None if bias is None else bias.data_ptr(), mask_ptr, ctypes.byref(g), None, 0, ops._stream())
            finally:
                g.w_layout = 0
            return
        if self._pending is not None:
            self._pending.append((x_ptr, w, y_ptr, addend_ptr, bias, g, mask_ptr))
            return
        ws = ops.workspace(self.device)
        ops.tune_geom(g, self._dt())
        call('sba_conv_igemm_bias', self._dt(), x_ptr, w.data_ptr(), y_ptr, addend_ptr,
             None, None if bias is None else bias.data_ptr(), mask_ptr, ctypes.byref(g), ws.data_ptr(),
             ops.WORKSPACE_BYTES, ops._stream())

    def _launch(self, fn):
        """run a launch closure now, or keep it for the level-by-level issue of the block being recorded"""
        if self._thunks is not None:
            self._thunks.append(fn)
        else:
            fn()

    def _flush(self):
        """issue the implicit-GEMM launches collected for one level as one grid (several when there are more than
        SBA_GROUP_MAX)"""
        items, self._pending = self._pending, None
        grp = items if len(items) >= 2 else []
        taken = set(id(it) for it in grp)
        for it in items:
            if id(it) not in taken:
                self._igemm(*it)
        for i0 in range(0, len(grp), _lib.GROUP_MAX):
            part = grp[i0:i0 + _lib.GROUP_MAX]
            if len(part) == 1:
                self._igemm(*part[0])
                continue
            arr = (ConvGroupItem * len(part))()
            m128 = m96 = m7 = 0
            for a, (x_ptr, w, y_ptr, addend_ptr, bias, g, mask_ptr) in zip(arr, part):
                a.x, a.w, a.y, a.addend = x_ptr, w.data_ptr(), y_ptr, addend_ptr
                a.bias = None if bias is None else bias.data_ptr()
                a.relu_mask = mask_ptr
                a.g = ctypes.pointer(g)
                M = g.N * g.OHs * g.OWs
                ny = (g.Cout + 63) // 64
                m128 += ((M + 127) // 128) * ny
                m96 += ((M + 95) // 96) * ny
                m7 += ((M + 127) // 128) * ((g.Cout + 127) // 128)
            # bigger tiles move fewer bytes L2 -> LDS per FLOP; use them when they still fill two workgroups per CU
            tile = 5 if m128 >= GROUP_MIN_TILES else (3 if m96 >= GROUP_MIN_TILES else 1)
            if GROUP_T7_MIN and m7 >= GROUP_T7_MIN:
                tile = 7
            if GROUP_TILE:
                tile = GROUP_TILE
            call('sba_conv_igemm_group', _lib.SBA_BF16, len(part), arr, tile, ops._stream())
            if ops.IGEMM_LOG is not None:       # (bench.py's per-kernel accounting: these ran in the grouped kernel)
                ops.IGEMM_LOG.append(('group', tile, [it[5] for it in part]))
        self._keep = []             # temporaries the collected launches read (ReLU-masked gradients)

    def conv(self, name, x, out=None):
        """y = relu(conv(x) + bias) written to `out` (an _Act slice) or a new tensor."""
        return self.conv_L(self._convs[name], x, out, name)

    def fused(self, names):
        key = tuple(names)
        f = self._fused.get(key)
        if f is None:
            f = self._fused[key] = _FusedHeads([self._convs[n] for n in names])
        return f

    def conv_L(self, L, x, out=None, name='fused'):
        N, H, W, Ct = x.shape
        assert x.C == L.Ip, (name, x.C, L.Ip)
        OH = (H + 2 * L.ph - L.KH) // L.stride + 1
        OW = (W + 2 * L.pw - L.KW) // L.stride + 1
        if out is None:
            out = _Act(self._new(N, OH, OW, L.Op))
        assert out.C == L.Op and out.shape[1] == OH and out.shape[2] == OW, (name, out.C, L.Op)
        if self._train and getattr(L, 'bn', None) is not None:
            self._conv_train(L, x, out, OH, OW)
            return out
        taps = [(t // L.KW - L.ph, t % L.KW - L.pw) for t in range(L.KH * L.KW)]
        g = _geom(N, H, W, L.Ip, OH, OW, L.Op, taps, sy=L.stride, xcs=Ct, xco=x.coff, ycs=out.shape[3],
                  yco=out.coff, relu=1 if L.relu else 0)
        xp, yp = x.t.data_ptr(), out.t.data_ptr()
        self._launch(lambda: self._igemm(xp, L.w_fwd, yp, None, L.bias, g))
        self._record(('conv', L, x, out))
        self._consume(x)
        if L.relu:
            self._relu_slices.add(self._key(out))
        return out

    def _record(self, op):
        (self._branch_ops if self._branch_ops is not None else self.tape).append(op)

    def _consume(self, a):
        """one more reader of tensor a.t: the backward of the LAST reader to run completes d(loss)/d(a.t)
        and -- when it is a data-gradient conv and a.t came out of ReLU convs -- applies their ReLU mask in
        its epilogue, so the producers skip the separate relu_bwd pass"""
        k = self._key(a)
        self._readers[k] = self._readers.get(k, 0) + 1

    @staticmethod
    def _key(a):
        return (id(a.t), a.coff, a.C)

    @staticmethod
    def _covered(keys, a):
        """True when the channel range of view `a` is covered by the union of the views in `keys`"""
        tid, lo, hi = id(a.t), a.coff, a.coff + a.C
        for c0, c1 in sorted((c0, c0 + n) for (t, c0, n) in keys if t == tid):
            if c0 <= lo < c1:
                lo = c1
        return lo >= hi

    def _is_masked(self, a):
        """the ReLU mask was already applied to every channel of view `a` of its gradient"""
        return self._covered(self._masked, a)

    def _has_grad(self, a):
        """every channel of view `a` of the gradient tensor holds a value.  Tracked per VIEW: several views
        of one tensor (fused-head temps, the concat proper) are written by different ops, and the first
        write into each must not accumulate onto uninitialised memory"""
        return self._covered(self._filled, a)

    def _grad_of(self, a):
        """gradient buffer of the tensor behind activation `a`, and whether it already holds a value"""
        if a.grad is None:
            key = id(a.t)
            holder = self._grads.get(key)
            if holder is None:
                holder = [torch.empty_like(a.t)]             # gradient of the whole tensor
                self._grads[key] = holder
            a.grad = holder
        return a.grad

    def _conv_bwd(self, L, x, out):
        gy = self._grad_of(out)
        assert self._has_grad(out), 'gradient of a conv output was never produced'
        N, OH, OW, Ct_o = out.shape
        dt = self._dt()
        if L.relu and not self._is_masked(out):
            dpre = self._new(N, OH, OW, L.Op)
            if self._pending is not None:
                self._keep.append(dpre)     # read by a launch that is issued later (at the end of the level)
            call('sba_relu_bwd', dt, out.t.data_ptr(), gy[0].data_ptr(), dpre.data_ptr(), N * OH * OW, L.Op, Ct_o,
                 out.coff, Ct_o, out.coff, ops._stream())
            dsrc, dcs, dco = dpre, L.Op, 0
        else:       # no ReLU, or its mask was already applied by the data-gradient(s) that completed gy
            dsrc, dcs, dco = gy[0], Ct_o, out.coff
        gx = self._grad_of(x)
        _, H, W, Ct_x = x.shape
        addend = gx[0].data_ptr() if self._has_grad(x) else None
        # last reader of this view of x.t to run its backward: fold the ReLU mask of its producers into
        # this epilogue (the mask tensor is indexed exactly like the output, so slices work unchanged)
        kx = self._key(x)
        self._readers[kx] -= 1
        final = self._readers[kx] == 0 and kx in self._relu_slices
        mask = x.t.data_ptr() if final else None
        if L.stride == 1:
            g = _geom(N, OH, OW, L.Op, H, W, L.Ip, L.dtaps[0], xcs=dcs, xco=dco, ycs=Ct_x, yco=x.coff)
            self._igemm(dsrc.data_ptr(), L.w_dgrad[0], gx[0].data_ptr(), addend, None, g, mask)
        else:
            for cls in range(4):
                py, px = cls // 2, cls % 2
                OHs, OWs = (H - py + 1) // 2, (W - px + 1) // 2
                if OHs <= 0 or OWs <= 0 or not L.dtaps[cls]:
                    final = False
                    continue
                g = _geom(N, OH, OW, L.Op, H, W, L.Ip, L.dtaps[cls], OHs=OHs, OWs=OWs, osy=2, ooy=py, oox=px,
                          xcs=dcs, xco=dco, ycs=Ct_x, yco=x.coff)
                self._igemm(dsrc.data_ptr(), L.w_dgrad[cls], gx[0].data_ptr(), addend, None, g, mask)
        self._filled.add(kx)
        if final:
            self._masked.add(kx)

    def maxpool(self, x, out=None):
        N, H, W, Ct = x.shape
        OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        if out is None:
            out = _Act(self._new(N, OH, OW, x.C))
        arg = None
        xp, yp, dt, C, ocs, oco, xco = x.t.data_ptr(), out.t.data_ptr(), self._dt(), x.C, out.shape[3], out.coff, x.coff
        if SAVE_ARGMAX and getattr(self, '_want_grad', False):
            arg = torch.empty((N, OH, OW, x.C), dtype=torch.uint8, device=self.device)
            ap = arg.data_ptr()
            self._launch(lambda: call('sba_maxpool3x3s2_fwd_arg', dt, xp, yp, ap, N, H, W, C, Ct, xco, ocs, oco,
                                      ops._stream()))
        else:
            self._launch(lambda: call('sba_maxpool3x3s2_fwd', dt, xp, yp, N, H, W, C, Ct, xco, ocs, oco, ops._stream()))
        self._record(('maxpool', arg, x, out))
        self._consume(x)
        return out

    def _maxpool_bwd(self, x, out, arg=None):
        gy, gx = self._grad_of(out), self._grad_of(x)
        N, H, W, Ct = x.shape
        kx = self._key(x)
        self._readers[kx] -= 1
        # the last reader of a ReLU output to run its backward applies the ReLU's mask (as the data-gradient convs do in
        # their epilogue): the stem's two pools, whose producers then skip the separate relu_bwd pass
        final = arg is not None and self._readers[kx] == 0 and kx in self._relu_slices
        if arg is not None:
            call('sba_maxpool3x3s2_bwd_arg', self._dt(), arg.data_ptr(), gy[0].data_ptr(), gx[0].data_ptr(), N, H, W, x.C,
                 out.shape[3], out.coff, Ct, x.coff, 1 if self._has_grad(x) else 0, x.t.data_ptr() if final else None,
                 ops._stream())
        else:
            call('sba_maxpool3x3s2_bwd', self._dt(), x.t.data_ptr(), gy[0].data_ptr(), gx[0].data_ptr(), N, H, W, x.C, Ct,
                 x.coff, out.shape[3], out.coff, Ct, x.coff, 1 if self._has_grad(x) else 0, ops._stream())
        self._filled.add(kx)
        if final:
            self._masked.add(kx)

    def avgpool(self, x):
        N, H, W, Ct = x.shape
        out = _Act(self._new(N, H, W, x.C))
        xp, yp, dt, C, xco = x.t.data_ptr(), out.t.data_ptr(), self._dt(), x.C, x.coff
        self._launch(lambda: call('sba_avgpool3x3', dt, xp, yp, N, H, W, C, Ct, xco, C, 0, 0, ops._stream()))
        self._record(('avgpool', None, x, out))
        self._consume(x)
        return out

    def _avgpool_bwd(self, x, out):
        gy, gx = self._grad_of(out), self._grad_of(x)
        N, H, W, Ct = x.shape
        call('sba_avgpool3x3', self._dt(), gy[0].data_ptr(), gx[0].data_ptr(), N, H, W, x.C, out.shape[3], out.coff,
             Ct, x.coff, 1 if self._has_grad(x) else 0, ops._stream())
        self._filled.add(self._key(x))
        self._readers[self._key(x)] -= 1

    # ------------------------------------------------------------------ Inception blocks
    # The branches of a block are independent given its input: each runs on its own HIP stream
    # (forward and backward), which matters because the individual convs (M = 20*35^2 ... 20*8^2
    # pixels) are far too small to fill 256 CUs one at a time.
    def _streams(self):
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(4)]
        return self._side

    class _Branch(object):
        def __init__(self, runner, k):
            self.r, self.k = runner, k

        def __enter__(self):
            r = self.r
            r._branch_ops = []
            if r.group:
                r._thunks = []              # the branch's launches are issued level by level in _end_block
            elif r.parallel:
                st = r._streams()[self.k]
                st.wait_stream(r._main)
                self.ctx = torch.cuda.stream(st)
                self.ctx.__enter__()
            return self

        def __exit__(self, *a):
            r = self.r
            if r.group:
                r._block_thunks.append(r._thunks)
                r._thunks = None
            elif r.parallel:
                self.ctx.__exit__(*a)
            r._block.append((self.k, r._branch_ops))
            r._branch_ops = None

    def _begin_block(self, x):
        self._main = torch.cuda.current_stream()
        self._block = []
        self._block_x = x
        self._block_pre = []
        self._block_thunks = []

    def _heads(self, names, x, ext):
        """the fused 1x1 head convs of a block: outputs occupy channels [0, sum Op) of `ext`; returns the
        per-branch views in the order of `names`"""
        f = self.fused(names)
        self._branch_ops = self._block_pre                 # recorded as a main-stream op of the block
        if self._train:     # training-mode BatchNorm: no folded (hence no fused) operands -- one conv per head
            off = 0
            for c in f.parts:
                self.conv_L(c, x, _Act(ext, off, c.Op))
                off += c.Op
        else:
            self.conv_L(f, x, _Act(ext, 0, f.Op))
        self._branch_ops = None
        views, off = [], 0
        for c in f.parts:
            v = _Act(ext, off, c.Op)
            self._relu_slices.add(self._key(v))
            views.append(v)
            off += c.Op
        return views

    def _end_block(self, cat_view):
        if self.group:
            # level l = the l-th launch of every branch: independent of each other, the convs among them as one grid
            for lvl in range(max(len(t) for t in self._block_thunks)):
                self._pending = []
                for t in self._block_thunks:
                    if lvl < len(t):
                        t[lvl]()
                self._flush()
            self._block_thunks = []
        elif self.parallel:
            for k, _ in self._block:
                self._main.wait_stream(self._streams()[k])
        self._relu_slices.add(self._key(cat_view))          # every slice of the concat is a ReLU (or max-pool of ReLU) output
        self.tape.append(('block', self._block, self._block_x, self._block_pre))
        self._block = None
        return cat_view

    # Layout of a block's output tensor `ext`: [temps of the fused heads | branch1x1 | other branches ...];
    # the concat proper is the channel slice behind the temps (the next block reads it through
    # x_cstride / x_coff), so that the fused head conv writes ONE contiguous channel range.
    def _A(self, p, x, pf):
        N, H, W, _ = x.shape
        names = [p + '.branch5x5_1', p + '.branch3x3dbl_1', p + '.branch1x1']
        T = sum(self._convs[n].Op for n in names[:-1])
        ext = self._new(N, H, W, T + 64 + 64 + 96 + pf)
        self._begin_block(x)
        t5, t3, _b1 = self._heads(names, x, ext)
        with self._Branch(self, 1):
            self.conv(p + '.branch5x5_2', t5, _Act(ext, T + 64, 64))
        with self._Branch(self, 2):
            t = self.conv(p + '.branch3x3dbl_2', t3)
            self.conv(p + '.branch3x3dbl_3', t, _Act(ext, T + 128, 96))
        with self._Branch(self, 3):
            self.conv(p + '.branch_pool', self.avgpool(x), _Act(ext, T + 224, pf))
        return self._end_block(_Act(ext, T, 64 + 64 + 96 + pf))

    def _B(self, p, x):
        N, H, W, _ = x.shape
        OH = (H - 3) // 2 + 1
        cat = self._new(N, OH, OH, 384 + 96 + x.C)
        self._begin_block(x)
        with self._Branch(self, 0):
            self.conv(p + '.branch3x3', x, _Act(cat, 0, 384))
        with self._Branch(self, 1):
            t = self.conv(p + '.branch3x3dbl_2', self.conv(p + '.branch3x3dbl_1', x))
            self.conv(p + '.branch3x3dbl_3', t, _Act(cat, 384, 96))
        with self._Branch(self, 2):
            self.maxpool(x, _Act(cat, 480, x.C))
        return self._end_block(_Act(cat))

    def _C(self, p, x):
        N, H, W, _ = x.shape
        names = [p + '.branch7x7_1', p + '.branch7x7dbl_1', p + '.branch1x1']
        T = sum(self._convs[n].Op for n in names[:-1])
        ext = self._new(N, H, W, T + 768)
        self._begin_block(x)
        t7, t7d, _b1 = self._heads(names, x, ext)
        with self._Branch(self, 1):
            t = self.conv(p + '.branch7x7_2', t7)
            self.conv(p + '.branch7x7_3', t, _Act(ext, T + 192, 192))
        with self._Branch(self, 2):
            t = t7d
            for k in (2, 3, 4):
                t = self.conv(p + '.branch7x7dbl_%d' % k, t)
            self.conv(p + '.branch7x7dbl_5', t, _Act(ext, T + 384, 192))
        with self._Branch(self, 3):
            self.conv(p + '.branch_pool', self.avgpool(x), _Act(ext, T + 576, 192))
        return self._end_block(_Act(ext, T, 768))

    def _D(self, p, x):
        N, H, W, _ = x.shape
        OH = (H - 3) // 2 + 1
        cat = self._new(N, OH, OH, 320 + 192 + x.C)
        names = [p + '.branch3x3_1', p + '.branch7x7x3_1']
        tmp = self._new(N, H, W, sum(self._convs[n].Op for n in names))
        self._begin_block(x)
        t3, t7 = self._heads(names, x, tmp)
        with self._Branch(self, 0):
            self.conv(p + '.branch3x3_2', t3, _Act(cat, 0, 320))
        with self._Branch(self, 1):
            t = t7
            for k in (2, 3):
                t = self.conv(p + '.branch7x7x3_%d' % k, t)
            self.conv(p + '.branch7x7x3_4', t, _Act(cat, 320, 192))
        with self._Branch(self, 2):
            self.maxpool(x, _Act(cat, 512, x.C))
        return self._end_block(_Act(cat))

    def _E(self, p, x, dense_out=False):
        """dense_out: the concat is its own tensor (the global average pool after Mixed_7c reads it densely),
        so only the two temp-producing heads are fused"""
        N, H, W, _ = x.shape
        if dense_out:
            names = [p + '.branch3x3_1', p + '.branch3x3dbl_1']
            ext = self._new(N, H, W, 2048)
            tmp = self._new(N, H, W, sum(self._convs[n].Op for n in names))
            T = 0
        else:
            names = [p + '.branch3x3_1', p + '.branch3x3dbl_1', p + '.branch1x1']
            T = sum(self._convs[n].Op for n in names[:-1])
            ext = tmp = self._new(N, H, W, T + 2048)
        self._begin_block(x)
        views = self._heads(names, x, tmp)
        t3, t3d = views[0], views[1]
        if dense_out:
            with self._Branch(self, 0):
                self.conv(p + '.branch1x1', x, _Act(ext, 0, 320))
        with self._Branch(self, 1):
            self.conv(p + '.branch3x3_2a', t3, _Act(ext, T + 320, 384))
            self.conv(p + '.branch3x3_2b', t3, _Act(ext, T + 704, 384))
        with self._Branch(self, 2):
            t = self.conv(p + '.branch3x3dbl_2', t3d)
            self.conv(p + '.branch3x3dbl_3a', t, _Act(ext, T + 1088, 384))
            self.conv(p + '.branch3x3dbl_3b', t, _Act(ext, T + 1472, 384))
        with self._Branch(self, 3):
            self.conv(p + '.branch_pool', self.avgpool(x), _Act(ext, T + 1856, 192))
        return self._end_block(_Act(ext, T, 2048))

    # ------------------------------------------------------------------ forward / backward
    def forward(self, img):
        if not self._train and getattr(self, '_stale', False):
            self.refold()
            self._stale = False
        img = img.float().contiguous()
        N, _, S, _ = img.shape
        dt = self._dt()
        self.tape, self._grads = [], {}
        self._readers, self._relu_slices, self._masked, self._filled = {}, set(), set(), set()
        self.named = {}
        st = ops._stream()
        # the 299 x 299 resize (model.py:210) is never written: the stem conv interpolates on the fly
        a0 = _Act(self._new(N, 149, 149, 32))
        if self._train:
            stem = self.enc.Conv2d_1a_3x3
            wraw = stem.conv.weight.detach().float().contiguous(memory_format=torch.channels_last)
            y0 = torch.empty((N, 149, 149, 32), dtype=self.dtype, device=self.device)
            call('sba_enc_stem_resize_fwd', dt, img.data_ptr(), wraw.data_ptr(), None, y0.data_ptr(), N, S, 299, 32, st)
            self._bn_relu_train(y0, stem.bn, N * 149 * 149, 32, a0.t, 32, 0)
        else:
            call('sba_enc_stem_resize_fwd', dt, img.data_ptr(), self.stem_w.data_ptr(), self.stem_b.data_ptr(),
                 a0.t.data_ptr(), N, S, 299, 32, st)
        nm = self.named
        nm['Conv2d_1a_3x3'] = a0
        a = nm['Conv2d_2a_3x3'] = self.conv('Conv2d_2a_3x3', a0)
        a = nm['Conv2d_2b_3x3'] = self.conv('Conv2d_2b_3x3', a)
        a = nm['pool1'] = self.maxpool(a)
        a = nm['Conv2d_3b_1x1'] = self.conv('Conv2d_3b_1x1', a)
        a = nm['Conv2d_4a_3x3'] = self.conv('Conv2d_4a_3x3', a)
        a = nm['pool2'] = self.maxpool(a)
        a = nm['Mixed_5b'] = self._A('Mixed_5b', a, 32)
        a = nm['Mixed_5c'] = self._A('Mixed_5c', a, 64)
        a = nm['Mixed_5d'] = self._A('Mixed_5d', a, 64)
        a = nm['Mixed_6a'] = self._B('Mixed_6a', a)
        for nme in ('Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e'):
            a = nm[nme] = self._C(nme, a)
        feat_in = a
        f = self.conv('emb_features', feat_in)                       # [N,17,17,nef_p]
        a = nm['Mixed_7a'] = self._D('Mixed_7a', a)
        a = nm['Mixed_7b'] = self._E('Mixed_7b', a)
        a = nm['Mixed_7c'] = self._E('Mixed_7c', a, dense_out=True)
        pooled = torch.empty((N, 2048), dtype=torch.float32, device=self.device)
        call('sba_global_avgpool', dt, a.t.data_ptr(), pooled.data_ptr(), N, 64, 2048, 0, st)
        pooled_t = _Act(pooled.to(self.dtype).view(N, 1, 1, 2048))
        code = self.conv('emb_cnn_code', pooled_t)                    # [N,1,1,nef_p]
        nef = self.nef
        features = torch.empty((N, f.C, 17, 17), dtype=torch.float32, device=self.device)
        call('sba_layout_nhwc_nchw', dt, f.t.data_ptr(), features.data_ptr(), N, 289, f.C, 0, st)
        self._saved = (img.shape, None, a0, f, a, pooled_t, code)
        return features[:, :nef], code.t.view(N, -1)[:, :nef].float()

    def trunk_features(self, img, train=False):
        """The FROZEN part only, without a tape: (Mixed_6e output B x 768 x 17 x 17 as an NHWC f32 tensor
        [B, 17, 17, 768], pooled Mixed_7c output [B, 2048] f32) -- the inputs of the two trainable embedding layers
        (emb_features, emb_cnn_code) that the DAMSM pre-training loop updates (pretrain_DAMSM.py:62-75).

        train=True: the reference's pre-training loop runs the whole encoder in TRAINING mode (pretrain_DAMSM.py:51
        cnn_model.train()): the Inception weights are frozen (model.py:174-175) but every BatchNorm normalises with
        BATCH statistics and moves its running statistics (momentum 0.1) -- those updated buffers are what
        image_encoder*.pth carries into the GAN.  Each BasicConv2d then runs as conv (unfolded weights, statistics in
        the epilogue) + one BatchNorm/ReLU pass; train=False (evaluate(), pretrain_DAMSM.py:134) is the folded trunk."""
        with torch.no_grad():
            saved = (self._train, self.group, self.parallel)
            if train:
                self._train, self.group, self.parallel = True, False, False
            try:
                self.forward(img)
            finally:
                self._train, self.group, self.parallel = saved
            if train:       # the folded operands of the next eval-mode call must see the moved running statistics
                self._stale = True
            (_, _, _, _, last, pooled_t, _) = self._saved
            feat_in = self.named['Mixed_6e']
            f768 = feat_in.t[..., feat_in.coff:feat_in.coff + 768].float().contiguous()
            pooled = pooled_t.t.view(img.shape[0], -1)[:, :2048].float().contiguous()
        self.tape, self._grads, self._saved, self.named = [], {}, None, {}
        return f768, pooled

    def backward(self, dfeat, dcode):
        (ishape, x299, a0, f, last, pooled_t, code) = self._saved
        N = ishape[0]
        dt, st = self._dt(), ops._stream()
        # seed the two head gradients
        gf = self._grad_of(f)
        if dfeat is not None:
            if dfeat.shape[1] == f.C and dfeat.dtype == torch.float32 and dfeat.is_contiguous():
                df = dfeat                  # nef is a multiple of 32: no padded staging copy
            else:
                df = torch.zeros((N, f.C, 17, 17), dtype=torch.float32, device=self.device)
                df[:, :self.nef] = dfeat
            call('sba_layout_nhwc_nchw', dt, gf[0].data_ptr(), df.data_ptr(), N, 289, f.C, 1, st)
        else:
            gf[0].zero_()
        self._filled.add(self._key(f))
        gc = self._grad_of(code)
        if dcode is not None and gc[0].numel() == dcode.numel() and dcode.dtype == torch.float32 and \
                dcode.is_contiguous():
            call('sba_cast', dt, gc[0].data_ptr(), _lib.SBA_F32, dcode.data_ptr(), dcode.numel(), st)
        else:
            gc[0].zero_()
            if dcode is not None:
                gc[0].view(N, -1)[:, :self.nef] = dcode.to(self.dtype)
        self._filled.add(self._key(code))
        def run(op):
            kind, L, x, out = op
            if kind == 'conv':
                self._conv_bwd(L, x, out)
                if x is pooled_t:
                    gp = self._grad_of(pooled_t)[0].view(N, 2048).float().contiguous()
                    gl = self._grad_of(last)
                    call('sba_global_avgpool', dt, gl[0].data_ptr(), gp.data_ptr(), N, 64, 2048, 1, st)
                    self._filled.add(self._key(last))
            elif kind == 'maxpool':
                self._maxpool_bwd(x, out, L)
            else:
                self._avgpool_bwd(x, out)

        for entry in reversed(self.tape):
            if entry[0] != 'block':
                run(entry)
                continue
            # a block: every branch back-propagates on its own stream; ops that read the block INPUT (they
            # accumulate into its shared gradient) are held back and run in order on the main stream after
            # the join -- pools first, convs next, the fused head conv last: the last accumulation is then
            # a conv data-gradient, whose epilogue applies the input's ReLU mask
            main = torch.cuda.current_stream()
            _, branches, bx, pre = entry
            held = []
            used = []
            if self.group:
                chains = []
                for k, ops_k in branches:
                    rest = list(ops_k)
                    if rest and rest[0][2].t is bx.t:
                        held.append(rest.pop(0))
                    if rest:
                        chains.append(list(reversed(rest)))
                for lvl in range(max([len(c) for c in chains] or [0])):
                    self._pending = []
                    for c in chains:
                        if lvl < len(c):
                            run(c[lvl])
                    self._flush()
                for op in sorted(held, key=lambda o: 0 if o[0] != 'conv' else 1) + list(reversed(pre)):
                    run(op)
                continue
            for k, ops_k in branches:
                rest = list(ops_k)
                if rest and rest[0][2].t is bx.t:
                    held.append(rest.pop(0))
                if not rest:
                    continue
                used.append(k)
                if self.parallel:
                    sk = self._streams()[k]
                    sk.wait_stream(main)
                    with torch.cuda.stream(sk):
                        for op in reversed(rest):
                            run(op)
                else:
                    for op in reversed(rest):
                        run(op)
            if self.parallel:
                for k in used:
                    main.wait_stream(self._streams()[k])
            for op in sorted(held, key=lambda o: 0 if o[0] != 'conv' else 1) + list(reversed(pre)):
                run(op)
        g0 = self._grad_of(a0)
        d299 = torch.empty((N, 3, 299, 299), dtype=torch.float32, device=self.device)
        call('sba_enc_stem_bwd', dt, self.stem_w.data_ptr(), a0.t.data_ptr(), g0[0].data_ptr(), d299.data_ptr(), N,
             299, 32, st)
        dimg = torch.empty(ishape, dtype=torch.float32, device=self.device)
        call('sba_resize_bilinear', d299.data_ptr(), dimg.data_ptr(), N * 3, ishape[2], 299, 1, st)
        if not getattr(self, 'keep_debug', False):
            self.tape, self._grads, self._saved, self.named = [], {}, None, {}
        self.d299 = d299 if getattr(self, 'keep_debug', False) else None
        return dimg

    def __call__(self, img):
        return _InceptionFn.apply(img, self)


class _InceptionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, runner):
        ctx.runner = runner
        runner._want_grad = bool(ctx.needs_input_grad[0])      # a backward will follow: keep the max-pool argmax
        feats, code = runner.forward(img)
        return feats.contiguous(), code.contiguous()

    @staticmethod
    def backward(ctx, dfeat, dcode):
        return ctx.runner.backward(dfeat, dcode), None
