"""Host-side operators over the C-ABI HIP library: tensor plumbing (torch owns the
device memory and the stream), convolution geometry, packed-weight caching, and the
torch.autograd.Function wrappers the module classes in model.py are built from.

Activations are logical NCHW tensors with channels_last strides, i.e. NHWC in
memory, of dtype COMPUTE_DTYPE (bfloat16 by default, float32 for the exact-f32
MFMA path).  Parameters, their gradients, statistics and losses are float32.

Parameter gradients are ACCUMULATED IN PLACE into `param.grad` by the kernels
(the reference only ever uses `loss.backward(); optimizer.step()`,
trainer.py:269-297); the autograd Functions therefore return None for
parameters.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_GLU, ACT_LRELU, ACT_NONE, ConvGeom, call

CL = torch.channels_last
COMPUTE_DTYPE = torch.bfloat16
BN_EPS, BN_MOMENTUM, IN_EPS = 1e-5, 0.1, 1e-5

_WEIGHT_EPOCH = [0]          # bumped whenever parameters are changed behind torch's back


def set_compute_dtype(dt):
    global COMPUTE_DTYPE
    assert dt in (torch.bfloat16, torch.float32)
    COMPUTE_DTYPE = dt
    _WEIGHT_EPOCH[0] += 1


def compute_dtype():
    return COMPUTE_DTYPE


def weights_changed():
    """Call after parameters were modified through raw pointers (fused Adam)."""
    _WEIGHT_EPOCH[0] += 1


_EPOCH_CELL = {}             # id(param) -> [counter] shared by all parameters of one network


def register_epoch(params, cell):
    """Give a group of parameters (one network) its own change counter, so that an optimizer step
    on one network does not invalidate the packed weights of the others."""
    for p in params:
        _EPOCH_CELL[id(p)] = cell


def _dt(t):
    if t.dtype == torch.float32:
        return _lib.SBA_F32
    if t.dtype == torch.bfloat16:
        return _lib.SBA_BF16
    if t.dtype == torch.float16:        # a raw conv output y in the bf16 path (Y_F16 below)
        return _lib.SBA_BF16_YH
    raise TypeError('unsupported activation dtype %s' % t.dtype)


# bf16 path, optional (SBA_Y_F16=1): the RAW conv outputs (the pre-BatchNorm tensors y, read only by the BatchNorm
# kernels, never an MFMA operand) stored as IEEE binary16 instead of bf16 -- the same bytes, three more mantissa bits:
# one of the two roundings per conv + BatchNorm + activation layer shrinks by 8x (include/sbagan_hip.h: SBA_BF16_YH;
# the epilogue saturates at +-65504).  Measured on the reference's B = 20 golden step in deterministic mode
# (profiles/r03_bf16_rounding_points.txt): it moves the bf16 step's deviation from the reference by +-3e-4 in BOTH
# directions (errD0 7.7e-4 -> 7.1e-4, errD2 9.3e-4 -> 1.2e-3): that deviation is a sum of ~30 roundings of ~1e-4 with
# random signs (tools/bf16_bisect.py), no single tensor dominates it -- so the default stays plain bf16.
Y_F16 = os.environ.get('SBA_Y_F16', '0') == '1'


def _act_dtype(y):
    """storage dtype of the activations around a raw conv output y"""
    return torch.bfloat16 if y.dtype == torch.float16 else y.dtype


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('sbagan HIP operators need CUDA/HIP tensors (got a %s tensor); there is no CPU '
                           'fallback' % t.device)


def as_act(x, dtype=None):
    """logical NCHW, NHWC in memory, compute dtype."""
    dtype = dtype or COMPUTE_DTYPE
    if x.dtype != dtype:
        x = x.to(dtype)
    if x.dim() == 4 and not x.is_contiguous(memory_format=CL):
        x = x.contiguous(memory_format=CL)
    return x


def empty_act(n, c, h, w, like):
    return torch.empty((n, c, h, w), dtype=like.dtype, device=like.device, memory_format=CL)


def param_grad(p):
    """f32 gradient buffer of a parameter with the parameter's own memory layout."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
    return p.grad


# ----------------------------------------------------------------------------
# convolution geometry
# ----------------------------------------------------------------------------
_GEOM_CACHE = {}


def _geom(key):
    g = _GEOM_CACHE.get(key)
    if g is not None:
        return g
    kind, N, IH, IW, Cin, Cout, extra = key
    g = ConvGeom()
    g.N, g.IH, g.IW, g.Cin, g.Cout = N, IH, IW, Cin, Cout
    g.sy = g.sx = 1
    g.osy = g.osx = 1
    g.ooy = g.oox = 0
    g.ups = 0
    if kind in ('3x3', '3x3up'):
        up = 2 if kind == '3x3up' else 1
        g.OH, g.OW = IH * up, IW * up
        g.OHs, g.OWs = g.OH, g.OW
        g.ups = 1 if kind == '3x3up' else 0
        g.ntaps = 9
        for t in range(9):
            g.ty[t], g.tx[t] = t // 3 - 1, t % 3 - 1
    elif kind == '4x4s2':
        g.OH, g.OW = IH // 2, IW // 2
        g.OHs, g.OWs = g.OH, g.OW
        g.sy = g.sx = 2
        g.ntaps = 16
        for t in range(16):
            g.ty[t], g.tx[t] = t // 4 - 1, t % 4 - 1
    elif kind == '4x4s2_dgrad':
        # input = dY (IH x IW), output = dX (2IH x 2IW), parity class extra = (py, px)
        py, px = extra
        g.OH, g.OW = IH * 2, IW * 2
        g.OHs, g.OWs = IH, IW
        g.osy = g.osx = 2
        g.ooy, g.oox = py, px
        g.ntaps = 4
        for t in range(4):
            g.ty[t], g.tx[t] = py - t // 2, px - t % 2
    else:
        raise ValueError(kind)
    _GEOM_CACHE[key] = g
    return g


def _conv_out_hw(kind, H, W):
    if kind == '3x3':
        return H, W
    if kind == '3x3up':
        return 2 * H, 2 * W
    if kind == '4x4s2':
        return H // 2, W // 2
    raise ValueError(kind)


# data-gradient operand layouts (sba_pack_weight modes): flipped 3x3, four parity classes of the 4x4/s2
# conv, and the 4x4/s2 collapse of (nearest x2 -> conv3x3)
_DGRAD_MODE = {'3x3': 1, '4x4s2': 2, '3x3up': 3}


class PackedWeight(object):
    """Packed copies of one OIHW conv parameter (stored channels_last, i.e.
    [O][KH][KW][I] f32 in memory): the forward operand in the compute dtype and
    the data-gradient operand, rebuilt lazily when the parameter changes.  Inside a PackGroup
    (one per trained network) a stale copy triggers ONE launch that refreshes the whole network."""

    def __init__(self, param, kind=None):
        self.param = param
        self.kind = kind
        self.group = None
        self._fwd = self._dgrad = None
        self._kf = self._kd = None
        self._ffwd = self._fdgrad = None        # fragment-major copies (halo-tile 3x3 kernel: weights in registers)
        self._kff = self._kfd = None

    def _key(self, dtype):
        cell = _EPOCH_CELL.get(id(self.param))
        return (self.param._version, _WEIGHT_EPOCH[0], cell[0] if cell else 0, dtype, self.param.data_ptr())

    def _master(self):
        p = self.param.detach()
        if not p.is_contiguous(memory_format=CL):
            raise RuntimeError('conv parameters must be stored channels_last (use sbagan layers)')
        return p

    def fwd(self, dtype):
        if dtype == torch.float32:
            return self._master()
        k = self._key(dtype)
        if self._kf != k:
            if self.group is not None:
                self.group.refresh(dtype)
                return self._fwd
            p = self._master()
            O, I, KH, KW = p.shape
            if self._fwd is None or self._fwd.dtype != dtype:
                self._fwd = torch.empty(O * KH * KW * I, dtype=dtype, device=p.device)
            call('sba_pack_weight', _lib.SBA_BF16, _p(p), _p(self._fwd), O, KH, KW, I, 0, _stream())
            self._kf = k
        return self._fwd

    def fwd_frag(self, dtype):
        """the forward operand FRAGMENT-MAJOR (include/sbagan_hip.h: sba_pack_frag_multi), or None when the layer does not
        qualify (bf16, Cin 64 / 128, Cout % 64 == 0, 3 x 3)"""
        O, I, KH, KW = self.param.shape
        if not FRAG_WEIGHTS or dtype != torch.bfloat16 or KH != 3 or KW != 3 or I not in (64, 128) or O % 64:
            return None
        src = self.fwd(dtype)                   # (refreshes the group, frag copies included, when stale)
        k = self._key(dtype)
        if self._kff != k:
            if self._ffwd is None:
                self._ffwd = torch.empty(O * 9 * I, dtype=dtype, device=src.device)
            _pack_frag([(src, self._ffwd, O, 9, I)], src.device)
            self._kff = k
        return self._ffwd

    def dgrad_frag(self, dtype, kind):
        """the data-gradient operand of a stride-1 3 x 3 conv ([Cin][flipped tap][Cout]) fragment-major, or None"""
        O, I, KH, KW = self.param.shape
        if not FRAG_WEIGHTS or dtype != torch.bfloat16 or kind != '3x3' or O not in (64, 128) or I % 64:
            return None
        src = self.dgrad(dtype, kind)
        k = self._key(dtype) + (kind,)
        if self._kfd != k:
            if self._fdgrad is None:
                self._fdgrad = torch.empty(O * 9 * I, dtype=dtype, device=src.device)
            _pack_frag([(src, self._fdgrad, I, 9, O)], src.device)
            self._kfd = k
        return self._fdgrad

    def dgrad(self, dtype, kind):
        k = self._key(dtype) + (kind,)
        if self._kd != k:
            if self.group is not None and kind == self.kind:
                self.group.refresh(dtype)
                return self._dgrad
            p = self._master()
            O, I, KH, KW = p.shape
            mode = _DGRAD_MODE[kind]
            n = O * I * (16 if mode == 3 else KH * KW)
            if self._dgrad is None or self._dgrad.dtype != dtype or self._dgrad.numel() != n:
                self._dgrad = torch.empty(n, dtype=dtype, device=p.device)
            dcode = _lib.SBA_BF16 if dtype == torch.bfloat16 else _lib.SBA_F32
            call('sba_pack_weight', dcode, _p(p), _p(self._dgrad), O, KH, KW, I, mode, _stream())
            self._kd = k
        return self._dgrad


FRAG_WEIGHTS = os.environ.get('SBA_FRAG_WEIGHTS', '1') != '0'
_FRAG_DESC = [('src', '<u8'), ('dst', '<u8'), ('R', '<i4'), ('taps', '<i4'), ('K', '<i4'), ('unit_begin', '<i4')]


def _frag_descs(items, device):
    """device array of sba_frag_desc for items = [(src tensor, dst tensor, R, taps, K)] and the total unit count"""
    import numpy as np
    desc = np.zeros(len(items), dtype=np.dtype(_FRAG_DESC, align=True))
    assert desc.dtype.itemsize == 32
    units = 0
    for d, (src, dst, R, taps, K) in zip(desc, items):
        d['src'], d['dst'], d['R'], d['taps'], d['K'], d['unit_begin'] = src.data_ptr(), dst.data_ptr(), R, taps, K, units
        units += R * taps * K // 8
    return torch.from_numpy(desc.view(np.uint8).copy()).to(device), units


def _pack_frag(items, device):
    descs, units = _frag_descs(items, device)
    call('sba_pack_frag_multi', descs.data_ptr(), len(items), units, _stream())
    _KEEP_DESC.append(descs)
    del _KEEP_DESC[:-64]


_KEEP_DESC = []         # descriptor arrays of the un-grouped path stay alive until their launch has surely run


class PackGroup(object):
    """The packed weights of all conv layers of one network, kept in two flat buffers (forward
    operands, data-gradient operands) and refreshed by ONE sba_pack_weights_multi launch when any
    of them is stale (normally: once after the network's optimizer step) instead of two launches
    per layer.  `layers` = [(PackedWeight, kind)] with kind in '3x3' | '3x3up' | '4x4s2'."""

    _DESC = [('w', '<u8'), ('fwd', '<u8'), ('tr', '<u8'), ('Cout', '<i4'), ('KH', '<i4'), ('KW', '<i4'),
             ('Cin', '<i4'), ('mode', '<i4'), ('tile_begin', '<i4'), ('co_tiles', '<i4'), ('ci_tiles', '<i4')]

    def __init__(self, layers):
        self.layers = [(pw, kind) for pw, kind in layers if pw.param.shape[1] % 4 == 0]
        self.state = {}             # dtype -> dict(fwd, tr, descs, ptrs, tiles)
        for pw, kind in self.layers:
            pw.group, pw.kind = self, kind

    def _build(self, dtype):
        import numpy as np
        dev = self.layers[0][0].param.device
        sizes = [pw.param.numel() for pw, _ in self.layers]
        tsizes = [pw.param.shape[0] * pw.param.shape[1] * (16 if kind == '3x3up' else pw.param.shape[2] * pw.param.shape[3])
                  for pw, kind in self.layers]
        offs, n = [], 0
        for k in sizes:
            offs.append(n)
            n += (k + 7) // 8 * 8
        toffs, tn = [], 0
        for k in tsizes:
            toffs.append(tn)
            tn += (k + 7) // 8 * 8
        st = {'fwd': torch.empty(n, dtype=dtype, device=dev) if dtype != torch.float32 else None,
              'tr': torch.empty(tn, dtype=dtype, device=dev)}
        esz = st['tr'].element_size()
        desc = np.zeros(len(self.layers), dtype=np.dtype(self._DESC, align=True))
        assert desc.dtype.itemsize == 56
        tiles = 0
        for i, ((pw, kind), o) in enumerate(zip(self.layers, offs)):
            O, I, KH, KW = pw.param.shape
            d = desc[i]
            d['w'] = pw.param.data_ptr()
            d['fwd'] = st['fwd'].data_ptr() + o * esz if st['fwd'] is not None else 0
            d['tr'] = st['tr'].data_ptr() + toffs[i] * esz
            d['Cout'], d['KH'], d['KW'], d['Cin'] = O, KH, KW, I
            d['mode'] = _DGRAD_MODE[kind]
            d['tile_begin'] = tiles
            d['co_tiles'], d['ci_tiles'] = (O + 63) // 64, (I + 63) // 64
            tiles += (16 if d['mode'] == 3 else KH * KW) * d['co_tiles'] * d['ci_tiles']
            if st['fwd'] is not None:
                pw._fwd = st['fwd'][o:o + sizes[i]]
            pw._dgrad = st['tr'][toffs[i]:toffs[i] + tsizes[i]]
        st['descs'] = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        st['ptrs'] = [pw.param.data_ptr() for pw, _ in self.layers]
        st['tiles'] = tiles
        # fragment-major copies of the 3 x 3 layers the halo-tile kernel serves (ONE more launch per refresh)
        st['frag'] = None
        if FRAG_WEIGHTS and dtype == torch.bfloat16:
            items = []
            for pw, kind in self.layers:
                O, I, KH, KW = pw.param.shape
                if kind != '3x3' or KH != 3:
                    continue
                if I in (64, 128) and O % 64 == 0:
                    pw._ffwd = torch.empty(O * 9 * I, dtype=dtype, device=dev)
                    items.append((pw._fwd, pw._ffwd, O, 9, I))
                if kind == '3x3' and O in (64, 128) and I % 64 == 0:
                    pw._fdgrad = torch.empty(O * 9 * I, dtype=dtype, device=dev)
                    items.append((pw._dgrad, pw._fdgrad, I, 9, O))
            if items:
                st['frag'] = _frag_descs(items, dev) + (len(items),)
        self.state = {dtype: st}        # one compute dtype at a time (the views above belong to it)
        return st

    def refresh(self, dtype):
        st = self.state.get(dtype)
        if st is None or st['ptrs'] != [pw.param.data_ptr() for pw, _ in self.layers]:
            st = self._build(dtype)
        for pw, _ in self.layers:
            if not pw.param.is_contiguous(memory_format=CL):
                raise RuntimeError('conv parameters must be stored channels_last (use sbagan layers)')
        dcode = _lib.SBA_BF16 if dtype == torch.bfloat16 else _lib.SBA_F32
        call('sba_pack_weights_multi', dcode, st['descs'].data_ptr(), len(self.layers), st['tiles'], _stream())
        if st.get('frag') is not None:
            fd, units, nfrag = st['frag']
            call('sba_pack_frag_multi', fd.data_ptr(), nfrag, units, _stream())
        for pw, kind in self.layers:
            k = pw._key(dtype)
            pw._kf, pw._kd = k, k + (kind,)
            pw._kff, pw._kfd = k, k + (kind,)


class ZeroArena(object):
    """Bump allocator over one pre-zeroed f32 buffer for the many small accumulators of a step
    (BN statistics, backward reductions): ONE memset per step instead of one fill kernel per layer.
    Only active between begin() and end() (GANStep.step); otherwise zeros_f32 falls back to
    torch.zeros, so stand-alone module calls stay safe."""

    def __init__(self):
        self.buf, self.off, self.active = None, 0, False

    def begin(self, device, nfloats=4 << 20):
        if self.buf is None or self.buf.device != device or self.buf.numel() < nfloats:
            self.buf = torch.empty(nfloats, dtype=torch.float32, device=device)
        self.buf.zero_()
        self.off, self.active = 0, True

    def end(self):
        self.active = False

    def take(self, n, device):
        n4 = (n + 3) // 4 * 4
        if not self.active or self.buf.device != device or self.off + n4 > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n]
        self.off += n4
        return t


ARENA = ZeroArena()


def zeros_f32(shape, device):
    n = 1
    for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
        n *= d
    t = ARENA.take(n, device)
    if t is None:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    return t.view(shape)


_WORKSPACE = {}
WORKSPACE_BYTES = 64 << 20


def workspace(device):
    """Per-(device, stream) scratch for split-K partial sums: launches on one stream reuse it in
    stream order; concurrent streams must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACE.get(key)
    if ws is None:
        with torch.cuda.stream(torch.cuda.default_stream(device)):
            ws = torch.zeros(WORKSPACE_BYTES, dtype=torch.uint8, device=device)   # contract: zero-filled
        torch.cuda.current_stream(device).wait_stream(torch.cuda.default_stream(device))
        _WORKSPACE[key] = ws
    return ws


_REDUCE_SCRATCH = {}
REDUCE_SCRATCH_BYTES = int(os.environ.get('SBA_REDUCE_SCRATCH_MB', '64')) << 20


def reduce_scratch(device):
    """The ring for the default mode's two-stage reductions (include/sbagan_hip.h: sba_set_reduce_scratch), handed to
    the library the first time an operator that uses it runs on `device` (SBA_REDUCE_SCRATCH_MB=0: none, atomics)."""
    if device not in _REDUCE_SCRATCH:
        if _REDUCE_SCRATCH and REDUCE_SCRATCH_BYTES:
            # the library keeps ONE ring (include/sbagan_hip.h): a second device would silently re-point it
            raise RuntimeError('sbagan runs one process per GPU: the reduction scratch ring already belongs to %s'
                               % next(iter(_REDUCE_SCRATCH)))
        buf = None
        if REDUCE_SCRATCH_BYTES:
            with torch.cuda.stream(torch.cuda.default_stream(device)):
                buf = torch.empty(REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=device)
            torch.cuda.current_stream(device).wait_stream(torch.cuda.default_stream(device))
            call('sba_set_reduce_scratch', buf.data_ptr(), REDUCE_SCRATCH_BYTES)
        _REDUCE_SCRATCH[device] = buf
    return _REDUCE_SCRATCH[device]


# ---- measured tile table (tools/tune_igemm.py -> sbagan/igemm_table.json) --------------------------------
# key -> [tile, ksplit]: the tile configuration / K split of sba_conv_igemm that was fastest for that layer shape
# on an MI355X (bf16).  Shapes that are not in the table use the library's rule table (tile = 0).
IGEMM_LOG = None            # set to a list to record the geometry of every implicit-GEMM launch (tuning aid)
_IGEMM_TABLE = None


def geom_key(g):
    return '%d_%dx%d_%d_%dx%d_%d_t%d_s%d_u%d' % (g.N, g.IH, g.IW, g.Cin, g.OHs, g.OWs, g.Cout, g.ntaps, g.sy, g.ups)


_TABLE_FILE = None


def _table_section(name):
    global _TABLE_FILE
    if _TABLE_FILE is None:
        _TABLE_FILE = {}
        path = os.environ.get('SBA_IGEMM_TABLE_FILE') or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                      'igemm_table.json')      # (the override: an A/B aid)
        if os.environ.get('SBA_IGEMM_TABLE', '1') != '0' and os.path.exists(path):
            import json
            with open(path) as f:
                _TABLE_FILE = json.load(f)
    return _TABLE_FILE.get(name, {})


def _igemm_table():
    global _IGEMM_TABLE
    if _IGEMM_TABLE is None:
        _IGEMM_TABLE = _table_section('bf16')
    return _IGEMM_TABLE


def tune_geom(g, dt):
    """Fill g.tile / g.ksplit from the measured table, once per geometry object."""
    if getattr(g, '_tuned', None) != dt:
        g._tuned = dt
        ent = _igemm_table().get(geom_key(g)) if dt == _lib.SBA_BF16 else None
        g.tile, g.ksplit = (int(ent[0]), int(ent[1])) if ent else (0, 0)
    if IGEMM_LOG is not None:
        IGEMM_LOG.append(g)


def _igemm(dt, x, w, y, addend, stats, g, device, w_frag=None):
    """w_frag: the fragment-major copy of `w` (or None): used when the geometry goes to the halo-tile 3 x 3 kernel"""
    ws = workspace(device)
    tune_geom(g, _lib.SBA_BF16 if dt == _lib.SBA_BF16_YH else dt)
    g.w_layout = 0
    if w_frag is not None and _halo_family(g):
        w, g.w_layout = w_frag.data_ptr(), 1
    try:
        call('sba_conv_igemm', dt, x, w, y, addend, stats, ctypes.byref(g), ws.data_ptr(), WORKSPACE_BYTES, _stream())
    finally:
        g.w_layout = 0


def _halo_family(g):
    """does sba_conv_igemm send this (bf16) geometry to the halo-tile 3 x 3 kernel?  (asked once per geometry object)"""
    h = getattr(g, '_halo', None)
    if h is None:
        plan = (ctypes.c_int * 3)()
        old = g.w_layout
        g.w_layout = 0
        call('sba_conv_igemm_plan', _lib.SBA_BF16, ctypes.byref(g), WORKSPACE_BYTES, plan)
        g.w_layout = old
        h = g._halo = plan[0] == 0
    return h


# (Tried at the end of round 4 and removed: the BatchNorm-backward sums of a ResBlock's first BatchNorm taken in the epilogue of
# the second conv's data gradient -- sba_conv_igemm_bnred, commit 6ee6d14.  Correct, but 0.08 ms SLOWER per step when used, and
# the extra state in the epilogue every implicit-GEMM kernel shares cost 0.13 ms even when unused: DESIGN.md §8,
# profiles/r04_ab_fuse_bn_red.txt.)


# ----------------------------------------------------------------------------
# raw (non-autograd) building blocks
# ----------------------------------------------------------------------------
def conv_forward(x, pw, kind, want_stats=True, addend=None, pre_bn=False):
    """y = conv(x) in NHWC; returns (y, stats) with stats = BN_STAT_SLOTS replicas of the per-channel
    (sum, sumsq) pair (add them up; see SBA_BN_STAT_SLOTS in sbagan_hip.h).
    pre_bn: y goes to a BatchNorm and nowhere else -- in the bf16 path it is then a float16 tensor (Y_F16)."""
    _need_gpu(x)
    N, Cin, H, W = x.shape
    O = pw.param.shape[0]
    OH, OW = _conv_out_hw(kind, H, W)
    yh = pre_bn and Y_F16 and x.dtype == torch.bfloat16 and addend is None and O % 8 == 0
    if yh:
        y = torch.empty((N, O, OH, OW), dtype=torch.float16, device=x.device, memory_format=CL)
    else:
        y = empty_act(N, O, OH, OW, x)
    stats = zeros_f32((BN_STAT_SLOTS, 2 * O), x.device) if want_stats else None
    g = _geom((kind, N, H, W, Cin, O, None))
    # (behind the nearest x2 upsample the register-weight kernel measured no faster than the LDS-weight one at four
    #  workgroups per CU -- G3.up 124.5 vs 124.9 us, it is bound by its 335 MB of output -- so '3x3up' keeps row-major weights)
    wf = pw.fwd_frag(x.dtype) if kind == '3x3' else None
    _igemm(_lib.SBA_BF16_YH if yh else _dt(x), _p(x), _p(pw.fwd(x.dtype)), _p(y), _p(addend), _p(stats), g, x.device,
           w_frag=wf)
    return y, stats


def conv_dgrad(dy, pw, kind, in_hw, addend=None):
    """dx = conv_transpose(dy); `addend` (same shape as dx) is added in the epilogue."""
    N, O, OH, OW = dy.shape
    I = pw.param.shape[1]
    H, W = in_hw
    wd = pw.dgrad(dy.dtype, kind)
    if kind == '3x3':
        g = _geom(('3x3', N, OH, OW, O, I, None))
        dx = empty_act(N, I, H, W, dy)
        _igemm(_dt(dy), _p(dy), _p(wd), _p(dx), _p(addend), None, g, dy.device, w_frag=pw.dgrad_frag(dy.dtype, kind))
        return dx
    if kind == '3x3up':
        # every source pixel collects a 4x4 window of dy with the tap sums packed by mode 3: one stride-2
        # conv at the SOURCE resolution instead of a conv at the upsampled resolution + 2x2 sum pooling
        g = _geom(('4x4s2', N, OH, OW, O, I, None))
        dx = empty_act(N, I, H, W, dy)
        _igemm(_dt(dy), _p(dy), _p(wd), _p(dx), _p(addend), None, g, dy.device)
        return dx
    if kind == '4x4s2':
        dx = empty_act(N, I, H, W, dy)
        esz = dy.element_size()
        gs = [_geom(('4x4s2_dgrad', N, OH, OW, O, I, (cls // 2, cls % 2))) for cls in range(4)]
        if DGRAD4_GROUP and dy.dtype == torch.bfloat16 and O % 64 == 0:
            # the four parity classes read the same dy and write disjoint pixels of dx: ONE grid (4x the workgroups of a
            # class, one launch floor, one tail) instead of four launches of 80..640 workgroups
            arr = (_lib.ConvGroupItem * 4)()
            for cls, (a, g) in enumerate(zip(arr, gs)):
                a.x, a.w, a.y, a.addend = _p(dy), wd.data_ptr() + cls * I * 4 * O * esz, _p(dx), _p(addend)
                a.bias = a.relu_mask = None
                a.g = ctypes.pointer(g)
            tile, split = dgrad4_plan(gs[0])
            ws = workspace(dy.device)
            call('sba_conv_igemm_group_splitk', _lib.SBA_BF16, 4, arr, tile, split, ws.data_ptr(), WORKSPACE_BYTES, _stream())
            if IGEMM_LOG is not None:
                IGEMM_LOG.append(('group', tile, gs))
            return dx
        for cls in range(4):
            wptr = wd.data_ptr() + cls * I * 4 * O * esz
            _igemm(_dt(dy), _p(dy), wptr, _p(dx), _p(addend), None, gs[cls], dy.device)
        return dx
    raise ValueError(kind)


DGRAD4_GROUP = os.environ.get('SBA_DGRAD4_GROUP', '1') != '0'
_DGRAD4_FORCE = os.environ.get('SBA_DGRAD4_PLAN')          # tuning aid: "tile,split"


def dgrad4_plan(g):
    """(tile id, K splits) of the grouped launch of the four parity classes of a 4x4/s2 data gradient: from the measured
    table (igemm_table.json, section 'dgrad4'; tools/tune_dgrad4.py) or by rule -- the biggest tile that still gives
    two workgroups per CU, then K splits up to that occupancy while a split keeps >= 8 slabs of 64 channels."""
    if _DGRAD4_FORCE:
        t, s = _DGRAD4_FORCE.split(',')
        return int(t), int(s)
    ent = _table_section('dgrad4').get(geom_key(g))
    if ent:
        return int(ent[0]), int(ent[1])
    M = g.N * g.OHs * g.OWs
    ny = (g.Cout + 63) // 64
    for tile, bm in ((5, 128), (3, 96), (1, 64)):
        tiles = 4 * ((M + bm - 1) // bm) * ny
        if tiles >= 512 or tile == 1:
            break
    ns64 = g.ntaps * (g.Cin // 64)
    split = max(1, min(512 // max(tiles, 1), ns64 // 8))
    return tile, split


_KSPLIT_TARGET = int(os.environ.get('SBA_WGRAD_WGS', '640'))
_KSPLIT_MIN_CHUNKS = int(os.environ.get('SBA_WGRAD_MIN_CHUNKS', '24'))


def _ksplit(tiles, M):
    chunks = (M + 63) // 64
    want = max(1, (_KSPLIT_TARGET + tiles - 1) // tiles)       # ~2.5 workgroups per CU ...
    # ... but every split adds a full copy of the tile's outputs to the f32 atomics: keep >= 24 pixel
    # chunks (1536 pixels) of MFMA work per workgroup behind each copy
    return max(1, min(want, chunks // _KSPLIT_MIN_CHUNKS if chunks >= 2 * _KSPLIT_MIN_CHUNKS else 1))


FIRST_WRITE = os.environ.get('SBA_WGRAD_FIRST_WRITE', '1') != '0'


def conv_wgrad(x, dy, param, kind):
    """param.grad[O][KH][KW][I] += dy^T (*) x."""
    N, Cin, H, W = x.shape
    O = dy.shape[1]
    g = _geom((kind, N, H, W, Cin, O, None))
    gbuf = param_grad(param)
    M = N * g.OHs * g.OWs
    tiles = ((O + 63) // 64) * ((Cin + 63) // 64) * g.ntaps
    # first weight gradient of this parameter since its flat gradient buffer was cleared (trainer.FlatParams.zero_grad
    # bumps the cell): the kernel may store instead of read-modify-write.  Parameters outside a FlatParams never
    # qualify (somebody else owns their .grad).
    cell = getattr(param, '_sba_gepoch', None)
    first = 0
    if FIRST_WRITE and cell is not None and getattr(param, '_sba_wepoch', None) != cell[0]:
        param._sba_wepoch = cell[0]
        first = 1
    g.first_write = first
    call('sba_conv_wgrad', _dt(x), _p(x), _p(dy), _p(gbuf), ctypes.byref(g), _ksplit(tiles, M), _stream())


# ---- weight gradients on a companion stream ---------------------------------------------
# dW = dy^T (*) x and dx = conv^T(dy) of one layer are independent; with SIDE_WGRAD enabled the
# weight-gradient launch goes to a companion of the current stream so that it overlaps the
# data-gradient chain (the critical path of a backward pass).  The trainer joins the companions
# before the optimizer reads the gradients; tensors handed to the companion are kept alive until
# that join (no record_stream: safe under hipGraph capture).
SIDE_WGRAD = False
HOME_STREAM = None       # raw handle of the stream whose companion join_wgrads() will join (set by the trainer at the start of
#                          a backward pass): the dense layers use the companion only from there -- their backward may be
#                          replayed on a side stream (the mapping network's), whose companion nobody would join
_COMPANION, _KEEPALIVE, _COMP_NEXT = {}, {}, {}
N_COMPANIONS = int(os.environ.get('SBA_WGRAD_COMPANIONS', '1'))      # companion streams per stream, used in turn


def _companion(device, *keep):
    """the companion of the current stream, ordered behind everything issued to the current stream so far; `keep` =
    tensors the companion's launches read (held until the join)"""
    cur = torch.cuda.current_stream()
    key = cur.cuda_stream
    comps = _COMPANION.get(key)
    if comps is None:
        comps = _COMPANION[key] = [torch.cuda.Stream(device=device) for _ in range(N_COMPANIONS)]
        _KEEPALIVE[key] = []
        _COMP_NEXT[key] = 0
    comp = comps[_COMP_NEXT[key] % len(comps)]
    _COMP_NEXT[key] += 1
    comp.wait_stream(cur)
    _KEEPALIVE[key].extend(keep)
    return comp


def conv_wgrad_overlapped(x, dy, param, kind):
    if not SIDE_WGRAD:
        return conv_wgrad(x, dy, param, kind)
    param_grad(param)                     # allocate (if needed) on the caller's stream
    with torch.cuda.stream(_companion(x.device, x, dy)):
        conv_wgrad(x, dy, param, kind)


def join_wgrads():
    """Make the current stream wait for the weight gradients issued from it."""
    cur = torch.cuda.current_stream()
    comps = _COMPANION.get(cur.cuda_stream)
    if comps is not None:
        for comp in comps:
            cur.wait_stream(comp)
        _KEEPALIVE[cur.cuda_stream].clear()
        _COMP_NEXT[cur.cuda_stream] = 0


def wgrad_tail_stream():
    """One-way variant of join_wgrads for a stream that is itself a fork: returns the stream on
    which the rest of the update (all-reduce, Adam) must be issued -- the companion, ordered
    after everything the current stream has done -- instead of joining the companion back.
    hipStreamEndCapture (ROCm 7.2) crashes on a fork -> sub-fork -> join-into-the-fork round
    trip, while one-way edges that only re-join the capture's origin stream are fine
    (tools/debug_nested.py)."""
    cur = torch.cuda.current_stream()
    comps = _COMPANION.get(cur.cuda_stream)
    if comps is None:
        return cur
    comp = comps[0]
    comp.wait_stream(cur)
    for other in comps[1:]:
        comp.wait_stream(other)
    _KEEPALIVE[cur.cuda_stream].clear()
    _COMP_NEXT[cur.cuda_stream] = 0
    return comp


BN_STAT_SLOTS = _lib.lib.sba_bn_stat_slots()      # the replica count the library was COMPILED with (its kernels index
#                                                   the statistics buffers with it; include/sbagan_hip.h)


# ---- deterministic-reduction mode (include/sbagan_hip.h: sba_set_deterministic) ------------------------------
_DET_SCRATCH = [None]
DET_SCRATCH_BYTES = int(os.environ.get('SBA_DET_SCRATCH_MB', '2048')) << 20


def set_deterministic(flag, device=None):
    """Turn the library's deterministic-reduction mode on / off: with it on, two runs of the same launches on the
    same inputs are bit-identical in every launch mode (eager, hipGraph, native replayer).  Allocates the scratch
    ring the ordered reductions use (SBA_DET_SCRATCH_MB, default 2 GiB).  Call while the device is idle."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if flag:
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        if _DET_SCRATCH[0] is None or _DET_SCRATCH[0].device != dev:
            _DET_SCRATCH[0] = torch.empty(DET_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
        call('sba_set_deterministic', 1, _DET_SCRATCH[0].data_ptr(), DET_SCRATCH_BYTES)
    else:
        call('sba_set_deterministic', 0, None, 0)


def deterministic():
    return bool(_lib.lib.sba_get_deterministic())


def det_reset():
    """Rewind the scratch ring of the deterministic mode (start of a step / before a capture)."""
    if _lib.lib.sba_get_deterministic():
        call('sba_det_reset')


if os.environ.get('SBA_DETERMINISTIC', '0') == '1' and torch.cuda.is_available():
    set_deterministic(True)


class BNState(object):
    """per-forward BatchNorm quantities: aux[g] = (scale, shift, mean, rstd) of group g."""
    __slots__ = ('aux', 'C', 'rows', 'groups')


def bn_act_forward(y, stats, bn, act, residual=None, groups=1):
    """BatchNorm (train: batch statistics `stats` = per-group (sum, sumsq); eval: running stats) +
    activation (+ residual) in ONE launch for all `groups` BatchNorm batches of y; returns (out, state)."""
    N, C, H, W = y.shape
    Co = C // 2 if act == ACT_GLU else C
    st = BNState()
    st.C, st.rows, st.groups = C, (N // groups) * H * W, groups
    st.aux = torch.empty((groups, 4, C), dtype=torch.float32, device=y.device)
    out = torch.empty((N, Co, H, W), dtype=_act_dtype(y), device=y.device, memory_format=CL)
    call('sba_bn_act_fwd', _dt(y), _p(y), _p(stats), _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
         _p(bn.running_var), _p(bn.num_batches_tracked), _p(st.aux), _p(residual), _p(out), st.rows, groups, C,
         act, Co, 0, BN_EPS, BN_MOMENTUM, 1 if bn.training else 0, _stream())
    return out, st


BN_FUSED_BWD_ROWS = int(os.environ.get('SBA_BN_FUSED_ROWS', '2560'))       # rows per group up to which the one-launch backward is used


def bn_act_backward(y, dout, st, bn, act, need_param_grad=True, out=None, red=None):
    """red: the two backward sums when the caller has them already"""
    N, C, H, W = y.shape
    Co = C // 2 if act == ACT_GLU else C
    dy = out if out is not None else torch.empty(y.shape, dtype=_act_dtype(y), device=y.device, memory_format=CL)
    dg = db = None
    if need_param_grad:
        dg, db = param_grad(bn.weight), param_grad(bn.bias)
    if st.rows <= BN_FUSED_BWD_ROWS:
        call('sba_bn_act_bwd_fused', _dt(y), _p(y), _p(dout), _p(st.aux), _p(dy), _p(dg), _p(db), st.rows, st.groups,
             C, act, Co, 0, _stream())
        return dy
    if red is None:
        red = zeros_f32((st.groups, BN_STAT_SLOTS, 2 * C), y.device)
        call('sba_bn_act_bwd_reduce', _dt(y), _p(y), _p(dout), _p(st.aux), _p(red), st.rows, st.groups, C, act, Co, 0,
             _stream())
    call('sba_bn_act_bwd_apply', _dt(y), _p(y), _p(dout), _p(st.aux), _p(red), _p(dy), _p(dg), _p(db), st.rows,
         st.groups, C, act, Co, 0, _stream())
    return dy


def bn_act_forward_fused(y, bn, act, groups):
    """training-mode BatchNorm + activation of a small grouped map: statistics, finalize and normalise in one launch"""
    N, C, H, W = y.shape
    Co = C // 2 if act == ACT_GLU else C
    st = BNState()
    st.C, st.rows, st.groups = C, (N // groups) * H * W, groups
    st.aux = torch.empty((groups, 4, C), dtype=torch.float32, device=y.device)
    out = torch.empty((N, Co, H, W), dtype=_act_dtype(y), device=y.device, memory_format=CL)
    call('sba_bn_act_fwd_fused', _dt(y), _p(y), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
         _p(bn.num_batches_tracked), _p(st.aux), _p(out), st.rows, groups, C, act, Co, 0, BN_EPS, BN_MOMENTUM, _stream())
    return out, st


def bn_stats(y, groups=1):
    """per-group, per-channel (sum, sumsq) of an NHWC tensor by a separate pass (grouped batches)."""
    N, C, H, W = y.shape
    stats = zeros_f32((groups, BN_STAT_SLOTS, 2 * C), y.device)
    call('sba_bn_stats', _dt(y), _p(y), _p(stats), (N // groups) * H * W, groups, C, _stream())
    return stats


# ----------------------------------------------------------------------------
# autograd Functions
# ----------------------------------------------------------------------------
def _c(t):
    """gradient tensors arrive with arbitrary strides: force the NHWC layout."""
    if t.dim() == 4:
        return t.contiguous(memory_format=CL)
    return t.contiguous()


class ConvBNActFn(torch.autograd.Function):
    """conv (3x3 / nearest-x2 + 3x3 / 4x4 s2) -> BatchNorm(train) -> GLU | LeakyReLU | none (+ residual).
    upBlock model.py:39-45, Block3x3_leakRelu :540-546, downBlock :550-556, ResBlock halves :60-65.

    groups > 1: the batch holds `groups` independent BatchNorm batches back to back (the
    discriminator's real | fake passes, losses.py:139-140): ONE conv / dgrad / wgrad launch over
    the whole batch, BatchNorm statistics, running-stat updates and normalisation per group in
    order -- bit-for-bit the same module state as calling the block once per group."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, layer, kind, act, residual, groups=1):
        x = as_act(x)
        if groups == 1:
            y, stats = conv_forward(x, layer.pw, kind, want_stats=layer.bn.training, pre_bn=True)
        else:
            assert residual is None and x.shape[0] % groups == 0
            y, _ = conv_forward(x, layer.pw, kind, want_stats=False, pre_bn=True)
            rows_g = (y.shape[0] // groups) * y.shape[2] * y.shape[3]
            if layer.bn.training and rows_g <= BN_FUSED_BWD_ROWS:
                out, sts = bn_act_forward_fused(y, layer.bn, act, groups)
                ctx.layer, ctx.kind, ctx.act, ctx.sts, ctx.groups = layer, kind, act, sts, groups
                ctx.has_res = False
                ctx.save_for_backward(x, y)
                return out
            stats = bn_stats(y, groups) if layer.bn.training else None
        out, sts = bn_act_forward(y, stats, layer.bn, act, residual, groups)
        ctx.layer, ctx.kind, ctx.act, ctx.sts, ctx.groups = layer, kind, act, sts, groups
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y = ctx.saved_tensors
        layer, kind, act, groups = ctx.layer, ctx.kind, ctx.act, ctx.groups
        dout = _c(dout)
        dy = bn_act_backward(y, dout, ctx.sts, layer.bn, act, ctx.needs_input_grad[2])
        if ctx.needs_input_grad[1]:
            conv_wgrad_overlapped(x, dy, layer.conv.weight, kind)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv_dgrad(dy, layer.pw, kind, x.shape[2:])
        dres = dout if (ctx.has_res and ctx.needs_input_grad[7]) else None
        return dx, None, None, None, None, None, None, dres, None


class ResBlockFn(torch.autograd.Function):
    """ResBlock (model.py:57-71) as one node so that the skip gradient is added in the
    epilogue of the first conv's data-gradient instead of a separate pass."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, blk):
        x = as_act(x)
        y1, s1 = conv_forward(x, blk.l1.pw, '3x3', want_stats=blk.l1.bn.training, pre_bn=True)
        a1, st1 = bn_act_forward(y1, s1, blk.l1.bn, ACT_GLU)
        y2, s2 = conv_forward(a1, blk.l2.pw, '3x3', want_stats=blk.l2.bn.training, pre_bn=True)
        out, st2 = bn_act_forward(y2, s2, blk.l2.bn, ACT_NONE, residual=x)
        ctx.blk, ctx.st1, ctx.st2 = blk, st1, st2
        ctx.save_for_backward(x, y1, a1, y2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y1, a1, y2 = ctx.saved_tensors
        blk = ctx.blk
        dout = _c(dout)
        need_p = ctx.needs_input_grad[1]
        dy2 = bn_act_backward(y2, dout, ctx.st2, blk.l2.bn, ACT_NONE, need_p)
        if need_p:
            conv_wgrad_overlapped(a1, dy2, blk.l2.conv.weight, '3x3')
        da1 = conv_dgrad(dy2, blk.l2.pw, '3x3', a1.shape[2:])
        dy1 = bn_act_backward(y1, da1, ctx.st1, blk.l1.bn, ACT_GLU, need_p)
        if need_p:
            conv_wgrad_overlapped(x, dy1, blk.l1.conv.weight, '3x3')
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv_dgrad(dy1, blk.l1.pw, '3x3', x.shape[2:], addend=dout)
        return dx, None, None, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    """f32 dense layer y = x W^T + b (nn.Linear; model.py:278,306-313,330,354)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x)
        x = x.float().contiguous()
        B, K = x.shape
        N = weight.shape[0]
        y = torch.empty((B, N), dtype=torch.float32, device=x.device)
        call('sba_linear_fwd', _p(x), _p(weight), _p(bias), _p(y), B, K, N, _stream())
        ctx.save_for_backward(x, weight)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.float().contiguous()
        B, K = x.shape
        N = weight.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = param_grad(weight) if ctx.needs_input_grad[1] else None
        db = param_grad(ctx.bias) if (ctx.bias is not None and ctx.needs_input_grad[2]) else None
        if SIDE_WGRAD and HOME_STREAM == _stream() and dx is not None and dw is not None:
            # the chain of dense layers at the END of the generator's backward pass (MAPPING_NET, INIT_STAGE_G.fc,
            # CA_NET.fc): only dx continues the chain -- dW / db go to the weight-gradient companion stream
            with torch.cuda.stream(_companion(x.device, x, dy)):
                call('sba_linear_bwd', _p(x), _p(weight), _p(dy), None, _p(dw), _p(db), B, K, N, _stream())
            call('sba_linear_bwd', _p(x), _p(weight), _p(dy), _p(dx), None, None, B, K, N, _stream())
        else:
            call('sba_linear_bwd', _p(x), _p(weight), _p(dy), _p(dx), _p(dw), _p(db), B, K, N, _stream())
        return dx, None, None


class CAFn(torch.autograd.Function):
    """CA_NET tail: GLU, split, reparametrise with an explicit eps (model.py:281-294)."""

    @staticmethod
    def forward(ctx, h, eps):
        B, C4 = h.shape
        C = C4 // 4
        h = h.contiguous()
        eps = eps.float().contiguous()
        c, mu, lv = (torch.empty((B, C), dtype=torch.float32, device=h.device) for _ in range(3))
        call('sba_ca_fwd', _p(h), _p(eps), _p(c), _p(mu), _p(lv), B, C, _stream())
        ctx.save_for_backward(h, eps)
        return c, mu, lv

    @staticmethod
    def backward(ctx, dc, dmu, dlv):
        h, eps = ctx.saved_tensors
        B, C4 = h.shape
        dh = torch.empty_like(h)
        dc = None if dc is None else dc.contiguous()
        dmu = None if dmu is None else dmu.contiguous()
        dlv = None if dlv is None else dlv.contiguous()
        call('sba_ca_bwd', _p(h), _p(eps), _p(dc), _p(dmu), _p(dlv), _p(dh), B, C4 // 4, _stream())
        return dh, None


class FcBnGluFn(torch.autograd.Function):
    """INIT_STAGE_G.fc: Linear(no bias) -> BatchNorm1d(train) -> GLU -> view(B, ngf, 4, 4)
    (model.py:353-356,372-373); the output is written NHWC."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mod):
        _need_gpu(x)
        x = x.float().contiguous()
        B, K = x.shape
        F = weight.shape[0]
        y = torch.empty((B, F), dtype=torch.float32, device=x.device)
        call('sba_linear_fwd', _p(x), _p(weight), None, _p(y), B, K, F, _stream())
        bn = mod.bn
        aux = torch.empty((2, F), dtype=torch.float32, device=x.device)
        out = torch.empty((B, F // 32, 4, 4), dtype=COMPUTE_DTYPE, device=x.device, memory_format=CL)
        if not bn.training:
            # inference (netG.eval(): trainer.py:368 sampling / :437 gen_example): running statistics, plain
            # tensor ops -- this layer runs once per generated batch and is off the training path (nets._FcBnGlu
            # refuses the call when a gradient could be asked of it)
            yn = (y - bn.running_mean) * torch.rsqrt(bn.running_var + BN_EPS) * bn.weight + bn.bias
            glu = yn[:, :F // 2] * torch.sigmoid(yn[:, F // 2:])
            return glu.view(B, F // 32, 4, 4).to(COMPUTE_DTYPE).contiguous(memory_format=CL)
        call('sba_bn1d_glu_fwd', _dt(out), _p(y), _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
             _p(bn.running_var), _p(bn.num_batches_tracked), _p(aux[0]), _p(aux[1]), _p(out), B, F, BN_EPS,
             BN_MOMENTUM, _stream())
        ctx.mod, ctx.aux = mod, aux
        ctx.save_for_backward(x, y, weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, weight = ctx.saved_tensors
        bn = ctx.mod.bn
        dout = _c(dout)
        B, K = x.shape
        F = weight.shape[0]
        dy = torch.empty_like(y)
        call('sba_bn1d_glu_bwd', _dt(dout), _p(y), _p(dout), _p(bn.weight), _p(bn.bias), _p(ctx.aux[0]),
             _p(ctx.aux[1]), _p(dy), _p(param_grad(bn.weight)), _p(param_grad(bn.bias)), B, F, _stream())
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = param_grad(weight)
        if SIDE_WGRAD and HOME_STREAM == _stream() and dx is not None:       # (as LinearFn: beside the rest of the chain)
            with torch.cuda.stream(_companion(x.device, x, dy)):
                call('sba_linear_bwd', _p(x), _p(weight), _p(dy), None, _p(dw), None, B, K, F, _stream())
            call('sba_linear_bwd', _p(x), _p(weight), _p(dy), _p(dx), None, None, B, K, F, _stream())
        else:
            call('sba_linear_bwd', _p(x), _p(weight), _p(dy), _p(dx), _p(dw), None, B, K, F, _stream())
        return dx, None, None, None, None


# BASELINE config 5 ("fp8 MFMA for the attention / context GEMM"): the attention key projection (GlobalAttention.py:97
# conv_context) AND the two attention contractions (GlobalAttention.py:103 scores, :117 context) with FP8 (OCP e4m3)
# operands on v_mfma_f32_32x32x16_fp8_fp8 (forward only; the backward keeps bf16 / f32 operands).  Off by default.
ATTN_FP8 = os.environ.get('SBA_ATTN_FP8', '0') == '1'


def set_attention_fp8(flag):
    global ATTN_FP8
    ATTN_FP8 = bool(flag)


def _ctx_proj_fwd(words, wc, src, N, C, cdf, L):
    if ATTN_FP8 and cdf % 16 == 0:
        call('sba_ctx_proj_fwd_fp8', _p(words), _p(wc), _p(src), N, C, cdf, L, _stream())
    else:
        call('sba_ctx_proj_fwd', _p(words), _p(wc), _p(src), N, C, cdf, L, _stream())


def _word_attn_fwd(h, src, m8, out, att, N, HW, C, L, mask_mode, ocs, oco):
    if ATTN_FP8 and h.dtype == torch.bfloat16 and C in (32, 64):
        call('sba_word_attn_fwd_fp8', _p(h), _p(src), _p(m8), _p(out), _p(att), N, HW, C, L, mask_mode, ocs, oco,
             _stream())
    else:
        call('sba_word_attn_fwd', _dt(h), _p(h), _p(src), _p(m8), _p(out), _p(att), N, HW, C, L, mask_mode, ocs, oco,
             _stream())


def _mask_u8(mask):
    if mask is None:
        return None
    if mask.dtype == torch.bool and mask.is_contiguous():
        return mask.view(torch.uint8)       # the same bytes (False / True are stored as 0 / 1): no conversion launch
    # (a sliced mask -- trainer.py:253-256 cuts it to the longest caption -- is not contiguous.)  Both later stages of the
    # generator pass the SAME mask object: convert it once per generator forward (the cache is cleared at the start of every
    # forward, nets._GBase._run -- a recording must hold its own conversion launch, the caller refills a static mask between
    # replays) and per content (tensor identity + version counter)
    global _MASK_U8_CACHE
    c = _MASK_U8_CACHE
    if c is not None and c[0]() is mask and c[1] == mask._version and c[2].device == mask.device:
        return c[2]
    import weakref
    u8 = mask.to(torch.uint8).contiguous()
    try:
        _MASK_U8_CACHE = (weakref.ref(mask), mask._version, u8)
    except TypeError:
        _MASK_U8_CACHE = None
    return u8


_MASK_U8_CACHE = None


def reset_mask_cache():
    global _MASK_U8_CACHE
    _MASK_U8_CACHE = None


class CtxProjFn(torch.autograd.Function):
    """conv_context of GlobalAttentionGeneral (GlobalAttention.py:75,97) as a node of its own: src[b] = W words[b] depends on
    the captions and the weight only, so the generator evaluates it beside its first stage (nets._GBase._styles) and hands
    it to AttnAdainCatFn; autograd replays this node's backward on the stream of its forward."""

    @staticmethod
    def forward(ctx, words, w_ctx):
        words = words.float().contiguous()
        N, cdf, L = words.shape
        C = w_ctx.shape[0]
        src = torch.empty((N, C, L), dtype=torch.float32, device=words.device)
        _ctx_proj_fwd(words, w_ctx.detach().reshape(C, cdf), src, N, C, cdf, L)
        ctx.save_for_backward(words)
        ctx.w_ctx = w_ctx
        return src

    @staticmethod
    def backward(ctx, dsrc):
        (words,) = ctx.saved_tensors
        dsrc = dsrc.float().contiguous()
        N, cdf, L = words.shape
        C = ctx.w_ctx.shape[0]
        dwords = torch.empty_like(words) if ctx.needs_input_grad[0] else None
        wc = ctx.w_ctx.detach().reshape(C, cdf)
        if ctx.needs_input_grad[1] or dwords is not None:
            dW = param_grad(ctx.w_ctx) if ctx.needs_input_grad[1] else torch.zeros_like(wc)
            call('sba_ctx_proj_bwd', _p(words), _p(wc), _p(dsrc), _p(dW), _p(dwords), N, C, cdf, L, _stream())
        return dwords, None


class AttnAdainCatFn(torch.autograd.Function):
    """Entry of NEXT_STAGE_G (model.py:415-418): word attention (GlobalAttention.py:82-121),
    AdaIN (model.py:332-339) and the channel concat, written straight into one NHWC
    tensor [adain(h) | ctx].  Returns (h_c_code, att or empty)."""

    @staticmethod
    def forward(ctx, h, style, words, w_ctx, mask, want_att, mask_mode, src_in=None):
        """src_in: the key projection when the caller has evaluated it already (CtxProjFn): its gradient is then returned
        instead of being folded into dW / dwords here"""
        h = as_act(h)
        N, C, H, W = h.shape
        HW = H * W
        words = words.float().contiguous()
        cdf, L = words.shape[1], words.shape[2]
        style = style.float().contiguous()
        m8 = _mask_u8(mask)
        dev = h.device
        ctx.src_given = src_in is not None
        if src_in is not None:
            src = src_in.float().contiguous()
        else:
            src = torch.empty((N, C, L), dtype=torch.float32, device=dev)
            wc = w_ctx.detach().reshape(C, cdf)
            _ctx_proj_fwd(words, wc, src, N, C, cdf, L)
        out = empty_act(N, 2 * C, H, W, h)
        att = torch.empty((N, L, H, W), dtype=torch.float32, device=dev) if want_att else None
        _word_attn_fwd(h, src, m8, out, att, N, HW, C, L, mask_mode, 2 * C, C)
        mr = torch.empty((2, N, C), dtype=torch.float32, device=dev)
        call('sba_instnorm_stats', _dt(h), _p(h), _p(mr[0]), _p(mr[1]), N, HW, C, IN_EPS, _stream())
        call('sba_adain_fwd', _dt(h), _p(h), _p(mr[0]), _p(mr[1]), _p(style), _p(out), N, HW, C, 2 * C, 0,
             _stream())
        ctx.save_for_backward(h, style, words, src, mr)
        ctx.m8, ctx.w_ctx, ctx.mask_mode = m8, w_ctx, mask_mode
        if att is None:
            att = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, dout, _datt):
        h, style, words, src, mr = ctx.saved_tensors
        dout = _c(dout)
        N, C, H, W = h.shape
        HW = H * W
        cdf, L = words.shape[1], words.shape[2]
        dev = h.device
        dh = torch.empty_like(h)
        dsrc = zeros_f32((N, C, L), dev)
        call('sba_word_attn_bwd', _dt(h), _p(h), _p(src), _p(ctx.m8), _p(dout), _p(dh), _p(dsrc), N, HW, C, L,
             ctx.mask_mode, 2 * C, C, 0, _stream())
        red = zeros_f32((N, C, 2), dev)
        call('sba_adain_bwd_reduce', _dt(h), _p(h), _p(dout), _p(mr[0]), _p(mr[1]), _p(red), N, HW, C, 2 * C, 0,
             _stream())
        dstyle = torch.empty_like(style)
        call('sba_adain_bwd_apply', _dt(h), _p(h), _p(dout), _p(mr[0]), _p(mr[1]), _p(style), _p(red), _p(dh),
             _p(dstyle), N, HW, C, 2 * C, 0, 1, _stream())
        if ctx.src_given:
            return dh, dstyle, None, None, None, None, None, dsrc
        dwords = torch.empty_like(words) if ctx.needs_input_grad[2] else None
        wc = ctx.w_ctx.detach().reshape(C, cdf)
        if ctx.needs_input_grad[3] or dwords is not None:
            dW = param_grad(ctx.w_ctx) if ctx.needs_input_grad[3] else torch.zeros_like(wc)
            call('sba_ctx_proj_bwd', _p(words), _p(wc), _p(dsrc), _p(dW), _p(dwords), N, C, cdf, L, _stream())
        return dh, dstyle, dwords, None, None, None, None, None


class WordAttnFn(torch.autograd.Function):
    """Stand-alone GlobalAttentionGeneral.forward (GlobalAttention.py:82-121)."""

    @staticmethod
    def forward(ctx, h, words, w_ctx, mask, mask_mode):
        h = as_act(h)
        N, C, H, W = h.shape
        words = words.float().contiguous()
        cdf, L = words.shape[1], words.shape[2]
        m8 = _mask_u8(mask)
        src = torch.empty((N, C, L), dtype=torch.float32, device=h.device)
        wc = w_ctx.detach().reshape(C, cdf)
        _ctx_proj_fwd(words, wc, src, N, C, cdf, L)
        out = empty_act(N, C, H, W, h)
        att = torch.empty((N, L, H, W), dtype=torch.float32, device=h.device)
        _word_attn_fwd(h, src, m8, out, att, N, H * W, C, L, mask_mode, C, 0)
        ctx.save_for_backward(h, words, src)
        ctx.m8, ctx.w_ctx, ctx.mask_mode = m8, w_ctx, mask_mode
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, dout, _datt):
        h, words, src = ctx.saved_tensors
        dout = _c(dout)
        N, C, H, W = h.shape
        cdf, L = words.shape[1], words.shape[2]
        dh = torch.empty_like(h)
        dsrc = torch.zeros((N, C, L), dtype=torch.float32, device=h.device)
        call('sba_word_attn_bwd', _dt(h), _p(h), _p(src), _p(ctx.m8), _p(dout), _p(dh), _p(dsrc), N, H * W, C, L,
             ctx.mask_mode, C, 0, 0, _stream())
        dwords = torch.empty_like(words) if ctx.needs_input_grad[1] else None
        wc = ctx.w_ctx.detach().reshape(C, cdf)
        dW = param_grad(ctx.w_ctx) if ctx.needs_input_grad[2] else torch.zeros_like(wc)
        call('sba_ctx_proj_bwd', _p(words), _p(wc), _p(dsrc), _p(dW), _p(dwords), N, C, cdf, L, _stream())
        return dh, dwords, None, None, None


class AdainFn(torch.autograd.Function):
    """Stand-alone ADAIN_NORM core (model.py:336-337) given style = Linear(w)."""

    @staticmethod
    def forward(ctx, h, style):
        h = as_act(h)
        N, C, H, W = h.shape
        style = style.float().contiguous()
        mr = torch.empty((2, N, C), dtype=torch.float32, device=h.device)
        call('sba_instnorm_stats', _dt(h), _p(h), _p(mr[0]), _p(mr[1]), N, H * W, C, IN_EPS, _stream())
        out = torch.empty_like(h)
        call('sba_adain_fwd', _dt(h), _p(h), _p(mr[0]), _p(mr[1]), _p(style), _p(out), N, H * W, C, C, 0, _stream())
        ctx.save_for_backward(h, style, mr)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, style, mr = ctx.saved_tensors
        dout = _c(dout)
        N, C, H, W = h.shape
        red = torch.zeros((N, C, 2), dtype=torch.float32, device=h.device)
        call('sba_adain_bwd_reduce', _dt(h), _p(h), _p(dout), _p(mr[0]), _p(mr[1]), _p(red), N, H * W, C, C, 0,
             _stream())
        dh = torch.empty_like(h)
        dstyle = torch.empty_like(style)
        call('sba_adain_bwd_apply', _dt(h), _p(h), _p(dout), _p(mr[0]), _p(mr[1]), _p(style), _p(red), _p(dh),
             _p(dstyle), N, H * W, C, C, 0, 0, _stream())
        return dh, dstyle


class ImgHeadFn(torch.autograd.Function):
    """GET_IMAGE_G: conv3x3(ngf->3) + tanh, NHWC features -> NCHW f32 image (model.py:426-437)."""

    @staticmethod
    def forward(ctx, h, weight):
        h = as_act(h)
        reduce_scratch(h.device)         # (allocated here, outside any later graph capture of the backward)
        N, C, H, W = h.shape
        if not weight.is_contiguous(memory_format=CL):
            raise RuntimeError('img head weight must be channels_last')
        img = torch.empty((N, 3, H, W), dtype=torch.float32, device=h.device)
        call('sba_img_head_fwd', _dt(h), _p(h), _p(weight), _p(img), N, H, W, C, _stream())
        ctx.save_for_backward(h, weight, img)
        return img

    @staticmethod
    def backward(ctx, dimg):
        h, weight, img = ctx.saved_tensors
        dimg = dimg.float().contiguous()
        N, C, H, W = h.shape
        dh = torch.empty_like(h)
        dw = param_grad(weight) if ctx.needs_input_grad[1] else torch.zeros_like(weight)
        call('sba_img_head_bwd', _dt(h), _p(h), _p(weight), _p(img), _p(dimg), _p(dh), _p(dw), N, H, W, C, 0,
             _stream())
        return dh, None


class DStemFn(torch.autograd.Function):
    """conv4x4 s2 (3->ndf) + LeakyReLU(0.2): NCHW f32 image -> NHWC features (model.py:563-564)."""

    @staticmethod
    def forward(ctx, img, weight):
        _need_gpu(img)
        reduce_scratch(img.device)       # (allocated here, outside any later graph capture of the backward)
        img = img.float().contiguous()
        N, _, S, S2 = img.shape
        assert S == S2
        C = weight.shape[0]
        out = torch.empty((N, C, S // 2, S // 2), dtype=COMPUTE_DTYPE, device=img.device, memory_format=CL)
        call('sba_d_stem_fwd', _dt(out), _p(img), _p(weight), _p(out), N, S, C, _stream())
        ctx.save_for_backward(img, weight, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        img, weight, out = ctx.saved_tensors
        dout = _c(dout)
        N, _, S, _ = img.shape
        C = weight.shape[0]
        dimg = torch.empty_like(img) if ctx.needs_input_grad[0] else None
        dw = param_grad(weight) if ctx.needs_input_grad[1] else None
        if dimg is not None or dw is not None:
            call('sba_d_stem_bwd', _dt(out), _p(img), _p(weight), _p(out), _p(dout), _p(dimg), _p(dw), N, S, C,
                 _stream())
        return dimg, None


class LogitsFn(torch.autograd.Function):
    """outlogits: conv4x4 s4 (8ndf->1, bias) + sigmoid on the 4x4 map (model.py:590-592,606-607)."""

    @staticmethod
    def forward(ctx, h, weight, bias):
        h = as_act(h)
        B, C, H, W = h.shape
        assert H == 4 and W == 4
        prob = torch.empty(B, dtype=torch.float32, device=h.device)
        call('sba_logits_fwd', _dt(h), _p(h), _p(weight), _p(bias), _p(prob), B, 16 * C, _stream())
        ctx.save_for_backward(h, weight, prob)
        ctx.bias = bias
        return prob

    @staticmethod
    def backward(ctx, dprob):
        h, weight, prob = ctx.saved_tensors
        dprob = dprob.float().contiguous()
        B, C = h.shape[0], h.shape[1]
        dh = torch.empty_like(h)
        dw = param_grad(weight) if ctx.needs_input_grad[1] else None
        db = param_grad(ctx.bias) if ctx.needs_input_grad[2] else None
        call('sba_logits_bwd', _dt(h), _p(h), _p(weight), _p(prob), _p(dprob), _p(dh), _p(dw), _p(db), B, 16 * C, 0,
             _stream())
        return dh, None, None


class CondCatFn(torch.autograd.Function):
    """cat(h_code, c_code tiled 4x4) along channels (model.py:597-600)."""

    @staticmethod
    def forward(ctx, h, sent):
        h = as_act(h)
        B, C = h.shape[0], h.shape[1]
        sent = sent.float().contiguous()
        E = sent.shape[1]
        out = empty_act(B, C + E, 4, 4, h)
        call('sba_cond_cat_fwd', _dt(h), _p(h), _p(sent), _p(out), B, C, E, _stream())
        ctx.dims = (B, C, E)
        ctx.like = h
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, E = ctx.dims
        dout = _c(dout)
        dh = empty_act(B, C, 4, 4, dout)
        ds = torch.zeros((B, E), dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[1] else None
        call('sba_cond_cat_bwd', _dt(dout), _p(dout), _p(dh), _p(ds), B, C, E, 0, _stream())
        return dh, ds


class BCEMultiFn(torch.autograd.Function):
    """sum_s weight_s * BCELoss(prob_s, target_s) in one launch (losses.py:144-158,175-182)."""

    @staticmethod
    def forward(ctx, targets, weights, *probs):
        dev = probs[0].device
        sizes = [int(p.numel()) for p in probs]
        cat = torch.cat([p.float().reshape(-1) for p in probs])
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        key = (tuple(offs), tuple(targets), tuple(weights), dev)
        meta = _BCE_META.get(key)
        if meta is None:
            meta = (torch.tensor(offs, dtype=torch.int32, device=dev),
                    torch.tensor(targets, dtype=torch.float32, device=dev),
                    torch.tensor(weights, dtype=torch.float32, device=dev))
            _BCE_META[key] = meta
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dprob = torch.empty_like(cat)
        call('sba_bce_multi', _p(cat), _p(meta[0]), _p(meta[1]), _p(meta[2]), len(sizes), _p(loss), _p(dprob),
             _stream())
        ctx.sizes = sizes
        ctx.save_for_backward(dprob)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dprob,) = ctx.saved_tensors
        d = dprob * g
        return (None, None) + tuple(d.split(ctx.sizes))


_BCE_META = {}


def _bce_meta(offs, targets, weights, dev):
    key = (tuple(offs), tuple(targets), tuple(weights), dev)
    meta = _BCE_META.get(key)
    if meta is None:
        meta = (torch.tensor(offs, dtype=torch.int32, device=dev),
                torch.tensor(targets, dtype=torch.float32, device=dev),
                torch.tensor(weights, dtype=torch.float32, device=dev))
        _BCE_META[key] = meta
    return meta


GROUP_COND_HEADS = os.environ.get('SBA_GROUP_COND_HEADS', '1') != '0'


class DHeadsFn(torch.autograd.Function):
    """Every head of one discriminator's loss term as ONE autograd node: the conditional heads
    (cat with the sentence code, jointConv + BatchNorm + LeakyReLU, logits; model.py:594-607), the unconditional
    heads (model.py:590-592) and the weighted BCE sum (losses.py:141-158 for the discriminator update,
    :169-178 for the generator term).

    `heads` = tuple of (row0, rows, cond_row0 | None, target, weight, segment): the head reads feats[row0:row0+rows]
    (and cond[cond_row0:cond_row0+rows] when conditional); they are evaluated in the order given (the reference's
    call order, which fixes the order of the jointConv BatchNorm running-statistic updates) and summed in `segment`
    order (the reference's order of the terms of errD / g_loss).  The same kernels as the per-head Functions
    run; what goes away is the autograd glue between them (row slices copied into zero-filled gradients, gradient sums
    of the shared feature map, cat of the probabilities): each head's backward writes or accumulates straight into
    its rows of d feats."""

    @staticmethod
    def forward(ctx, feats, cond, netD, heads, *params):
        feats = as_act(feats)
        dev = feats.device
        dt = _dt(feats)
        C = feats.shape[1]
        assert feats.shape[2] == 4 and feats.shape[3] == 4
        K = 16 * C
        cnet, unet = netD.COND_DNET, netD.UNCOND_DNET
        if cond is not None:
            cond = cond.reshape(-1, cnet.ef_dim).float().contiguous()
        sizes = [0] * len(heads)
        for (r0, rows, c0, tgt, wt, seg) in heads:
            sizes[seg] = rows
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        prob = torch.empty(offs[-1], dtype=torch.float32, device=dev)
        tape = []
        # The conditional heads of one term (real / fake / wrong pairs, losses.py:143-149) share jointConv's weights: their
        # inputs go into ONE tensor, one conv / data-gradient / weight-gradient launch serves all of them (M = 944 rows at
        # B = 20 instead of three GEMM-like launches of 320 / 320 / 304 rows, each with its own K split + finishing launch);
        # BatchNorm stays per head, in the reference's call order (batch statistics and running-statistic updates of three
        # separate module calls).
        cond_idx = [i for i, hd in enumerate(heads) if hd[2] is not None]
        grouped = GROUP_COND_HEADS and len(cond_idx) >= 2 and cnet.jointConv._layer().bn.training
        xin_all = y_all = None
        slices = {}
        if grouped:
            layer = cnet.jointConv._layer()
            E = cond.shape[1]
            total = sum(heads[i][1] for i in cond_idx)
            xin_all = empty_act(total, C + E, 4, 4, feats)
            off = 0
            for i in cond_idx:
                r0, rows, c0 = heads[i][0], heads[i][1], heads[i][2]
                call('sba_cond_cat_fwd', dt, _p(feats[r0:r0 + rows]), _p(cond[c0:c0 + rows]), _p(xin_all[off:off + rows]),
                     rows, C, E, _stream())
                slices[i] = (off, rows)
                off += rows
            y_all, _ = conv_forward(xin_all, layer.pw, '3x3', want_stats=False, pre_bn=True)
        for hi, (r0, rows, c0, tgt, wt, seg) in enumerate(heads):
            h = feats[r0:r0 + rows]
            pslice = prob[offs[seg]:offs[seg] + rows]
            if c0 is not None and grouped:
                layer = cnet.jointConv._layer()
                off = slices[hi][0]
                xin, y = xin_all[off:off + rows], y_all[off:off + rows]
                if rows * 16 <= BN_FUSED_BWD_ROWS:
                    hc, st = bn_act_forward_fused(y, layer.bn, ACT_LRELU, 1)
                else:
                    hc, st = bn_act_forward(y, bn_stats(y, 1), layer.bn, ACT_LRELU)
                o = cnet.outlogits[0]
                call('sba_logits_fwd', dt, _p(hc), _p(o.weight), _p(o.bias), _p(pslice), rows, K, _stream())
                tape.append((xin, y, st, hc))
            elif c0 is not None:
                layer = cnet.jointConv._layer()
                E = cond.shape[1]
                xin = empty_act(rows, C + E, 4, 4, feats)
                call('sba_cond_cat_fwd', dt, _p(h), _p(cond[c0:c0 + rows]), _p(xin), rows, C, E, _stream())
                y, stats = conv_forward(xin, layer.pw, '3x3', want_stats=layer.bn.training, pre_bn=True)
                hc, st = bn_act_forward(y, stats, layer.bn, ACT_LRELU)
                o = cnet.outlogits[0]
                call('sba_logits_fwd', dt, _p(hc), _p(o.weight), _p(o.bias), _p(pslice), rows, K, _stream())
                tape.append((xin, y, st, hc))
            else:
                o = unet.outlogits[0]
                call('sba_logits_fwd', dt, _p(h), _p(o.weight), _p(o.bias), _p(pslice), rows, K, _stream())
                tape.append(None)
        targets = [0.] * len(heads)
        weights = [0.] * len(heads)
        for (r0, rows, c0, tgt, wt, seg) in heads:
            targets[seg], weights[seg] = tgt, wt
        meta = _bce_meta(offs, targets, weights, dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dprob = torch.empty_like(prob)
        call('sba_bce_multi', _p(prob), _p(meta[0]), _p(meta[1]), _p(meta[2]), len(heads), _p(loss), _p(dprob),
             _stream())
        ctx.netD, ctx.heads, ctx.offs, ctx.tape = netD, heads, offs, tape
        ctx.grouped = (xin_all, y_all, slices) if grouped else None
        ctx.has_cond = cond is not None
        ctx.cond_shape = None if cond is None else tuple(cond.shape)
        ctx.save_for_backward(feats, prob, dprob)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        feats, prob, dprob = ctx.saved_tensors
        netD, heads, offs = ctx.netD, ctx.heads, ctx.offs
        cnet, unet = netD.COND_DNET, netD.UNCOND_DNET
        dt = _dt(feats)
        C = feats.shape[1]
        K = 16 * C
        d = dprob * g
        need_feats = ctx.needs_input_grad[0]
        need_cond = ctx.has_cond and ctx.needs_input_grad[1]
        need_p = any(ctx.needs_input_grad[4:])
        dfeats = torch.empty_like(feats) if need_feats else None
        dcond = torch.zeros(ctx.cond_shape, dtype=torch.float32, device=feats.device) if need_cond else None
        # a head whose rows nobody has written yet stores, one whose rows are all written accumulates; widest
        # heads first, so that the row ranges of the reference's five heads never overlap partially
        order = sorted(range(len(heads)), key=lambda i: -heads[i][1])
        written = []

        def mode(r0, rows):
            inside = any(a <= r0 and r0 + rows <= b for a, b in written)
            if not inside:
                assert all(r0 + rows <= a or b <= r0 for a, b in written), 'partially overlapping head rows'
                written.append((r0, r0 + rows))
            return 1 if inside else 0

        dy_all = None
        if ctx.grouped is not None:
            xin_all, y_all, slices = ctx.grouped
            dy_all = torch.empty(y_all.shape, dtype=_act_dtype(y_all), device=y_all.device, memory_format=CL)
        for i in order:
            r0, rows, c0, tgt, wt, seg = heads[i]
            ps, ds = prob[offs[seg]:offs[seg] + rows], d[offs[seg]:offs[seg] + rows]
            if c0 is not None and dy_all is not None:
                # grouped conditional heads: this head's BatchNorm backward into its rows of dy_all; the conv gradients
                # follow once, below
                xin, y, st, hc = ctx.tape[i]
                layer = cnet.jointConv._layer()
                o = cnet.outlogits[0]
                dhc = torch.empty_like(hc)
                call('sba_logits_bwd', dt, _p(hc), _p(o.weight), _p(ps), _p(ds), _p(dhc),
                     _p(param_grad(o.weight)) if need_p else None, _p(param_grad(o.bias)) if need_p else None,
                     rows, K, 0, _stream())
                off = slices[i][0]
                bn_act_backward(y, dhc, st, layer.bn, ACT_LRELU, need_p, out=dy_all[off:off + rows])
                continue
            if c0 is None:
                o = unet.outlogits[0]
                if not (need_feats or need_p):
                    continue
                dh = dfeats[r0:r0 + rows] if need_feats else torch.empty_like(feats[r0:r0 + rows])
                call('sba_logits_bwd', dt, _p(feats[r0:r0 + rows]), _p(o.weight), _p(ps), _p(ds), _p(dh),
                     _p(param_grad(o.weight)) if need_p else None, _p(param_grad(o.bias)) if need_p else None,
                     rows, K, mode(r0, rows) if need_feats else 0, _stream())
                continue
            xin, y, st, hc = ctx.tape[i]
            layer = cnet.jointConv._layer()
            o = cnet.outlogits[0]
            dhc = torch.empty_like(hc)
            call('sba_logits_bwd', dt, _p(hc), _p(o.weight), _p(ps), _p(ds), _p(dhc),
                 _p(param_grad(o.weight)) if need_p else None, _p(param_grad(o.bias)) if need_p else None,
                 rows, K, 0, _stream())
            dy = bn_act_backward(y, dhc, st, layer.bn, ACT_LRELU, need_p)
            if need_p:
                conv_wgrad_overlapped(xin, dy, layer.conv.weight, '3x3')
            if need_feats or need_cond:
                dxin = conv_dgrad(dy, layer.pw, '3x3', (4, 4))
                E = xin.shape[1] - C
                dh = dfeats[r0:r0 + rows] if need_feats else torch.empty_like(feats[r0:r0 + rows])
                call('sba_cond_cat_bwd', dt, _p(dxin), _p(dh), _p(dcond[c0:c0 + rows]) if need_cond else None,
                     rows, C, E, mode(r0, rows) if need_feats else 0, _stream())
        if dy_all is not None:
            layer = cnet.jointConv._layer()
            if need_p:
                conv_wgrad_overlapped(xin_all, dy_all, layer.conv.weight, '3x3')
            if need_feats or need_cond:
                dxin_all = conv_dgrad(dy_all, layer.pw, '3x3', (4, 4))
                E = xin_all.shape[1] - C
                for i in order:
                    r0, rows, c0, tgt, wt, seg = heads[i]
                    if c0 is None:
                        continue
                    off = slices[i][0]
                    dh = dfeats[r0:r0 + rows] if need_feats else torch.empty_like(feats[r0:r0 + rows])
                    call('sba_cond_cat_bwd', dt, _p(dxin_all[off:off + rows]), _p(dh),
                         _p(dcond[c0:c0 + rows]) if need_cond else None, rows, C, E,
                         mode(r0, rows) if need_feats else 0, _stream())
        if need_feats:
            covered = sorted(written)
            assert covered and covered[0][0] == 0 and covered[-1][1] == feats.shape[0] and \
                all(covered[k][1] == covered[k + 1][0] for k in range(len(covered) - 1)), 'rows without a head'
        return (dfeats, dcond, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


def d_heads(netD, feats, cond, heads):
    """DHeadsFn with the parameters of netD's heads passed as autograd inputs (their gradients go to the flat
    gradient buffers like every other layer's)."""
    params = []
    if netD.COND_DNET is not None:
        c = netD.COND_DNET
        if c.bcondition:
            l = c.jointConv._layer()
            params += [l.conv.weight, l.bn.weight, l.bn.bias]
        params += [c.outlogits[0].weight, c.outlogits[0].bias]
    if netD.UNCOND_DNET is not None:
        u = netD.UNCOND_DNET
        params += [u.outlogits[0].weight, u.outlogits[0].bias]
    return DHeadsFn.apply(feats, cond, netD, tuple(heads), *params)


class KLFn(torch.autograd.Function):
    """KL_loss (losses.py:210-214)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=mu.device)
        dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
        call('sba_kl_loss', _p(mu), _p(logvar), _p(loss), _p(dmu), _p(dlv), mu.numel(), _stream())
        ctx.save_for_backward(dmu, dlv)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dmu, dlv = ctx.saved_tensors
        return dmu * g, dlv * g


DAMSM_MFMA = os.environ.get('SBA_DAMSM_MFMA', '1') != '0'


def _damsm_prep(feat, words, cap_lens, B, nef, R, L):
    """bf16 hi + lo operand copies for the matrix-core words loss (include/sbagan_hip.h: sba_damsm_prep), or None when the
    shape is outside that path (the f32 kernels then run)"""
    if not DAMSM_MFMA:
        return None
    nbytes = int(_lib.lib.sba_damsm_prep_bytes(B, nef, R, L))
    if nbytes <= 0:
        return None
    prep = torch.empty(nbytes, dtype=torch.uint8, device=feat.device)
    call('sba_damsm_prep', _p(feat), _p(words), _p(cap_lens), _p(prep), nbytes, B, nef, R, L, _stream())
    return prep


def _damsm_words_fwd(prep, feat, words, cap_lens, sim, attn, attn1, wctx, B, nef, R, L, g1, g2, st):
    if prep is not None:
        call('sba_damsm_words_fwd_mfma', _p(prep), _p(words), _p(cap_lens), _p(sim), _p(attn), _p(attn1), _p(wctx), B, nef,
             R, L, g1, g2, st)
    else:
        call('sba_damsm_words_fwd', _p(feat), _p(words), _p(cap_lens), _p(sim), _p(attn), _p(attn1), _p(wctx), B, nef, R,
             L, g1, g2, st)


def _damsm_words_bwd(prep, feat, words, cap_lens, sim, attn, attn1, wctx, dsim, dwords, B, nef, R, L, g1, g2, st):
    """returns dfeat (f32, feat's shape); dwords (zero-filled or None) is accumulated into"""
    if prep is not None:
        nbytes = int(_lib.lib.sba_damsm_bwd_bytes(B, nef, R, L))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=feat.device)
        dfeat = torch.empty_like(feat)              # (the matrix-core path STORES every element)
        call('sba_damsm_words_bwd_mfma', _p(prep), _p(words), _p(cap_lens), _p(sim), _p(attn), _p(attn1), _p(wctx),
             _p(dsim), _p(scratch), nbytes, _p(dfeat), 0, _p(dwords), B, nef, R, L, g1, g2, st)
    else:
        dfeat = torch.zeros_like(feat)
        call('sba_damsm_words_bwd', _p(feat), _p(words), _p(cap_lens), _p(sim), _p(attn), _p(attn1), _p(wctx), _p(dsim),
             _p(dfeat), _p(dwords), B, nef, R, L, g1, g2, st)
    return dfeat


class WordsLossFn(torch.autograd.Function):
    """words_loss (losses.py:62-132): returns (loss0, loss1)."""

    @staticmethod
    def forward(ctx, feat, words, cap_lens, mask, gammas):
        _need_gpu(feat)
        g1, g2, g3 = gammas
        B, nef = feat.shape[0], feat.shape[1]
        R = feat.shape[2] * feat.shape[3]
        feat = feat.float().contiguous()
        words = words.float().contiguous()
        L = words.shape[2]
        dev = feat.device
        cap_lens = cap_lens.to(device=dev, dtype=torch.int64).contiguous()
        sim = torch.empty((B, B), dtype=torch.float32, device=dev)
        attn = torch.empty((B * B, L, R), dtype=torch.float32, device=dev)
        attn1 = torch.empty((B * B, L, R), dtype=torch.float32, device=dev)
        wctx = torch.empty((B * B, L, nef), dtype=torch.float32, device=dev)
        prep = _damsm_prep(feat, words, cap_lens, B, nef, R, L)
        _damsm_words_fwd(prep, feat, words, cap_lens, sim, attn, attn1, wctx, B, nef, R, L, g1, g2, _stream())
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        d0, d1 = torch.empty_like(sim), torch.empty_like(sim)
        call('sba_ce_pair', _p(sim), _p(mask), g3, _p(loss), _p(d0), _p(d1), B, _stream())
        ctx.save_for_backward(feat, words, cap_lens, sim, attn, attn1, wctx, d0, d1)
        ctx.g = (g1, g2)
        ctx.prep = prep
        ctx.fshape = None
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, gl0, gl1):
        feat, words, cap_lens, sim, attn, attn1, wctx, d0, d1 = ctx.saved_tensors
        B, nef = feat.shape[0], feat.shape[1]
        R = feat.shape[2] * feat.shape[3]
        L = words.shape[2]
        dev = feat.device
        z = torch.zeros(1, dtype=torch.float32, device=dev)
        gl0 = z if gl0 is None else gl0.reshape(1).float()
        gl1 = z if gl1 is None else gl1.reshape(1).float()
        dsim = torch.empty_like(sim)
        call('sba_combine2', _p(dsim), _p(d0), _p(gl0), _p(d1), _p(gl1), B * B, _stream())
        dwords = torch.zeros_like(words) if ctx.needs_input_grad[1] else None
        dfeat = _damsm_words_bwd(ctx.prep, feat, words, cap_lens, sim, attn, attn1, wctx, dsim, dwords, B, nef, R, L,
                                 ctx.g[0], ctx.g[1], _stream())
        return dfeat, dwords, None, None, None


class SentLossFn(torch.autograd.Function):
    """sent_loss (losses.py:20-59): returns (loss0, loss1)."""

    @staticmethod
    def forward(ctx, cnn, rnn, mask, gamma3, eps):
        _need_gpu(cnn)
        cnn, rnn = cnn.float().contiguous(), rnn.float().contiguous()
        B, nef = cnn.shape
        dev = cnn.device
        s = torch.empty((B, B), dtype=torch.float32, device=dev)
        call('sba_damsm_sent_fwd', _p(cnn), _p(rnn), _p(s), B, nef, gamma3, eps, _stream())
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        d0, d1 = torch.empty_like(s), torch.empty_like(s)
        call('sba_ce_pair', _p(s), _p(mask), 1.0, _p(loss), _p(d0), _p(d1), B, _stream())
        ctx.save_for_backward(cnn, rnn, d0, d1)
        ctx.k = (gamma3, eps)
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, gl0, gl1):
        cnn, rnn, d0, d1 = ctx.saved_tensors
        B, nef = cnn.shape
        dev = cnn.device
        z = torch.zeros(1, dtype=torch.float32, device=dev)
        gl0 = z if gl0 is None else gl0.reshape(1).float()
        gl1 = z if gl1 is None else gl1.reshape(1).float()
        ds = torch.empty_like(d0)
        call('sba_combine2', _p(ds), _p(d0), _p(gl0), _p(d1), _p(gl1), B * B, _stream())
        dcnn = torch.zeros_like(cnn) if ctx.needs_input_grad[0] else None
        drnn = torch.zeros_like(rnn) if ctx.needs_input_grad[1] else None
        call('sba_damsm_sent_bwd', _p(cnn), _p(rnn), _p(ds), _p(dcnn), _p(drnn), B, nef, ctx.k[0], ctx.k[1],
             _stream())
        return dcnn, drnn, None, None, None


_LAMBDA_CELL = {}
_DAMSM_SIDE = {}


def _damsm_side_stream(dev):
    s = _DAMSM_SIDE.get(dev)
    if s is None:
        s = _DAMSM_SIDE[dev] = torch.cuda.Stream(device=dev)
    return s


def damsm_terms_direct(region_features, cnn_code, words_embs, sent_emb, cap_lens, mask, gammas, lam, eps=1e-8):
    """w_loss = LAMBDA (loss0 + loss1) of words_loss, s_loss likewise of sent_loss (losses.py:187-204), TOGETHER WITH
    their gradients w.r.t. the image-side inputs (region features, global code), no autograd in between.  For a frozen
    text side the upstream gradient of the four cross-entropy terms is the constant LAMBDA, so the words head is
    prep (2 launches) -> forward -> cross entropies + their gradient (sba_ce_pair_direct) -> backward (2 launches), the
    sentence loss ONE launch (sba_damsm_sent_direct): 7 launches on the image encoder's chain -- the critical one of the
    step -- where the autograd path issues ~35 (scalar adds, LAMBDA multiplications, fills).  B <= 96.
    Returns (w_loss, s_loss, d region_features, d cnn_code)."""
    g1, g2, g3 = gammas
    feat = region_features.detach().float().contiguous()
    cnn = cnn_code.detach().float().contiguous()
    words = words_embs.detach().float().contiguous()
    rnn = sent_emb.detach().float().contiguous()
    B, nef = feat.shape[0], feat.shape[1]
    R = feat.shape[2] * feat.shape[3]
    L = words.shape[2]
    dev = feat.device
    cap_lens = cap_lens.to(device=dev, dtype=torch.int64).contiguous()
    st = _stream()
    losses = torch.empty(2, dtype=torch.float32, device=dev)        # [w_loss, s_loss], written by the two head kernels
    dcnn = torch.empty_like(cnn)
    # the sentence loss depends on nothing of the words path: a one-workgroup launch (96 us inside the step) -- on a side
    # stream, beside the words kernels; joined below (its outputs were allocated above, on this stream, and outlive the join)
    cur = torch.cuda.current_stream()
    side = _damsm_side_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        call('sba_damsm_sent_direct', _p(cnn), _p(rnn), _p(mask), g3, eps, float(lam), _p(losses[1:2]), _p(dcnn), B, nef,
             _stream())
    sim = torch.empty((B, B), dtype=torch.float32, device=dev)
    attn = torch.empty((B * B, L, R), dtype=torch.float32, device=dev)
    attn1 = torch.empty((B * B, L, R), dtype=torch.float32, device=dev)
    wctx = torch.empty((B * B, L, nef), dtype=torch.float32, device=dev)
    prep = _damsm_prep(feat, words, cap_lens, B, nef, R, L)
    _damsm_words_fwd(prep, feat, words, cap_lens, sim, attn, attn1, wctx, B, nef, R, L, g1, g2, st)
    dsim = torch.empty_like(sim)
    call('sba_ce_pair_direct', _p(sim), _p(mask), g3, float(lam), _p(losses[0:1]), _p(dsim), B, st)
    dfeat = _damsm_words_bwd(prep, feat, words, cap_lens, sim, attn, attn1, wctx, dsim, None, B, nef, R, L, g1, g2, st)
    cur.wait_stream(side)
    w_loss, s_loss = losses[0], losses[1]
    return w_loss, s_loss, dfeat, dcnn


# ----------------------------------------------------------------------------
# text encoder
def lstm_bidir_forward(captions, cap_lens, emb_weight, w_ih, w_hh, b_ih, b_hh, hidden=None, max_len=None, out=None):
    """RNN_ENCODER.forward (model.py:127-159) of the frozen text encoder, sync-free: cap_lens stays on the
    device.  w_ih [2][4H][ninput], w_hh [2][4H][H], b_* [2][4H]; returns (words_emb B x 2H x L,
    sent_emb B x 2H) with L = max_len (the reference's max(cap_lens), if the host knows it) or T."""
    _need_gpu(captions)
    B, T = captions.shape
    H = w_hh.shape[2]
    L = T if max_len is None else int(max_len)
    dev = captions.device
    captions = captions.to(torch.int64).contiguous()
    cap_lens = cap_lens.to(device=dev, dtype=torch.int64).contiguous()
    gx = torch.empty((2, B * T, 4 * H), dtype=torch.float32, device=dev)
    if out is not None:         # caller-owned (static) output tensors: no copy in a recorded / captured step
        words, sent = out
        assert words.shape == (B, 2 * H, L) and sent.shape == (B, 2 * H) and words.is_contiguous() and \
            sent.is_contiguous() and words.dtype == torch.float32 and sent.dtype == torch.float32
    else:
        words = torch.empty((B, 2 * H, L), dtype=torch.float32, device=dev)
        sent = torch.empty((B, 2 * H), dtype=torch.float32, device=dev)
    h0 = c0 = None
    if hidden is not None:
        h0, c0 = hidden[0].float().contiguous(), hidden[1].float().contiguous()
    call('sba_lstm_bidir_fwd', _p(captions), _p(cap_lens), _p(emb_weight), _p(w_ih), _p(w_hh), _p(b_ih), _p(b_hh),
         _p(h0), _p(c0), _p(gx), _p(words), _p(sent), B, T, L, emb_weight.shape[0], emb_weight.shape[1], H, _stream())
    return words, sent


class LstmBidirTrainFn(torch.autograd.Function):
    """One-layer bidirectional LSTM over packed sequences with gradients (RNN_ENCODER in training mode: DAMSM
    pre-training, pretrain_DAMSM.py:79-100).  x [B][T][ninput] is the (dropped-out) embedded caption batch; the
    recurrence and back-propagation through time are csrc/text.hip kernels, the dense projections around them are
    plain GEMMs through the BLAS library (torch.matmul).  Returns (words_emb B x 2H x L, sent_emb B x 2H)."""

    @staticmethod
    def forward(ctx, x, cap_lens, w_ih, w_hh, b_ih, b_hh, h0, c0, L):
        _need_gpu(x)
        B, T, _ = x.shape
        H = w_hh.shape[2]
        dev = x.device
        x2 = x.reshape(B * T, -1).float().contiguous()
        gx = torch.matmul(x2.unsqueeze(0), w_ih.transpose(1, 2)) + (b_ih + b_hh).unsqueeze(1)     # [2][B*T][4H]
        gx = gx.contiguous()
        cap_lens = cap_lens.to(device=dev, dtype=torch.int64).contiguous()
        words = torch.empty((B, 2 * H, L), dtype=torch.float32, device=dev)
        sent = torch.empty((B, 2 * H), dtype=torch.float32, device=dev)
        gates = torch.zeros((2, B * T, 4 * H), dtype=torch.float32, device=dev)
        cs = torch.zeros((2, B * T, H), dtype=torch.float32, device=dev)
        hs = torch.zeros((2, B * T, H), dtype=torch.float32, device=dev)
        w_hh = w_hh.contiguous()
        call('sba_lstm_recur_train', _p(gx), _p(cap_lens), _p(w_hh), _p(h0), _p(c0), _p(words), _p(sent), _p(gates),
             _p(cs), _p(hs), B, T, L, H, _stream())
        ctx.save_for_backward(x2, cap_lens, w_ih, w_hh, gates, cs, hs)
        ctx.h0c0, ctx.dims = (h0, c0), (B, T, L, H)
        return words, sent

    @staticmethod
    def backward(ctx, dwords, dsent):
        x2, cap_lens, w_ih, w_hh, gates, cs, hs = ctx.saved_tensors
        B, T, L, H = ctx.dims
        h0, c0 = ctx.h0c0
        dev = x2.device
        dwords = torch.zeros((B, 2 * H, L), device=dev) if dwords is None else dwords.float().contiguous()
        dsent = torch.zeros((B, 2 * H), device=dev) if dsent is None else dsent.float().contiguous()
        dG = torch.zeros((2, B * T, 4 * H), dtype=torch.float32, device=dev)
        hprev = torch.zeros((2, B * T, H), dtype=torch.float32, device=dev)
        call('sba_lstm_recur_bwd', _p(cap_lens), _p(w_hh), _p(h0), _p(c0), _p(gates), _p(cs), _p(hs), _p(dwords),
             _p(dsent), _p(dG), _p(hprev), B, T, L, H, _stream())
        dGt = dG.transpose(1, 2)                                   # [2][4H][B*T]
        dw_ih = torch.matmul(dGt, x2.unsqueeze(0))                 # [2][4H][ninput]
        dw_hh = torch.matmul(dGt, hprev)                           # [2][4H][H]
        db = dG.sum(1)                                             # [2][4H]
        dx = torch.matmul(dG, w_ih).sum(0).view(B, T, -1) if ctx.needs_input_grad[0] else None
        return dx, None, dw_ih, dw_hh, db, db, None, None, None
