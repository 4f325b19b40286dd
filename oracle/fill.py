"""Closed-form, bit-reproducible tensor fills shared by the golden-vector
generator (tools/make_golden.py, run against the reference in the build
container) and the tests that rebuild the same inputs on the GPU box.

TEST INFRASTRUCTURE ONLY (see oracle/sbagan_oracle.py header).

Values come from exact 64-bit integer hashing (no libm, no RNG state), so the
same (shape, tag) gives the same bits on any machine.  weights_init's
orthogonal_ (miscc/utils.py:286-296) is LAPACK dependent and is deliberately
not used for parity fixtures.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on a uint64 array."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def uniform(shape, tag, lo=-1.0, hi=1.0):
    """U[lo, hi) float32 tensor of `shape`, a pure function of (shape, tag)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(tag) * np.uint64(0x9E3779B97F4A7C15)
        h = _mix(idx + np.uint64(0x632BE59BD9B4E019))
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)       # 24-bit mantissa
    v = (lo + (hi - lo) * u).astype(np.float32)
    return torch.from_numpy(v.reshape(shape))


def unit(shape, tag):
    """zero-mean unit-variance (uniform) tensor."""
    s = float(np.sqrt(3.0))
    return uniform(shape, tag, -s, s)


def tag_of(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def fill_state_dict(shapes, salt=0, gain=1.4):
    """Deterministic parameters for a network given {name: shape} in
    state_dict order.  Conv/Linear weights: uniform with variance
    gain^2 / fan_in; BN gamma around 1, beta around 0; biases small;
    running stats at their construction values."""
    P = {}
    names = list(shapes.keys())
    bn_prefixes = set(n[:-len('.running_mean')] for n in names if n.endswith('.running_mean'))
    for name in names:
        shape = tuple(shapes[name])
        t = tag_of(name) + salt
        pre = name.rsplit('.', 1)[0]
        if name.endswith('.running_mean'):
            P[name] = torch.zeros(shape)
        elif name.endswith('.running_var'):
            P[name] = torch.ones(shape)
        elif name.endswith('.num_batches_tracked'):
            P[name] = torch.zeros(shape, dtype=torch.long)
        elif pre in bn_prefixes and name.endswith('.weight'):
            P[name] = 1.0 + 0.1 * uniform(shape, t)
        elif pre in bn_prefixes and name.endswith('.bias'):
            P[name] = 0.1 * uniform(shape, t)
        elif name.endswith('.bias'):
            P[name] = 0.05 * uniform(shape, t)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            P[name] = unit(shape, t) * (gain / np.sqrt(fan_in))
    return P


class StandInImageEncoder(object):
    """Small differentiable stand-in for CNN_ENCODER (model.py:162-267) used
    ONLY to close the DAMSM gradient path in fixtures and tests: the real
    encoder is torchvision's Inception-v3, which is third-party arithmetic and
    not available offline.  regions = conv1x1(avgpool->17x17),
    code = Linear(global mean)."""

    def __init__(self, nef=256, device='cpu', dtype=torch.float32):
        self.nef = nef
        self.wr = (unit((nef, 3, 1, 1), 7001) * 0.8).to(device=device, dtype=dtype)
        self.wc = (unit((nef, 3), 7002) * 0.8).to(device=device, dtype=dtype)
        self.bc = (0.1 * uniform((nef,), 7003)).to(device=device, dtype=dtype)

    def __call__(self, x):
        import torch.nn.functional as F
        p = F.adaptive_avg_pool2d(x, 17)
        region = F.conv2d(p, self.wr)
        code = F.linear(x.mean((2, 3)), self.wc, self.bc)
        return region, code


def synthetic_captions(B, words_num=20, lmax=18, vocab=5450, tag=11):
    """CUB-shaped caption batch: int64 ids in [1, vocab),
    zero padded, lengths in [5, lmax] sorted descending with max == lmax."""
    lens = (uniform((B,), tag, 5, lmax + 1)).floor().long().clamp(5, lmax)
    lens[0] = lmax
    lens, _ = torch.sort(lens, 0, True)
    ids = uniform((B, words_num), tag + 1, 1, vocab).floor().long().clamp(1, vocab - 1)
    pos = torch.arange(words_num)[None, :]
    ids = torch.where(pos < lens[:, None], ids, torch.zeros_like(ids))
    return ids, lens
