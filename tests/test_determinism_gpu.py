"""Deterministic-reduction mode (include/sbagan_hip.h: sba_set_deterministic): with it on, the training step is
BIT-reproducible -- run to run, and across the launch modes bench.py chooses from (eager launches over several
streams, one hipGraph of the whole step, the native multi-stream replayer, one hipGraph per phase).

What this proves: every run-to-run difference of the default mode comes from the ORDER in which f32 partial sums
meet (atomics, split-K), not from a race -- a missing dependency between two launches, a buffer reused while a
kernel on another stream still reads it, or a stale packed weight would break bit equality here, whatever the order
of the sums.  The reference itself (PyTorch on cuDNN, cudnn.deterministic = False) makes no such promise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import FULL, d_shapes, rel_l2  # noqa: E402
from oracle import fill  # noqa: E402
from test_step_gpu import _build_step, _cfg  # noqa: E402,F401  (autouse cfg fixture)

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.fixture()
def det():
    from sbagan import ops
    ops.set_deterministic(True)
    yield
    ops.set_deterministic(False)


class _OrderedStandIn(fill.StandInImageEncoder):
    """The stand-in image encoder of the fixtures (a TEST DOUBLE: 17 x 17 adaptive average pool, 1 x 1 conv, linear
    code) with the pooling written as two dense products: torch's adaptive_avg_pool2d backward adds overlapping
    windows with atomics, which made the gradient w.r.t. the 256 px image -- and with it the whole generator
    gradient -- differ in the last bits between two replays (found with tools/det_repro.py: 2 of 299 replays).
    The product's own image encoder (sbagan.inception_hip) has no such step; it is the other parametrisation."""

    def _pool(self, n, dev):
        key = (n, str(dev))
        if getattr(self, '_pm', None) is None:
            self._pm = {}
        if key not in self._pm:
            m = torch.zeros(17, n)
            for i in range(17):
                a, b = (i * n) // 17, -((-(i + 1) * n) // 17)      # adaptive pooling window [floor, ceil)
                m[i, a:b] = 1.0 / (b - a)
            self._pm[key] = m.to(dev)
        return self._pm[key]

    def __call__(self, x):
        ph, pw = self._pool(x.shape[2], x.device), self._pool(x.shape[3], x.device)
        p = torch.matmul(torch.matmul(ph, x), pw.t())                # [N, 3, 17, 17]
        region = torch.einsum('oc,ncij->noij', self.wr.view(self.nef, 3), p)
        code = torch.nn.functional.linear(x.mean((2, 3)), self.wc, self.bc)
        return region, code


def _state(st, out):
    r = {'loss/%s' % k: v.detach().float().reshape(1).clone() for k, v in out.items() if torch.is_tensor(v)}
    r['grad/G'] = st.flatG.grad.clone()
    r['param/G'] = st.flatG.data.clone()
    r['ema/G'] = st.flatG.avg.clone()
    for i, f in enumerate(st.flatD):
        r['grad/D%d' % i] = f.grad.clone()
        r['param/D%d' % i] = f.data.clone()
    for i, net in enumerate([st.netG] + st.netsD):
        for n, t in net.named_buffers():
            if n.endswith(('running_mean', 'running_var')):
                r['buf/%d/%s' % (i, n)] = t.detach().float().clone()
    for i, f in enumerate(st.fake_imgs):
        r['fake/%d' % i] = f.float().clone()
    return r


def _diff(a, b, tag=None):
    """keys whose tensors are not bit-identical, with their relative L2 distance"""
    assert set(a) == set(b)
    d = sorted((k, rel_l2(a[k], b[k])) for k in a if not torch.equal(a[k], b[k]))
    if d and tag:       # the full list for the post-mortem (the assertion message shows the first few)
        import json
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        if os.path.isdir(out):
            with open(os.path.join(out, 'det_diff_%s.json' % tag.replace(' ', '_')), 'w') as f:
                json.dump(d, f, indent=1)
    return d


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('encoder', ['standin', 'inception'])
def test_step_is_bit_reproducible_in_every_launch_mode(dev, det, encoder, dt):
    """ONE step from IDENTICAL state (B = 20; `inception` + bf16 = the benched combination): the training state is
    snapshotted after warm-up and capture, and restored before every run.  Two eager runs, two replays of the
    whole-step hipGraph and two native replays must agree BIT FOR BIT in every loss, every network's gradient, the
    updated parameters, the EMA shadow, the BatchNorm running statistics and the fake images; so must two eager runs
    with the reference's placement of the DAMSM terms (inside generator_loss) and two replays of the per-phase
    graphs (the benched launch mode), which share that placement."""
    from sbagan import ops
    from sbagan.synth import synthetic_batch
    from sbagan.trainer import GraphedStep, ReplayedStep
    ops.set_compute_dtype(dt)
    B = 20
    b = synthetic_batch(B, device=dev, seed=100)
    gen = torch.Generator(device='cpu')
    gen.manual_seed(1234)
    noise = torch.randn((B, 100), generator=gen).to(dev)
    eps = torch.randn((B, 100), generator=gen).to(dev)
    args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    st = _build_step(dev, B, encoder=encoder)
    if encoder == 'standin':
        st.image_encoder = _OrderedStandIn(256, device=dev)
    orig = st.phase_a
    st.phase_a = lambda se, we, m, nz, e=None: orig(se, we, m, nz, eps)      # fixed eps, eager and captured
    for _ in range(2):
        st.step(*args)
    whole = GraphedStep(st, *args, single=True)
    phases = GraphedStep(st, *args)
    phases_native = GraphedStep(st, *args, native=True)      # the same per-phase captures through the native replayer
    rs = ReplayedStep(st, *args)
    rs.draw = False
    rs.eps.copy_(eps)
    torch.cuda.synchronize()
    snap = st.snapshot()

    def run(fn):
        st.restore(snap)
        phases.resync()
        out = fn()
        torch.cuda.synchronize()
        return _state(st, out)

    def eager():
        return st.step(*args)

    def eager_late():       # DAMSM terms inside generator_loss, as the per-phase graphs (and the reference) have them
        st.early_damsm = False
        try:
            return st.step(*args)
        finally:
            st.early_damsm = True

    def replay_of(g):
        def f():
            g.replay()
            return g.out
        return f

    tag = '%s_%s' % (encoder, str(dt).split('.')[-1])
    e1, e2 = run(eager), run(eager)
    assert not _diff(e1, e2, 'eager_' + tag), ('eager vs eager', _diff(e1, e2)[:8])
    for name, fn in (('whole-step hipGraph', replay_of(whole)), ('native replayer', replay_of(rs))):
        for k in range(2):
            d = _diff(run(fn), e1, name + '_' + tag)
            assert not d, ('%s, replay %d vs eager' % (name, k), d[:8])
    # sba_replay_prioritize: its calibration pass IS one execution of the recording (every launch alone, in order, on one
    # stream), and the streams it assigns afterwards -- one pool, or two pools of different HIP priority -- enforce the
    # recorded dependencies: the pass itself and the replays behind it are the same step, bit for bit
    def prioritized(spec):
        def f():
            rs.prioritize(spec)
            return rs.out
        return f
    for spec in ('c:4:1:0.05', '2:5:2:0.1', '1:6:3:0.3'):
        d = _diff(run(prioritized(spec)), e1, 'prioritize_' + tag)
        assert not d, ('calibration pass %s vs eager' % spec, d[:8])
        d = _diff(run(replay_of(rs)), e1, 'prioritized_replay_' + tag)
        assert not d, ('native replayer after prioritize(%s) vs eager' % spec, d[:8])
        assert rs.info['nodes'] > 100 and rs.info['streams'] >= 2, rs.info
    st.restore(snap)
    rs.prioritize('c:4:1:0.0')          # back to one pool of four streams
    l1, l2 = run(eager_late), run(eager_late)
    assert not _diff(l1, l2, 'late_' + tag), ('eager (late DAMSM) vs itself', _diff(l1, l2)[:8])
    for k in range(2):
        d = _diff(run(replay_of(phases)), l1, 'phases_' + tag)
        assert not d, ('per-phase hipGraphs, replay %d vs eager' % k, d[:8])
    for k in range(2):      # (the natively replayed phases overlap for real, so they keep the early placement of the DAMSM terms)
        d = _diff(run(replay_of(phases_native)), e1, 'phases_native_' + tag)
        assert not d, ('per-phase captures through the native replayer, replay %d vs eager' % k, d[:8])
    # the two placements of the DAMSM terms give the same gradient by linearity, in a different summation order
    worst = max([r for _, r in _diff(e1, l1)] or [0.0])
    assert worst <= (2e-3 if dt == torch.float32 else 0.5), worst
    from sbagan._lib import lib
    assert lib.sba_det_high_water() < ops.DET_SCRATCH_BYTES, 'one step must fit the scratch ring'


@pytest.mark.parametrize('encoder,dt', [('standin', torch.float32), ('inception', torch.bfloat16)])
def test_dependency_relaxations_do_not_change_a_bit(dev, det, encoder, dt):
    """The step's fine-grained dependencies (every discriminator's generator-loss term right behind
    its own update with the image gradients handed to the generator's single backward pass, the 64 / 128 px updates
    forked from the point their fake image is issued) only reorder launches: one step from
    the same state with all of them OFF -- the reference's order, trainer.py:261-299 -- and with all of them ON must
    agree bit for bit in every loss, gradient, parameter, Adam-updated weight, EMA value, BatchNorm buffer and image;
    so must the Adam update issued in two pieces (FusedAdam.step_range), the bucketed discriminator update, and the DAMSM
    loss heads called directly (ops.damsm_terms_direct) instead of through their autograd Functions."""
    from sbagan import nets, ops
    from sbagan.synth import synthetic_batch
    ops.set_compute_dtype(dt)
    B = 20
    b = synthetic_batch(B, device=dev, seed=100)
    gen = torch.Generator(device='cpu')
    gen.manual_seed(4321)
    noise = torch.randn((B, 100), generator=gen).to(dev)
    eps = torch.randn((B, 100), generator=gen).to(dev)
    args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    st = _build_step(dev, B, encoder=encoder)
    if encoder == 'standin':
        st.image_encoder = _OrderedStandIn(256, device=dev)
    orig = st.phase_a
    st.phase_a = lambda se, we, m, nz, e=None: orig(se, we, m, nz, eps)
    for _ in range(2):
        st.step(*args)
    torch.cuda.synchronize()
    snap = st.snapshot()

    from miscc import losses
    fork_default = nets._GBase.fork_mapping

    def run(relaxed, bucket=False, direct=True):
        st.restore(snap)
        st.early_g_terms = st.early_d = relaxed
        st.bucket_adam = bucket
        # MAPPING_NET on its (guarded) side stream belongs to the set since round 4: off in the reference order
        nets._GBase.fork_mapping = bool(relaxed) and fork_default
        losses.DIRECT_DAMSM = direct
        try:
            out = st.step(*args)
            torch.cuda.synchronize()
            return _state(st, out)
        finally:
            st.early_g_terms = st.early_d = True
            st.bucket_adam = False
            nets._GBase.fork_mapping = fork_default
            losses.DIRECT_DAMSM = True

    tag = '%s_%s' % (encoder, str(dt).split('.')[-1])
    ref = run(False, direct=False)      # the DAMSM terms through WordsLossFn / SentLossFn and autograd, too
    d = _diff(run(True), ref, 'relaxed_' + tag)
    assert not d, ('relaxed dependencies vs the reference order', d[:8])
    d = _diff(run(True, bucket=True), ref, 'bucket_adam_' + tag)
    assert not d, ('bucketed discriminator update vs one Adam launch', d[:8])
    # the 64 / 128 px image heads on their discriminators' streams (opt-in, SBA_FORK_HEADS: nets._GBase.image_stream)
    st.fork_heads = True
    try:
        d = _diff(run(True), ref, 'fork_heads_' + tag)
    finally:
        st.fork_heads = False
    assert not d, ('image heads on the discriminator streams vs the reference order', d[:8])

    # The two-pass layouts (opt-in on one GPU, the data-parallel step's shape): real-image forwards first (real_first), and
    # with the real half's loss terms + backward pass at the start of the step too (real_bwd_early).  They regroup sums
    # (per-pass launches, the order the halves' gradients are added in), so they are held to rounding, not to bits: the
    # losses, and -- in f32 -- the discriminators' gradients.
    def two_pass(early):
        st.real_first, st.real_bwd_early = True, early
        try:
            return run(True)
        finally:
            st.real_first = st.real_bwd_early = False
    a, b2 = two_pass(False), two_pass(True)
    again = two_pass(True)
    assert not _diff(b2, again), 'the early real-half backward pass is not reproducible'
    for k in a:
        if k.startswith('loss/errD'):
            assert abs(float(a[k]) - float(b2[k])) <= 2e-6 * abs(float(a[k])), (k, float(a[k]), float(b2[k]))
            assert abs(float(a[k]) - float(ref[k])) <= (2e-5 if dt == torch.float32 else 2e-2) * abs(float(ref[k])), k
        if k.startswith('grad/D') and dt == torch.float32:
            assert rel_l2(b2[k], a[k]) < 1e-5, (k, rel_l2(b2[k], a[k]))


def test_small_batch_discriminator_gradients_are_reproducible(dev, det):
    """The 'bimodal' run-to-run differences of round 2 (a discriminator / generator-loss gradient at B = 3..4 moving by
    ~2e-3 between two f32 runs, tests/test_kernels_gpu.py::test_generator_loss_vs_oracle, tests/dist_worker.py) under
    the deterministic mode: eight runs of generator_loss + backward through the three discriminators are bit-identical,
    so the jumps were summation order meeting an ill-conditioned BatchNorm batch (3 samples), not a race."""
    import model
    from miscc import losses
    from sbagan import ops
    ops.set_compute_dtype(torch.float32)
    B, L = 3, 7
    nets = []
    for i, cls in enumerate((model.D_NET64, model.D_NET128, model.D_NET256)):
        P = fill.fill_state_dict(d_shapes(FULL, i), salt=i)
        for k in P:
            if k.endswith('outlogits.0.weight'):
                P[k] = P[k] * 0.1
        n = cls()
        n.load_state_dict(P)
        nets.append(n.to(dev).train())
    enc = fill.StandInImageEncoder(256, device=dev)
    fakes = [fill.uniform((B, 3, 64 * 2 ** i, 64 * 2 ** i), 970 + i) for i in range(3)]
    words, sent = fill.unit((B, 256, L), 975).to(dev), fill.unit((B, 256), 976).to(dev)
    lens, labels, cids = torch.tensor([7, 5, 2]).to(dev), torch.arange(B).to(dev), np.array([0, 1, 0])
    runs = []
    for _ in range(8):
        ops.det_reset()
        fa = [f.to(dev).requires_grad_(True) for f in fakes]
        for n in nets:
            n.zero_grad()
        tot, _ = losses.generator_loss(nets, enc, fa, torch.ones(B, device=dev), words, sent, labels, lens, cids)
        tot.backward()
        ops.join_wgrads()
        torch.cuda.synchronize()
        runs.append([tot.detach().clone()] + [f.grad.clone() for f in fa])
    for r in runs[1:]:
        for a, b0 in zip(r, runs[0]):
            assert torch.equal(a, b0), rel_l2(a, b0)


@pytest.mark.parametrize('dt', DTYPES)
def test_deterministic_kernels_match_the_oracle(dev, det, dt):
    """The ordered-reduction code paths (scratch-ring slots + fold, wave-by-wave LDS accumulation, no split-K,
    statistics by the ordered bn_stats pass) against the same references as the default paths: the kernel- and
    block-level parity tests of tests/test_kernels_gpu.py re-run with the deterministic mode on.  (Conv cases whose
    Cout / 8 is not a power of two are left out: the ordered statistics pass does not take them -- no network of the
    reference has such a BatchNorm; nor the two cases whose bf16 statistics bound assumes f32-accumulator sums: in this
    mode the statistics are those of the stored bf16 tensor.)"""
    import test_kernels_gpu as K
    for i in (0, 1, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15, 19, 20, 21, 23, 24):
        K.test_conv_fwd_dgrad_wgrad(dev, dt, K.CONV_CASES[i])
    K.test_conv_addend_epilogue(dev, dt)
    for which in ('up', 'leak', 'down', 'res'):
        K.test_conv_bn_act_blocks(dev, dt, which)
    for mask_mode in (0, 1):
        K.test_word_attention(dev, dt, mask_mode)
    K.test_adain_and_stage_entry(dev, dt)
    K.test_image_head_and_d_stem_and_logits(dev, dt)
    K.test_init_stage_and_conditioning(dev, dt)
    K.test_grouped_real_fake_pass_equals_two_calls(dev, dt)
    if dt == torch.float32:
        K.test_damsm_losses(dev)
        K.test_generator_loss_vs_oracle(dev)
