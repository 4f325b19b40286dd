"""The multi-stream step's allocator-reuse hazard class (sbagan/stream_audit.py): controls that the audit sees a planted
hazard and accepts the two legitimate protections, the eager default-mode step audited clean for every generator variant,
the unguarded MAPPING_NET fork flagged and the guarded one clean, and the round-3 stress run as a test: 40 eager
generator passes with the guarded fork, bit-equal to the single-stream pass in the deterministic mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import FULL, g_shapes, make_inputs  # noqa: E402
from oracle import fill  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _cfg():
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3, s.LAMBDA = 4.0, 5.0, 10.0, 5.0
    yield


def _planted(dev, protect):
    """a block allocated on stream A, used on stream B, dropped, and allocated again on A"""
    from sbagan.stream_audit import StreamAudit
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    n = 48 << 20
    torch.cuda.synchronize()
    audit = StreamAudit(dev).start()
    try:
        with torch.cuda.stream(sa):
            x = torch.empty(n, dtype=torch.uint8, device=dev)
            x.fill_(1)
        sb.wait_stream(sa)
        with torch.cuda.stream(sb):
            y = x + 1                   # the foreign use
        if protect == 'record_stream':
            x.record_stream(sb)
        elif protect == 'join':
            sa.wait_stream(sb)
        ptr = x.data_ptr()
        del x
        with torch.cuda.stream(sa):
            z = torch.empty(n, dtype=torch.uint8, device=dev)
            z.fill_(7)
        reused = z.data_ptr() == ptr
        torch.cuda.synchronize()
        del y, z
    finally:
        hz = audit.stop()
    return hz, reused, audit.stats


def test_audit_controls(dev):
    hz, reused, stats = _planted(dev, None)
    assert stats['launches'] >= 3 and stats['foreign_uses'] >= 1, stats
    if reused:          # (the allocator handed the same bytes out again: the hazard is real)
        assert hz and hz[0]['unjoined_stream'] != hz[0]['pool_stream'], (hz, stats)
    else:
        assert stats['frees_with_unjoined_foreign_use'] >= 1, stats
    hz, _, stats = _planted(dev, 'record_stream')
    assert not hz and stats['frees_with_unjoined_foreign_use'] == 0, (hz, stats)
    hz, _, stats = _planted(dev, 'join')
    assert not hz and stats['frees_with_unjoined_foreign_use'] == 0, (hz, stats)


def _build_step(dev, B, variant):
    import test_step_gpu as T
    return T._build_step(dev, B, variant, 3, encoder='inception')


@pytest.mark.parametrize('variant', ['model', 'bert', 'mix'])
def test_eager_default_mode_step_audits_clean(dev, variant):
    """three eager steps of the benched configuration (default mode, bf16, Inception encoder, every fork the step makes:
    discriminator updates, image encoder branches, weight-gradient companions) under the audit: no block is handed out
    again while a use on another stream is unjoined"""
    from sbagan import ops
    from sbagan.stream_audit import StreamAudit
    from sbagan.synth import synthetic_batch
    ops.set_compute_dtype(torch.bfloat16)
    B = 8
    st = _build_step(dev, B, variant)
    b = synthetic_batch(B, device=dev, seed=100)
    nshape = (2, B, 100) if variant == 'mix' else (B, 100)
    noise = torch.randn(nshape, device=dev)
    args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    for _ in range(2):
        st.step(*args)
    torch.cuda.synchronize()
    audit = StreamAudit(dev).start()
    try:
        for _ in range(3):
            out = st.step(*args)
        float(out['errG_total'])
    finally:
        hz = audit.stop()
    s = audit.stats
    assert s['launches'] > 1500 and s['foreign_uses'] > 100 and s['allocator_events'] > 1000, s
    assert not hz, (len(hz), hz[:6], s)


def _generator_pass(dev, variant, B=2):
    import model
    import model_bert
    from miscc.losses import KL_loss
    x = make_inputs(FULL, B, 18, lmax=18, tag=900)
    net = {'model': model.G_NET, 'bert': model_bert.G_NET, 'mix': model_bert.G_NET_MIX}[variant]()
    P = fill.fill_state_dict(g_shapes(FULL, 3, 'model' if variant == 'model' else 'bert'))
    net.load_state_dict(P)
    net.to(dev).train()
    z = (x['z2'] if variant == 'mix' else x['z']).to(dev)
    eps = fill.unit((B, 100), 901).to(dev)
    sent, words, mask = x['sent'].to(dev), x['words'].to(dev), x['mask'].to(dev)
    shapes = [(B, 3, s, s) for s in (64, 128, 256)]
    douts = [fill.unit(s, 910 + k).to(dev) for k, s in enumerate(shapes)]

    def run():
        for p in net.parameters():
            p.grad = None
        net.ca_net.eps = eps
        imgs, _, mu, lv = net(z, sent, words, mask)
        (sum((i * d).sum() for i, d in zip(imgs, douts)) + KL_loss(mu, lv)).backward()
        torch.cuda.synchronize()
        return [i.detach().clone() for i in imgs] + [p.grad.clone() for p in net.parameters()]
    return net, run


@pytest.mark.parametrize('variant', ['model', 'bert', 'mix'])
def test_mapping_fork_stress_bit_equal(dev, variant):
    """tools/stress_generator_test.py as a test: 40 eager forward + backward passes of the generator with MAPPING_NET on
    its side stream (the guarded fork, the default) against the single-stream pass, in the deterministic mode: every
    image and every parameter gradient bit-equal every time."""
    from sbagan import nets, ops
    ops.set_compute_dtype(torch.float32)
    ops.set_deterministic(True)
    old = (nets._GBase.fork_mapping, nets._GBase._fork_guard)
    try:
        net, run = _generator_pass(dev, variant)
        nets._GBase.fork_mapping, nets._GBase._fork_guard = False, False
        ref = run()
        nets._GBase.fork_mapping, nets._GBase._fork_guard = True, True
        for k in range(40):
            got = run()
            bad = [i for i, (a, b) in enumerate(zip(got, ref)) if not torch.equal(a, b)]
            assert not bad, (variant, k, bad[:8])
    finally:
        nets._GBase.fork_mapping, nets._GBase._fork_guard = old
        ops.set_deterministic(False)


def test_audit_flags_the_unguarded_mapping_fork(dev):
    """the round-3 race (two mapping calls, fork without record_stream): the audit reports the hazard class without
    having to win the race; with the guard (record_stream in both directions at the boundary) it is clean"""
    from sbagan import nets, ops
    from sbagan.stream_audit import StreamAudit
    ops.set_compute_dtype(torch.float32)
    old = (nets._GBase.fork_mapping, nets._GBase._fork_guard)
    res = {}
    try:
        net, run = _generator_pass(dev, 'mix')
        for guard in (False, True):
            nets._GBase.fork_mapping, nets._GBase._fork_guard = True, guard
            run()
            audit = StreamAudit(dev).start()
            try:
                for _ in range(3):
                    run()
            finally:
                res[guard] = (audit.stop(), audit.stats)
    finally:
        nets._GBase.fork_mapping, nets._GBase._fork_guard = old
    assert not res[True][0], ('guarded fork', res[True][0][:6], res[True][1])
    # the unguarded fork frees blocks with an unjoined use on the other stream (whether the allocator reuses them in
    # these three passes or not)
    assert res[False][0] or res[False][1]['frees_with_unjoined_foreign_use'] > 0, res[False][1]
