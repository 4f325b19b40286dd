"""Repeat ONE eager step from the same state in deterministic mode and list what is not bit-identical to the first run.
    python tools/det_repro.py [iters=40] [dtype=bf16] [encoder=standin]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sba-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == 'f32') else torch.bfloat16
    encoder = sys.argv[3] if len(sys.argv) > 3 else 'standin'
    from miscc.config import cfg, reset_cfg
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3, s.LAMBDA = 4.0, 5.0, 10.0, 5.0
    from sbagan import ops
    from sbagan.synth import synthetic_batch
    from test_determinism_gpu import _state
    from test_step_gpu import _build_step
    dev = torch.device('cuda:0')
    ops.set_deterministic(True)
    ops.set_compute_dtype(dt)
    B = 20
    b = synthetic_batch(B, device=dev, seed=100)
    gen = torch.Generator(device='cpu')
    gen.manual_seed(1234)
    noise = torch.randn((B, 100), generator=gen).to(dev)
    eps = torch.randn((B, 100), generator=gen).to(dev)
    args = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    st = _build_step(dev, B, encoder=encoder)
    if encoder == 'standin' and os.environ.get('REPRO_TORCH_POOL', '0') != '1':
        from test_determinism_gpu import _OrderedStandIn
        st.image_encoder = _OrderedStandIn(256, device=dev)
    if os.environ.get('REPRO_SERIAL', '0') == '1':
        st.concurrent_d = False
        st.early_damsm = False
        st.overlap_wgrad = False
    mode = os.environ.get('REPRO_MODE', 'eager')
    orig = st.phase_a
    st.phase_a = lambda se, we, m, nz, e=None: orig(se, we, m, nz, eps)
    for _ in range(2):
        st.step(*args)
    graph = None
    if mode in ('whole', 'phases'):
        from sbagan.trainer import GraphedStep
        graph = GraphedStep(st, *args, single=(mode == 'whole'))
    elif mode == 'native':
        from sbagan.trainer import ReplayedStep
        graph = ReplayedStep(st, *args)
        graph.draw = False
        graph.eps.copy_(eps)
    torch.cuda.synchronize()
    snap = st.snapshot()
    ref, bad = None, []
    names = [n for n, _ in st.netG.named_parameters()]
    for it in range(iters):
        st.restore(snap)
        if graph is not None:
            graph.resync()
            graph.replay()
            out = graph.out
        else:
            out = st.step(*args)
        torch.cuda.synchronize()
        cur = _state(st, out)
        for n, p in zip(names, st.flatG.params):
            cur['gradG/' + n] = p.grad.clone()
        if ref is None:
            ref = cur
            continue
        d = sorted((k, float((cur[k].double() - ref[k].double()).norm() / ref[k].double().norm().clamp(min=1e-30)))
                   for k in cur if not torch.equal(cur[k], ref[k]))
        if d:
            bad.append({'iter': it, 'n': len(d), 'keys': d})
            print('iter %d: %d tensors differ; first %s' % (it, len(d), d[:3]), flush=True)
    print('%d of %d runs differ from run 0' % (len(bad), iters - 1))
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'det_repro.json'), 'w') as f:
            json.dump(bad, f, indent=1)


if __name__ == '__main__':
    main()
