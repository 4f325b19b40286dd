"""Capture the generator forward alone in a hipGraph and compare every sub-module output of the replay with
the eager forward on the same noise / eps: the first mismatch names the operator that misbehaves under replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    args = bench.parse()
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from sbagan.synth import synthetic_batch
    from sbagan import ops
    step = bench.build(args, dev)
    b = synthetic_batch(args.batch, branch_num=args.branch, device=dev, seed=100)
    noise = torch.empty((args.batch, 100), device=dev)
    torch.manual_seed(100)
    for it in range(3):
        noise.normal_(0, 1)
        step.step(b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    torch.cuda.synchronize()
    netG = step.netG
    netG.ca_net.eps = torch.randn(args.batch, netG.ca_net.fc.weight.shape[0] // 4, device=dev) \
        if os.environ.get('FIX_EPS', '1') == '1' else None
    store = {}
    hooks = []
    for name, m in netG.named_modules():
        if name and name.count('.') <= int(os.environ.get('DEPTH', '1')):
            def hook(mod, inp, out, name=name):
                o = out[0] if isinstance(out, (tuple, list)) else out
                if torch.is_tensor(o):
                    store[name] = o
            hooks.append(m.register_forward_hook(hook))
    use_arena = os.environ.get('USE_ARENA', '1') == '1'

    def fwd():
        if use_arena:
            ops.ARENA.begin(dev)
        with torch.no_grad():
            r = netG(noise, b['sent_emb'], b['words_embs'], b['mask'])[0]
        if use_arena:
            ops.ARENA.end()
        return r
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        fwd()
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    ref_imgs = [x.clone() for x in fwd()]
    ref = {k: v.clone() for k, v in store.items()}
    order = list(store.keys())
    torch.cuda.synchronize()
    store.clear()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
        imgs = fwd()
    kept = dict(store)
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        print('replay %d' % r)
        for k in order:
            if k in kept and kept[k].shape == ref[k].shape:
                d = (kept[k].float() - ref[k].float()).abs()
                d = float(d.nan_to_num(1e9, 1e9, 1e9).max())
                flag = '' if d < 0.05 * (float(ref[k].float().abs().max()) + 1e-6) else '   <<<<'
                print('  %-40s maxdiff %.4g (ref absmax %.4g)%s' % (k, d, float(ref[k].float().abs().max()), flag))
        for i, (a, c) in enumerate(zip(imgs, ref_imgs)):
            print('  img%d maxdiff %.4g' % (i, float((a.float() - c.float()).abs().nan_to_num(1e9, 1e9, 1e9).max())))
main()
