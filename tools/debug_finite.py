"""Run eager steps of the bench workload and print every loss per step; at the first non-finite value
report which parameters / gradients of which network are non-finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

PREV = {}


def poison():
    """torch.empty / empty_like return NaN-filled float tensors: a kernel that reads (or accumulates into) an
    output it assumed zero shows up as NaN on the first step instead of depending on what the allocator
    hands back."""
    e, el = torch.empty, torch.empty_like
    def fill(t):
        if t.is_floating_point() and t.is_cuda:
            t.fill_(float('nan'))
        return t
    torch.empty = lambda *a, **k: fill(e(*a, **k))
    torch.empty_like = lambda *a, **k: fill(el(*a, **k))


def main():
    if os.environ.get('POISON', '0') == '1':
        poison()
    n = int(os.environ.get('N_STEPS', '45'))
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    args = bench.parse()
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from sbagan.synth import synthetic_batch
    step = bench.build(args, dev)
    b = synthetic_batch(args.batch, branch_num=args.branch, device=dev, seed=100)
    noise = torch.empty((args.batch, 100), device=dev)
    torch.manual_seed(100)
    graph = None
    n_eager = int(os.environ.get('N_EAGER', str(n)))
    for it in range(n):
        if it < n_eager:
            noise.normal_(0, 1)
            out = step.step(b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
        else:
            if graph is None:
                from sbagan.trainer import GraphedStep
                torch.cuda.synchronize()
                graph = GraphedStep(step, b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'],
                                    b['class_ids'], noise, prologue=lambda: noise.normal_(0, 1),
                                    single=os.environ.get('SBA_GRAPH_SINGLE', '0') == '1')
                out = graph.out
                print('captured', flush=True)
                global GA_IMGS
                GA_IMGS = step.fake_imgs
            mode = os.environ.get('REPLAY', 'normal')
            if os.environ.get('PROBE_GA') and it > n_eager:
                def stats(tag, imgs_):
                    print('   %s ' % tag + ' | '.join('mean %.4f std %.4f absmax %.3f' % (float(x.float().mean()), float(x.float().std()), float(x.float().abs().max())) for x in imgs_), flush=True)
                for r in range(3):
                    graph.gA.replay()
                    torch.cuda.synchronize()
                    stats('gA replay', step._ctx[0] if step._ctx else GA_IMGS)
                with torch.no_grad():
                    fi = step.netG(noise, b['sent_emb'], b['words_embs'], b['mask'])[0]
                torch.cuda.synchronize()
                stats('eager fwd', fi)
            if mode == 'normal':
                graph.replay()
            else:                     # 'serial': discriminator graphs on the main stream; 'sync': + device sync between graphs
                graph.gA.replay()
                for g in graph.gD:
                    if mode == 'sync':
                        torch.cuda.synchronize()
                    g.replay()
                if mode == 'sync':
                    torch.cuda.synchronize()
                graph.gB.replay()
            torch.cuda.synchronize()
        vals = {k: (float(v) if v.numel() == 1 else [round(float(x), 4) for x in v.flatten()[:6]])
                for k, v in out.items() if torch.is_tensor(v)}
        print(it, ' '.join('%s=%s' % (k, ('%.4g' % v) if isinstance(v, float) else v) for k, v in vals.items()), flush=True)
        if os.environ.get('GRADS', '0') == '1':
            torch.cuda.synchronize()
            nets_ = [('G', step.flatG, step.netG)] + [('D%d' % i, f, step.netsD[i]) for i, f in enumerate(step.flatD)]
            for name, fp, net in nets_:
                names = [n_ for n_, _ in net.named_parameters()]
                cur = {}
                for pn, p, o in zip(names, fp.params, fp.offsets):
                    g = fp.grad[o:o + p.numel()]
                    cur[pn] = (float(g.abs().nan_to_num(1e30, 1e30, 1e30).max()), float(p.data.abs().max()))
                prev = PREV.get(name)
                PREV[name] = cur
                tot = float(fp.grad.abs().nan_to_num(1e30, 1e30, 1e30).max())
                line = '   %s grad absmax %.4g' % (name, tot)
                if prev is not None:
                    ratios = sorted(((cur[k][0] / (prev[k][0] + 1e-12), k, cur[k][0], prev[k][0]) for k in cur), reverse=True)[:4]
                    line += '  top ratios: ' + '; '.join('%s %.3g (%.3g<-%.3g)' % (k, r, c, p_) for r, k, c, p_ in ratios)
                print(line[:700], flush=True)
        if os.environ.get('WATCH'):
            torch.cuda.synchronize()
            fp, net = step.flatG, step.netG
            for pn, p, o in zip([n_ for n_, _ in net.named_parameters()], fp.params, fp.offsets):
                if pn in os.environ['WATCH'].split(','):
                    sl = slice(o, o + p.numel())
                    f = lambda t: ' '.join('%.4g' % x for x in t[sl][:6].tolist()) + ' | absmax %.4g' % float(t[sl].abs().max())
                    print('   %s\n      data %s\n      grad %s\n      m    %s\n      v    %s' % (pn, f(fp.data), f(fp.grad), f(fp.m), f(fp.v)), flush=True)
        bad = [k for k, v in out.items() if torch.is_tensor(v) and not bool(torch.isfinite(v).all())]
        if bad:
            print('non-finite:', bad)
            for name, fp in [('G', step.flatG)] + [('D%d' % i, f) for i, f in enumerate(step.flatD)]:
                for attr in ('data', 'grad', 'm', 'v', 'avg'):
                    t = getattr(fp, attr, None)
                    if torch.is_tensor(t):
                        print(' ', name, attr, 'finite' if bool(torch.isfinite(t).all()) else 'NON-FINITE',
                              'absmax %.4g' % float(t.float().abs().nan_to_num(0, 0, 0).max()))
            break

main()
