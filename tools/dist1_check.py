"""ONE rank of the data-parallel path (gloo, world size 1) in the deterministic mode: two eager data-parallel steps vs
two replays of the per-phase hipGraphs from the same state, with the overlap features toggled.
    python tools/dist1_check.py [overlap_g=1] [bucket_d=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sba-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    og = (sys.argv[1] if len(sys.argv) > 1 else '1') == '1'
    bd = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', '29533'
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    dist.init_process_group('gloo', rank=0, world_size=1)
    from dist_worker import build
    from helpers import FULL, make_inputs, rel_l2
    from miscc.config import cfg, reset_cfg
    from oracle import fill
    from sbagan import ops
    from sbagan.trainer import GraphedStep
    reset_cfg()
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.TREE.BRANCH_NUM = 32, 64, 3
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3, s.LAMBDA = 4.0, 5.0, 10.0, 5.0
    ops.set_compute_dtype(torch.float32)
    ops.set_deterministic(True)
    B = 4
    x = make_inputs(FULL, B, 18, lmax=18, tag=500)
    imgs = [i.to(dev) for i in x['imgs']]
    sent, words, mask, lens = x['sent'].to(dev), x['words'].to(dev), x['mask'].to(dev), x['cap_lens'].to(dev)
    noise, eps = fill.unit((B, 100), 550).to(dev), fill.unit((B, 100), 560).to(dev)
    dp = build(dev, B, True)
    dp.overlap_g, dp.bucket_d = og, bd
    orig_a = dp.phase_a
    dp.phase_a = lambda se, we, m, nz, e=None: orig_a(se, we, m, nz, eps)
    gargs = (imgs, sent, words, mask, lens, x['class_ids'], noise)
    dp.early_damsm = False
    flats = [dp.flatG] + dp.flatD

    def state():
        d = {('G' if k == 0 else 'D%d' % (k - 1)) + '.' + n: t for k, f in enumerate(flats)
             for n, t in (('data', f.data), ('grad', f.grad), ('m', f.m), ('v', f.v))}
        d['G.state'] = dp.optG.state.float()
        d['G.avg'] = dp.flatG.avg
        return d
    dp.step(*gargs)
    dp.finish()
    snap = dp.snapshot()
    res = {}
    for steps in (1, 2):
        dp.restore(snap)
        for _ in range(steps):
            out = dp.step(*gargs)
        dp.finish()
        torch.cuda.synchronize()
        res['eager%d' % steps] = ({k: v.clone() for k, v in state().items()}, {k: float(v) for k, v in out.items()})
    graph = GraphedStep(dp, *gargs)
    for steps in (1, 2):
        dp.restore(snap)
        graph.resync()
        for _ in range(steps):
            graph.replay()
        graph.finish()
        torch.cuda.synchronize()
        st, lo = state(), {k: float(v) for k, v in graph.out.items()}
        ref_st, ref_lo = res['eager%d' % steps]
        bad = [(k, '%.2e' % rel_l2(st[k], ref_st[k])) for k in st if not torch.equal(st[k], ref_st[k])]
        badl = [(k, lo[k], ref_lo[k]) for k in ref_lo if lo[k] != ref_lo[k]]
        print('overlap_g=%d bucket_d=%d steps=%d: tensors %r losses %r' % (og, bd, steps, bad, badl[:4]), flush=True)
        print('   moved from the snapshot: eager G.data %.3e graph G.data %.3e; state eager %r graph %r' % (
            rel_l2(ref_st['G.data'], snap[0]['data']), rel_l2(st['G.data'], snap[0]['data']),
            ref_st['G.state'].tolist(), st['G.state'].tolist()), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
