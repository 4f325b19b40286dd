#!/usr/bin/env python3
"""VERDICT r3 next #6, go / no-go on numbers: BatchNorm folded into the conv where one tile owns a channel's whole column
(the discriminators' layers with 4x4 output maps: M = 16 B rows per real | fake group).  Folding needs the NO-split-K path
with an M tile that holds all rows of a group (tile ids 13 / 14: 320 x 64 / 160 x 64), so the question is whether

    conv with a column-owning tile, one K pass   (+ ~0 for statistics / normalise / LeakyReLU in its epilogue)

beats today's   conv with the table's tile and K split  +  splitk_finish  +  bn_fwd_fused.
Every 4x4-map implicit-GEMM shape of one step is timed both ways (20 launches per hipGraph replay, launch overhead included)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    sys.argv += ['--child']
    args = bench.parse()
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    from sbagan import ops
    from sbagan._lib import ConvGeom, call
    from sbagan.synth import synthetic_batch
    step = bench.build(args, dev)
    b = synthetic_batch(args.batch, branch_num=args.branch, device=dev, seed=100)
    noise = torch.randn((args.batch, 100), device=dev)
    a = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
    step.step(*a)
    ops.IGEMM_LOG = []
    step.step(*a)
    torch.cuda.synchronize()
    log, ops.IGEMM_LOG = ops.IGEMM_LOG, None
    uniq = {}
    for g in log:
        if isinstance(g, tuple):
            continue
        if g.OH == 4 and g.OW == 4 and g.OHs == 4 and g.Cin % 64 == 0:      # forward convs onto a 4x4 map
            uniq.setdefault(ops.geom_key(g), g)
    del step
    torch.cuda.empty_cache()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
    ws = ops.workspace(dev)
    st = torch.cuda.current_stream().cuda_stream
    print('%-40s %6s %6s %6s | %-22s | %s' % ('shape', 'M', 'N', 'K', 'today: tile/split us', 'one K pass, column-owning tile: us'))
    for k, g0 in sorted(uniq.items()):
        g = ConvGeom()
        ctypes.memmove(ctypes.byref(g), ctypes.byref(g0), ctypes.sizeof(ConvGeom))
        x = torch.randn(g.N, g.IH, g.IW, g.x_cstride or g.Cin, device=dev).bfloat16()
        w = (torch.randn(g.Cout, g.ntaps, g.Cin, device=dev) / (g.Cin * g.ntaps) ** 0.5).bfloat16()
        y = torch.empty(g.N, g.OH, g.OW, g.y_cstride or g.Cout, device=dev, dtype=torch.bfloat16)
        M = g.N * 16

        def timeit():
            def run():
                call('sba_conv_igemm', 1, x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, ctypes.byref(g),
                     ws.data_ptr(), ops.WORKSPACE_BYTES, st)
            run()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                for _ in range(20):
                    run()
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            return best
        g._tuned = None
        ops.tune_geom(g, torch.bfloat16)
        today = (g.tile, g.ksplit, timeit())
        owned = []
        for tile in (13, 14, 9):            # 320 x 64, 160 x 64 (one real | fake group of B = 20 per M tile ... two for 320), 256 x 64
            g.tile, g.ksplit = tile, 1
            owned.append('%d: %.1f' % (tile, timeit()))
        print('%-40s %6d %6d %6d | tile %2d split %2d %6.1f | %s' % (k, M, g.Cout, g.ntaps * g.Cin, today[0], today[1], today[2],
                                                                     '   '.join(owned)), flush=True)
    print('(today also pays splitk_finish 8.5-14 us where split > 1 and bn_fwd_fused 9-27 us per layer: '
          'profiles/r04_ab_stream_priorities.txt, longest path)')


if __name__ == '__main__':
    main()
