#!/usr/bin/env python3
"""Measure every tile configuration / K split of sba_conv_igemm on the implicit-GEMM shapes one training step
launches, and write the fastest per shape to sba-gan_amd/sbagan/igemm_table.json (run on an MI355X:
`python tools/tune_igemm.py [--batch 20]`).  Each candidate is timed as 20 back-to-back launches replayed from a
hipGraph (launch overhead included: that is what the step pays)."""
import copy
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

NTILES = 18
TILE_BM = {1: 64, 2: 64, 3: 96, 4: 96, 5: 128, 6: 128, 7: 128, 8: 128, 9: 256, 10: 256, 11: 320, 12: 96, 13: 320, 14: 160, 15: 160, 16: 256, 17: 256, 18: 256}
TILE_BN = {1: 64, 2: 64, 3: 64, 4: 64, 5: 64, 6: 64, 7: 128, 8: 128, 9: 64, 10: 64, 11: 128, 12: 128, 13: 64, 14: 64, 15: 64, 16: 128, 17: 128, 18: 128}


def main():
    sys.argv += ['--child']
    args = bench.parse()
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    from sbagan import ops
    from sbagan._lib import ConvGeom, call
    from sbagan.synth import synthetic_batch
    os.environ['SBA_IGEMM_TABLE'] = '0'
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
        if isinstance(g, tuple):        # ('group', tile, geometries): a grouped launch of the image encoder, not tuned here
            continue
        k = ops.geom_key(g)
        if k not in uniq:
            uniq[k] = [g, 0]
        uniq[k][1] += 1
    print('%d implicit-GEMM launches per step, %d distinct shapes' % (len(log), len(uniq)), flush=True)
    del step
    torch.cuda.empty_cache()
    ws = ops.workspace(dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
    ws = ops.workspace(dev)
    st = torch.cuda.current_stream().cuda_stream
    table, report = {}, []
    total_rule = total_best = 0.0
    # TUNE_MAX_M=n: re-measure only the shapes with at most n rows, keep the committed entries of the others
    max_m = int(os.environ.get('TUNE_MAX_M', '0'))
    # TUNE_MIN_GFLOP=f: re-measure only the shapes of at least f GFLOP (the large GEMM-like layers), keep the others
    min_gflop = float(os.environ.get('TUNE_MIN_GFLOP', '0'))
    out = os.path.join(ROOT, 'sba-gan_amd', 'sbagan', 'igemm_table.json')
    if max_m or min_gflop:
        with open(out) as f:
            table = json.load(f)['bf16']
    for k, (g0, count) in sorted(uniq.items()):
        if max_m and g0.N * g0.OHs * g0.OWs > max_m:
            continue
        if min_gflop and 2.0 * g0.N * g0.OHs * g0.OWs * g0.Cout * g0.ntaps * g0.Cin < min_gflop * 1e9:
            continue
        table.pop(k, None)
        g = ConvGeom()
        ctypes.memmove(ctypes.byref(g), ctypes.byref(g0), ctypes.sizeof(ConvGeom))
        xcs = g.x_cstride or g.Cin
        ycs = g.y_cstride or g.Cout
        x = torch.randn(g.N, g.IH, g.IW, xcs, device=dev).bfloat16()
        w = (torch.randn(g.Cout, g.ntaps, g.Cin, device=dev) / (g.Cin * g.ntaps) ** 0.5).bfloat16()
        y = torch.empty(g.N, g.OH, g.OW, ycs, device=dev, dtype=torch.bfloat16)
        M = g.N * g.OHs * g.OWs
        nslabs = g.ntaps * (g.Cin // 32)

        def timeit(tile, ksplit):
            g.tile, g.ksplit = tile, ksplit

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
            for _ in range(2):
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            return best
        t_rule = timeit(0, 0)
        cands = []
        for tile in range(1, NTILES + 1):
            bm, bn = TILE_BM[tile], TILE_BN[tile]
            if bm >= 2 * M + 64 or (bn == 128 and g.Cout <= 64) or (tile in (13, 14, 15) and (g.Cin % 64 or M > 4096)) or (tile >= 16 and (g.Cin % 64 or M < 2048)):
                continue
            tiles = -(-M // bm) * -(-g.Cout // bn)
            splits = [1]
            if M * g.Cout * 4 <= ops.WORKSPACE_BYTES and g.Cout % 4 == 0 and nslabs >= 16 and tiles < 512:
                splits += [s for s in (2, 3, 4, 6, 8, 12, 16, 24, 32, 48) if s <= nslabs // 4 and tiles * s <= 1024]
            for sp in splits:
                cands.append((timeit(tile, sp), tile, sp))
        cands.sort()
        t_best, tile, sp = cands[0]
        if t_best < 0.97 * t_rule:
            table[k] = [tile, sp]
        else:
            t_best = t_rule
        total_rule += t_rule * count
        total_best += t_best * count
        report.append('%-44s x%-3d M=%-6d N=%-5d K=%-6d rule %7.1f us  best %7.1f us  tile %2d split %2d   next: %s'
                      % (k, count, M, g.Cout, g.ntaps * g.Cin, t_rule, cands[0][0], tile, sp,
                         ', '.join('%d/%d %.1f' % (t2, s2, u2) for u2, t2, s2 in cands[1:5])))
        print(report[-1], flush=True)
    print('sum over one step: rule table %.3f ms -> measured table %.3f ms' % (total_rule / 1e3, total_best / 1e3))
    if os.environ.get('TUNE_OUT'):
        out = os.environ['TUNE_OUT']
    with open(out, 'w') as f:
        json.dump({'bf16': table, 'note': 'tools/tune_igemm.py on MI355X, B=%d: shape key -> [tile, ksplit]' % args.batch},
                  f, indent=0, sort_keys=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'tune_igemm_report.txt'), 'w') as f:
        f.write('\n'.join(report) + '\nsum over one step: rule table %.3f ms -> measured table %.3f ms\n'
                % (total_rule / 1e3, total_best / 1e3))


if __name__ == '__main__':
    main()
